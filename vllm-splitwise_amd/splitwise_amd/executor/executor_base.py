"""Executor interface (vllm/executor/executor_base.py:9-51)."""
from abc import ABC, abstractmethod
from typing import Dict, List, Tuple

from ..sequence import SamplerOutput, SequenceGroupMetadata


class ExecutorBase(ABC):
    @abstractmethod
    def execute_model(self, seq_group_metadata_list: List[SequenceGroupMetadata],
                      blocks_to_swap_in: Dict[int, int], blocks_to_swap_out: Dict[int, int],
                      blocks_to_copy: Dict[int, List[int]],
                      blocks_to_nw: Dict[int, List[Tuple[int, int]]]) -> SamplerOutput:
        raise NotImplementedError

    @abstractmethod
    def check_health(self) -> None:
        raise NotImplementedError

    def shutdown(self) -> None:
        """Ends the worker processes an executor has started (none by default)."""

    def kv_bytes_pushed(self) -> int:
        """Bytes of KV cache pushed prompt pool -> token pool(s) so far (0 without split pools)."""
        return 0
