"""Small host-side helpers shared by the scheduler, the workers and the engine."""
import enum
import socket
import uuid
from collections import deque
from typing import Dict, Iterable, List, Tuple

import torch

# Sequences that may be in flight at once when the prompt and token pools are split:
# one hand-off flag per sequence (vllm/utils.py:126-130 MAX_SLOT_IDS).
MAX_SLOT_IDS = 256

STR_DTYPE_TO_TORCH_DTYPE = {
    "half": torch.half, "float16": torch.half,
    "bfloat16": torch.bfloat16,
    "float": torch.float, "float32": torch.float,
}


class Device(enum.Enum):
    GPU = enum.auto()
    CPU = enum.auto()


class SeqToSlotMapper:
    """sequence id -> hand-off slot in [0, MAX_SLOT_IDS).

    Same assignment order as vllm/utils.py:133-161: fresh slots go out in ascending
    order, a freed slot is the next one handed out.
    """

    def __init__(self, num_slots: int = MAX_SLOT_IDS) -> None:
        self._free = deque(range(num_slots))
        self._slot_of: Dict[int, int] = {}

    def set_seq(self, seq_id: int) -> None:
        if not self._free:
            raise RuntimeError("No more slots available. Increase MAX_SLOT_IDS.")
        self._slot_of[seq_id] = self._free.popleft()

    def free_seq(self, seq_id: int) -> None:
        self._free.appendleft(self._slot_of.pop(seq_id))

    def num_free(self) -> int:
        return len(self._free)

    def get_slot_id(self, seq_id: int) -> int:
        return self._slot_of[seq_id]

    def __contains__(self, seq_id: int) -> bool:
        return seq_id in self._slot_of


def coalesce_blocks(block_list: Iterable[int]) -> List[Tuple[int, int]]:
    """Sorted block numbers -> maximal runs (start, length); vllm/utils.py:403-420.
    Duplicates start a new run, as in the reference."""
    runs: List[Tuple[int, int]] = []
    start = prev = None
    for b in sorted(block_list):
        if start is None:
            start = prev = b
        elif b == prev + 1:
            prev = b
        else:
            runs.append((start, prev - start + 1))
            start = prev = b
    if start is not None:
        runs.append((start, prev - start + 1))
    return runs


def coalesce_blocks_by_id(blocks_by_slot: Dict[int, List[int]]) -> Dict[int, List[Tuple[int, int]]]:
    """vllm/utils.py:423-426."""
    for slot in blocks_by_slot:
        blocks_by_slot[slot] = coalesce_blocks(blocks_by_slot[slot])
    return blocks_by_slot


def coalesce_block_pairs(src_blocks: List[int], dst_blocks: List[int]) -> List[Tuple[int, int, int]]:
    """Block i of ``src_blocks`` goes to block i of ``dst_blocks``: maximal runs (src start, dst start, length)
    over which BOTH sides are consecutive, in table order.  The counterpart of ``coalesce_blocks`` for a prompt
    pool whose block numbers differ from the token pool's."""
    assert len(src_blocks) == len(dst_blocks)
    runs: List[Tuple[int, int, int]] = []
    for src, dst in zip(src_blocks, dst_blocks):
        if runs and runs[-1][0] + runs[-1][2] == src and runs[-1][1] + runs[-1][2] == dst:
            runs[-1] = (runs[-1][0], runs[-1][1], runs[-1][2] + 1)
        else:
            runs.append((src, dst, 1))
    return runs


def random_uuid() -> str:
    return uuid.uuid4().hex


def get_open_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def cdiv(a: int, b: int) -> int:
    return -(-a // b)
