"""Prompt steps and decode steps of ONE GPU on two streams.

The engine drives this executor exactly as it drives a prompt pool and a token pool on two GPUs
(``LLMEngine._step_split_pipelined``: a prompt step is launched whenever the prompt side is idle and collected
when it has finished, a decode step is always queued), but both sides are the same worker, the same weights and
the same cache: a block number means the same block to a prompt step and to a decode step, so there is nothing
to push and no hand-off to wait for -- the engine only moves a sequence to the decode side after its prompt
step has finished on the device.  What the two streams buy: a prompt step is bound by the matrix cores, a decode
step by HBM, and side by side they take less time than one after the other (the reference has no counterpart;
its scheduler emits either a prompt step or a decode step, vllm/core/scheduler.py:168-340).
"""
from typing import Any, Dict, Optional

import torch

from ..config import CacheConfig, DeviceConfig, ModelConfig, ParallelConfig, SchedulerConfig
from ..worker.worker import Worker
from .gpu_executor import GPUExecutor


class OverlapExecutor(GPUExecutor):
    supports_pipelined_split = True
    supports_deferred_samples = True

    def __init__(self, model_config: ModelConfig, cache_config: CacheConfig, parallel_config: ParallelConfig,
                 scheduler_config: SchedulerConfig, device_config: DeviceConfig,
                 num_gpu_blocks_override: Optional[int] = None, device_index: int = 0, worker_cls=Worker) -> None:
        if not parallel_config.sep_prompt_token or parallel_config.tensor_parallel_size != 1:
            raise ValueError("OverlapExecutor runs the split engine's two sides on one GPU: sep_prompt_token, TP 1")
        if parallel_config.num_token_pools != 1 or parallel_config.staged_prompt_cache:
            raise ValueError("one GPU holds one cache: token_pools = 1, no staged prompt cache")
        # the worker itself is an ordinary co-located one
        local = ParallelConfig(1, 1, False, parallel_config.disable_custom_all_reduce)
        super().__init__(model_config, cache_config, local, scheduler_config, device_config,
                         num_gpu_blocks_override, device_index, worker_cls)
        self.parallel_config = parallel_config
        # (CPU stand-in workers -- dry runs of the benchmark launcher -- have no streams: their steps run in turn)
        self.prompt_stream = self.decode_stream = None
        if torch.cuda.is_available():
            device = torch.device("cuda", device_index)
            self.prompt_stream = torch.cuda.Stream(device)
            self.decode_stream = torch.cuda.Stream(device)
        self._previous = None          # the decode step in flight
        self._collectable = None       # the one before it

    def execute_split(self, prompt_kwargs, decode_kwargs):
        raise NotImplementedError("OverlapExecutor serves the pipelined split loop only (pipeline_steps=True)")

    def _run(self, stream: Optional["torch.cuda.Stream"], kwargs: Dict[str, Any], track_pending: bool):
        call = dict(seq_group_metadata_list=kwargs["seq_group_metadata_list"],
                    blocks_to_swap_in=kwargs.get("blocks_to_swap_in") or {},
                    blocks_to_swap_out=kwargs.get("blocks_to_swap_out") or {},
                    blocks_to_copy=kwargs.get("blocks_to_copy") or {}, blocks_to_nw={}, defer=True,
                    track_pending=track_pending)
        if stream is None:
            return self.driver_worker.execute_model(**call)
        # everything issued so far (weights, cache set-up, the other side's finished steps) is visible to the stream
        stream.wait_stream(torch.cuda.current_stream(stream.device))
        with torch.cuda.stream(stream):
            return self.driver_worker.execute_model(**call)

    def launch_prompt(self, kwargs: Dict[str, Any], lane: int = 0):
        # (its samples are not what the next decode step continues from: track_pending off)
        return self._run(self.prompt_stream, kwargs, track_pending=False)

    def decode_send(self, kwargs: Optional[Dict[str, Any]], lane: int = 0) -> None:
        self._collectable = self._previous
        self._previous = self._run(self.decode_stream, kwargs, track_pending=True) if kwargs else None
        moved = kwargs and any(kwargs.get(k) for k in ("blocks_to_swap_in", "blocks_to_swap_out", "blocks_to_copy"))
        if moved:
            # blocks moved or copied by this step may be handed to the next prompt: its stream waits for them
            # (rare: single-sequence groups are preempted by recomputation)
            if self.prompt_stream is not None:
                self.prompt_stream.wait_stream(self.decode_stream)

    def decode_collect(self, lane: int = 0):
        pending, self._collectable = self._collectable, None
        return pending.resolve() if pending is not None else None

    def kv_bytes_pushed(self) -> int:
        return 0
