"""Single-GPU, co-located executor (vllm/executor/gpu_executor.py:21-107): one MIXED worker in
this process; profile -> size the cache -> allocate -> capture decode graphs."""
from typing import Dict, List, Optional, Tuple

from ..config import CacheConfig, DeviceConfig, ModelConfig, ParallelConfig, SchedulerConfig
from ..sequence import SamplerOutput, SequenceGroupMetadata
from ..worker.worker import Worker
from .executor_base import ExecutorBase


def check_block_budget(max_model_len: int, block_size: int, num_gpu_blocks: int) -> None:
    """vllm/executor/utils.py:1-13."""
    if num_gpu_blocks <= 0:
        raise ValueError("No available memory for the cache blocks. Try increasing `gpu_memory_utilization` "
                         "when initializing the engine.")
    if max_model_len > block_size * num_gpu_blocks:
        raise ValueError(f"The model's max seq len ({max_model_len}) is larger than the maximum number of tokens "
                         f"that can be stored in KV cache ({block_size * num_gpu_blocks}). Try increasing "
                         "`gpu_memory_utilization` or decreasing `max_model_len` when initializing the engine.")


class GPUExecutor(ExecutorBase):
    def __init__(self, model_config: ModelConfig, cache_config: CacheConfig, parallel_config: ParallelConfig,
                 scheduler_config: SchedulerConfig, device_config: DeviceConfig,
                 num_gpu_blocks_override: Optional[int] = None, device_index: int = 0, worker_cls=Worker) -> None:
        assert parallel_config.world_size == 1, "GPUExecutor only supports single GPU."
        self.model_config = model_config
        self.cache_config = cache_config
        self.parallel_config = parallel_config
        self.scheduler_config = scheduler_config
        self.device_config = device_config
        self.driver_worker = worker_cls(model_config, parallel_config, scheduler_config, device_config,
                                        local_rank=device_index, kv_cache_dtype=cache_config.cache_dtype,
                                        is_driver_worker=True)
        self.driver_worker.init_model()
        self.driver_worker.load_model()
        self._init_cache(num_gpu_blocks_override)

    def _init_cache(self, override: Optional[int]) -> None:
        if override is None:
            num_gpu_blocks, num_cpu_blocks = self.driver_worker.profile_num_available_blocks(
                self.cache_config.block_size, self.cache_config.gpu_memory_utilization,
                self.cache_config.swap_space_bytes, self.cache_config.cache_dtype)
        else:
            per_block = self.driver_worker.get_cache_block_size_bytes(self.cache_config.block_size,
                                                                      self.cache_config.cache_dtype)
            num_gpu_blocks = override
            num_cpu_blocks = int(self.cache_config.swap_space_bytes // per_block)
        check_block_budget(self.model_config.max_model_len, self.cache_config.block_size, num_gpu_blocks)
        self.cache_config.num_gpu_blocks = num_gpu_blocks
        self.cache_config.num_cpu_blocks = num_cpu_blocks
        self.driver_worker.init_cache_engine(self.cache_config)
        self.driver_worker.warm_up_model()

    def execute_model(self, seq_group_metadata_list: List[SequenceGroupMetadata],
                      blocks_to_swap_in: Dict[int, int], blocks_to_swap_out: Dict[int, int],
                      blocks_to_copy: Dict[int, List[int]],
                      blocks_to_nw: Optional[Dict[int, List[Tuple[int, int]]]] = None, defer: bool = False):
        """``defer=True`` returns at once with the step's samples still on the device (an object with
        ``resolve() -> SamplerOutput``); the engine's pipelined step loop is its only caller."""
        return self.driver_worker.execute_model(seq_group_metadata_list, blocks_to_swap_in, blocks_to_swap_out,
                                                blocks_to_copy, blocks_to_nw or {}, defer=defer)

    supports_deferred_samples = True

    def check_health(self) -> None:
        return
