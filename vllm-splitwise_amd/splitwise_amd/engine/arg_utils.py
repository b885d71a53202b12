"""Engine arguments: one dataclass <-> argparse, turned into the config objects
(vllm/engine/arg_utils.py:10-51, :57-303, :305-341)."""
import argparse
import dataclasses
from dataclasses import dataclass
from typing import Optional, Tuple

from ..config import CacheConfig, DeviceConfig, ModelConfig, ParallelConfig, SchedulerConfig


@dataclass
class EngineArgs:
    model: str = "facebook/opt-125m"
    tokenizer: Optional[str] = None
    dtype: str = "auto"
    kv_cache_dtype: str = "auto"
    seed: int = 0
    max_model_len: Optional[int] = None
    pipeline_parallel_size: int = 1
    tensor_parallel_size: int = 1
    sep_prompt_token: bool = False
    token_pools: int = 1             # token pools fed by the prompt pool (the reference has one)
    staged_prompt_cache: bool = False  # prompt pool keeps only the prompts in flight, under its own block numbers
    disable_custom_all_reduce: bool = False
    block_size: int = 16
    enable_prefix_caching: bool = False
    swap_space: float = 4            # GiB
    gpu_memory_utilization: float = 0.90
    max_num_batched_tokens: Optional[int] = None
    max_num_seqs: int = 256
    max_paddings: int = 256
    prefill_batch_tokens: int = 0
    prefill_max_wait: float = 0.1
    pipeline_steps: bool = False     # launch step n+1 before step n's tokens reach the host (co-located engine)
    overlap_prompt_decode: bool = False  # one GPU: prompt steps beside the decode steps, on a stream of their own
    quantization: Optional[str] = None
    enforce_eager: bool = False
    max_context_len_to_capture: Optional[int] = None
    device: str = "cuda"
    num_gpu_blocks_override: Optional[int] = None

    @staticmethod
    def add_cli_args(parser: argparse.ArgumentParser) -> argparse.ArgumentParser:
        for f in dataclasses.fields(EngineArgs):
            name = "--" + f.name.replace("_", "-")
            if f.type is bool or f.default is False or f.default is True:
                parser.add_argument(name, action="store_true", default=f.default)
            else:
                kind = type(f.default) if f.default is not None else (
                    int if "int" in str(f.type) else float if "float" in str(f.type) else str)
                parser.add_argument(name, type=kind, default=f.default)
        return parser

    @classmethod
    def from_cli_args(cls, args: argparse.Namespace) -> "EngineArgs":
        return cls(**{f.name: getattr(args, f.name) for f in dataclasses.fields(cls)})

    def create_engine_configs(self) -> Tuple[ModelConfig, CacheConfig, ParallelConfig, SchedulerConfig,
                                             DeviceConfig]:
        model_config = ModelConfig(self.model, self.dtype, self.seed, self.max_model_len, self.quantization,
                                   self.enforce_eager, self.max_context_len_to_capture)
        cache_config = CacheConfig(self.block_size, self.gpu_memory_utilization, self.swap_space,
                                   self.kv_cache_dtype, model_config.get_sliding_window(),
                                   self.enable_prefix_caching)
        if self.overlap_prompt_decode:
            # the engine runs its split loop (a prompt side and a token side that run free of each other) over ONE
            # worker: executor/overlap_executor.py
            if (self.sep_prompt_token or self.tensor_parallel_size != 1 or self.token_pools != 1
                    or self.staged_prompt_cache or not self.pipeline_steps):
                raise ValueError("overlap_prompt_decode is a one-GPU option: it needs pipeline_steps and excludes "
                                 "sep_prompt_token, tensor parallelism, token_pools and staged_prompt_cache")
        parallel_config = ParallelConfig(self.pipeline_parallel_size, self.tensor_parallel_size,
                                         self.sep_prompt_token or self.overlap_prompt_decode,
                                         self.disable_custom_all_reduce, self.token_pools, self.staged_prompt_cache)
        parallel_config.one_gpu_overlap = self.overlap_prompt_decode
        if parallel_config.staged_prompt_cache and self.enable_prefix_caching:
            raise NotImplementedError("prefix caching needs the prompt pool to keep its blocks: it cannot be "
                                      "combined with staged_prompt_cache / token_pools > 1")
        model_config.verify_with_parallel_config(parallel_config)
        scheduler_config = SchedulerConfig(self.max_num_batched_tokens, self.max_num_seqs,
                                           model_config.max_model_len, self.max_paddings,
                                           self.prefill_batch_tokens, self.prefill_max_wait,
                                           self.pipeline_steps)
        return model_config, cache_config, parallel_config, scheduler_config, DeviceConfig(self.device)
