"""Configuration objects.  Field names follow vllm/config.py so engine arguments written
for the reference (sep_prompt_token, kv_cache_dtype, block_size, quantization, ...) apply."""
import json
import os
from typing import Optional

import torch

from .utils import STR_DTYPE_TO_TORCH_DTYPE

_GB = 1 << 30

# Geometry of the models the benchmark configurations name.  Weights are never fetched:
# a model is either a local directory with a config.json or one of these presets with
# seeded random weights.
MODEL_PRESETS = {
    "opt-125m": dict(architecture="opt", vocab_size=50272, hidden_size=768, num_hidden_layers=12,
                     num_attention_heads=12, num_key_value_heads=12, intermediate_size=3072,
                     max_position_embeddings=2048, torch_dtype="float16"),
    "llama-2-7b": dict(architecture="llama", vocab_size=32000, hidden_size=4096, num_hidden_layers=32,
                       num_attention_heads=32, num_key_value_heads=32, intermediate_size=11008,
                       max_position_embeddings=4096, rms_norm_eps=1e-5, rope_theta=10000.0,
                       torch_dtype="bfloat16"),
    "llama-2-13b": dict(architecture="llama", vocab_size=32000, hidden_size=5120, num_hidden_layers=40,
                        num_attention_heads=40, num_key_value_heads=40, intermediate_size=13824,
                        max_position_embeddings=4096, rms_norm_eps=1e-5, rope_theta=10000.0,
                        torch_dtype="float16"),
    "llama-2-70b": dict(architecture="llama", vocab_size=32000, hidden_size=8192, num_hidden_layers=80,
                        num_attention_heads=64, num_key_value_heads=8, intermediate_size=28672,
                        max_position_embeddings=4096, rms_norm_eps=1e-5, rope_theta=10000.0,
                        torch_dtype="bfloat16"),
    "llama-tiny": dict(architecture="llama", vocab_size=512, hidden_size=256, num_hidden_layers=2,
                       num_attention_heads=4, num_key_value_heads=2, intermediate_size=512,
                       max_position_embeddings=2048, rms_norm_eps=1e-5, rope_theta=10000.0,
                       torch_dtype="float16"),
    "opt-tiny": dict(architecture="opt", vocab_size=512, hidden_size=128, num_hidden_layers=2,
                     num_attention_heads=2, num_key_value_heads=2, intermediate_size=256,
                     max_position_embeddings=2048, torch_dtype="float16"),
}
_ALIASES = {
    "facebook/opt-125m": "opt-125m", "meta-llama/Llama-2-7b-hf": "llama-2-7b",
    "meta-llama/Llama-2-13b-hf": "llama-2-13b", "meta-llama/Llama-2-70b-hf": "llama-2-70b",
}


class ModelConfig:
    """vllm/config.py:19-331 reduced to what the hot path reads."""

    def __init__(self, model: str, dtype: str = "auto", seed: int = 0, max_model_len: Optional[int] = None,
                 quantization: Optional[str] = None, enforce_eager: bool = False,
                 max_context_len_to_capture: Optional[int] = None, overrides: Optional[dict] = None) -> None:
        self.model = model
        self.seed = seed
        self.quantization = quantization
        self.enforce_eager = enforce_eager
        hf = self._load_geometry(model)
        if overrides:
            hf.update(overrides)
        self.hf_config = hf
        self.architecture = hf["architecture"]
        if dtype == "auto":
            dtype = hf.get("torch_dtype", "float16")
            if dtype == "float32":
                dtype = "float16"              # vllm/config.py:600-607: fp32 checkpoints run in fp16
        if dtype not in STR_DTYPE_TO_TORCH_DTYPE:
            raise ValueError(f"Unknown dtype: {dtype}")
        self.dtype: torch.dtype = STR_DTYPE_TO_TORCH_DTYPE[dtype]
        derived = hf.get("max_position_embeddings", 2048)
        if max_model_len is None:
            max_model_len = derived
        elif max_model_len > derived:
            raise ValueError(f"User-specified max_model_len ({max_model_len}) is greater than the derived "
                             f"max_model_len ({derived}).")
        self.max_model_len = max_model_len
        if quantization is not None and quantization not in ("gptq", "awq"):
            raise ValueError(f"Unknown quantization method: {quantization}. Must be one of ['awq', 'gptq'].")
        if max_context_len_to_capture is None:
            max_context_len_to_capture = self.max_model_len
        self.max_context_len_to_capture = min(max_context_len_to_capture, self.max_model_len)

    @staticmethod
    def _load_geometry(model: str) -> dict:
        name = _ALIASES.get(model, model)
        if name in MODEL_PRESETS:
            return dict(MODEL_PRESETS[name])
        cfg_path = os.path.join(model, "config.json")
        if os.path.isfile(cfg_path):
            with open(cfg_path) as f:
                raw = json.load(f)
            arch = (raw.get("architectures") or [""])[0].lower()
            raw["architecture"] = "opt" if "opt" in arch else "llama"
            raw.setdefault("num_key_value_heads", raw["num_attention_heads"])
            if raw["architecture"] == "opt":
                raw.setdefault("intermediate_size", raw.get("ffn_dim"))
            return raw
        raise ValueError(f"model '{model}' is neither a known preset {sorted(MODEL_PRESETS)} nor a local "
                         "directory with a config.json (there is no network to fetch it from)")

    def verify_with_parallel_config(self, parallel_config: "ParallelConfig") -> None:
        tp = parallel_config.tensor_parallel_size
        if self.hf_config["num_attention_heads"] % tp != 0:
            raise ValueError(f"Total number of attention heads ({self.hf_config['num_attention_heads']}) must "
                             f"be divisible by tensor parallel size ({tp}).")

    def get_sliding_window(self) -> Optional[int]:
        return self.hf_config.get("sliding_window")

    def get_vocab_size(self) -> int:
        return self.hf_config["vocab_size"]

    def get_hidden_size(self) -> int:
        return self.hf_config["hidden_size"]

    def get_head_size(self) -> int:
        return self.hf_config["hidden_size"] // self.hf_config["num_attention_heads"]

    def get_total_num_kv_heads(self) -> int:
        return self.hf_config.get("num_key_value_heads", self.hf_config["num_attention_heads"])

    def get_num_kv_heads(self, parallel_config: "ParallelConfig") -> int:
        # kv heads are replicated when there are fewer of them than TP ranks (vllm/config.py:288-296)
        return max(1, self.get_total_num_kv_heads() // parallel_config.tensor_parallel_size)

    def get_num_layers(self, parallel_config: "ParallelConfig") -> int:
        return self.hf_config["num_hidden_layers"] // parallel_config.pipeline_parallel_size


class CacheConfig:
    """vllm/config.py:334-388.  fp8_e5m2 is allowed on ROCm here (the reference refuses it)."""

    def __init__(self, block_size: int, gpu_memory_utilization: float, swap_space: float,
                 cache_dtype: str = "auto", sliding_window: Optional[int] = None,
                 enable_prefix_caching: bool = False) -> None:
        self.block_size = block_size
        self.gpu_memory_utilization = gpu_memory_utilization
        self.swap_space_bytes = int(swap_space * _GB)
        self.cache_dtype = cache_dtype
        self.sliding_window = sliding_window
        self.enable_prefix_caching = enable_prefix_caching
        if gpu_memory_utilization > 1.0:
            raise ValueError(f"GPU memory utilization must be less than 1.0. Got {gpu_memory_utilization}.")
        if cache_dtype not in ("auto", "fp8_e5m2"):
            raise ValueError(f"Unknown kv cache dtype: {cache_dtype}")
        if enable_prefix_caching and cache_dtype != "auto":
            # the prefix-aware prompt kernel reads the cached context in the activation dtype
            raise NotImplementedError("prefix caching needs kv_cache_dtype 'auto'")
        self.num_gpu_blocks: Optional[int] = None
        self.num_cpu_blocks: Optional[int] = None

    def metrics_info(self) -> dict:
        """Label set of the ``vllm:cache_config`` info series: every setting as a string (config.py:352-355)."""
        return {key: str(value) for key, value in self.__dict__.items()}


class ParallelConfig:
    """vllm/config.py:391-471.  With sep_prompt_token the world is a prompt pool followed by
    a token pool of tensor_parallel_size ranks each (:433-443)."""

    def __init__(self, pipeline_parallel_size: int = 1, tensor_parallel_size: int = 1,
                 sep_prompt_token: bool = False, disable_custom_all_reduce: bool = False,
                 token_pools: int = 1, staged_prompt_cache: bool = False) -> None:
        """``token_pools`` (with sep_prompt_token): token pools fed by the one prompt pool.  The reference
        has exactly one (:434-438: as many token workers as prompt workers); with more, pool j is ranks
        tp * (1 + j) .. tp * (2 + j) - 1 and prompt rank r pushes a sequence's KV to rank r of the pool the
        sequence decodes on.  ``staged_prompt_cache``: the prompt pool keeps only the blocks of the prompts
        in flight, under block numbers of its own (implied by more than one token pool: their caches are
        separate block spaces); without it a block number names the same block on both pools, as in the
        reference (vllm/worker/comm_utils.py:132-147)."""
        if pipeline_parallel_size > 1:
            raise NotImplementedError("Pipeline parallelism is not supported yet.")
        if token_pools < 1 or (token_pools > 1 and not sep_prompt_token):
            raise ValueError("token_pools > 1 needs sep_prompt_token")
        self.pipeline_parallel_size = pipeline_parallel_size
        self.tensor_parallel_size = tensor_parallel_size
        self.sep_prompt_token = sep_prompt_token
        self.disable_custom_all_reduce = disable_custom_all_reduce
        self.world_size = pipeline_parallel_size * tensor_parallel_size
        self.num_prompt_workers = 0
        self.num_token_workers = 0
        self.num_token_pools = token_pools if sep_prompt_token else 0
        self.one_gpu_overlap = False      # both "pools" are one worker with two streams (executor/overlap_executor.py)
        self.staged_prompt_cache = sep_prompt_token and (staged_prompt_cache or token_pools > 1)
        if sep_prompt_token:
            self.num_prompt_workers = self.world_size
            self.num_token_workers = self.world_size * token_pools
            self.world_size = self.num_prompt_workers + self.num_token_workers


class SchedulerConfig:
    """vllm/config.py:474-520."""

    def __init__(self, max_num_batched_tokens: Optional[int], max_num_seqs: int, max_model_len: int,
                 max_paddings: int = 256, prefill_batch_tokens: int = 0, prefill_max_wait: float = 0.1,
                 pipeline_steps: bool = False) -> None:
        # prefill batching (0 = off: prompts are admitted as soon as they fit, as in the reference)
        self.prefill_batch_tokens = prefill_batch_tokens
        self.pipeline_steps = pipeline_steps
        self.prefill_max_wait = prefill_max_wait
        if max_num_batched_tokens is None:
            max_num_batched_tokens = max(max_model_len, 2048)
        self.max_num_batched_tokens = max_num_batched_tokens
        self.max_num_seqs = max_num_seqs
        self.max_model_len = max_model_len
        self.max_paddings = max_paddings
        if self.max_num_batched_tokens < self.max_model_len:
            raise ValueError(f"max_num_batched_tokens ({self.max_num_batched_tokens}) is smaller than "
                             f"max_model_len ({self.max_model_len}).")
        if self.max_num_batched_tokens < self.max_num_seqs:
            raise ValueError(f"max_num_batched_tokens ({self.max_num_batched_tokens}) must be greater than or "
                             f"equal to max_num_seqs ({self.max_num_seqs}).")


class DeviceConfig:
    def __init__(self, device: str = "cuda") -> None:
        self.device = torch.device(device)
