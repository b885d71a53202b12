"""KV-cache maintenance operators on CPU tensors (bit-exact restatements).

reshape_and_cache  csrc/cache_kernels.cu:154-268 (test reference tests/kernels/test_cache.py:148-161)
copy_blocks        csrc/cache_kernels.cu:64-150  (tests/kernels/test_cache.py:85-98)
swap_blocks        csrc/cache_kernels.cu:21-60   (tests/kernels/test_cache.py:222-226)
convert_fp8_e5m2   csrc/cache_kernels.cu:272-320
"""
from typing import Dict, List

import torch

from . import fp8


def reshape_and_cache(key: torch.Tensor, value: torch.Tensor, key_cache: torch.Tensor,
                      value_cache: torch.Tensor, slot_mapping: torch.Tensor,
                      kv_cache_dtype: str = "auto") -> None:
    """key/value [T, H, D]; key_cache [NB, H, D/x, BS, x]; value_cache [NB, H, D, BS];
    slot s -> block s // BS, offset s % BS; negative slots are skipped."""
    num_tokens, num_heads, head_size = key.shape
    block_size, x = key_cache.shape[3], key_cache.shape[4]
    if kv_cache_dtype == "fp8_e5m2":
        key, value = fp8.to_e5m2(key), fp8.to_e5m2(value)
    elif kv_cache_dtype != "auto":
        raise RuntimeError(f"Unsupported data type of kv cache: {kv_cache_dtype}")
    slots = slot_mapping.tolist()
    k5 = key.reshape(num_tokens, num_heads, head_size // x, x)
    for t, s in enumerate(slots):
        if s < 0:
            continue
        b, o = divmod(s, block_size)
        key_cache[b, :, :, o, :] = k5[t]
        value_cache[b, :, :, o] = value[t]


def copy_blocks(key_caches: List[torch.Tensor], value_caches: List[torch.Tensor],
                block_mapping: Dict[int, List[int]]) -> None:
    for src, dsts in block_mapping.items():
        for dst in dsts:
            for kc in key_caches:
                kc[dst].copy_(kc[src])
            for vc in value_caches:
                vc[dst].copy_(vc[src])


def swap_blocks(src: torch.Tensor, dst: torch.Tensor, block_mapping: Dict[int, int]) -> None:
    for s, d in block_mapping.items():
        dst[d].copy_(src[s])


def convert_fp8_e5m2(src: torch.Tensor, dst: torch.Tensor) -> None:
    if src.dtype == torch.uint8:
        dst.copy_(fp8.from_e5m2(src, dst.dtype))
    else:
        dst.copy_(fp8.to_e5m2(src))
