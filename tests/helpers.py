"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

DTYPES = {"float16": torch.float16, "bfloat16": torch.bfloat16, "float32": torch.float32}


def from_np(a: np.ndarray, dtype: torch.dtype) -> torch.Tensor:
    """Inverse of the fixture writer: bf16 tensors are stored as raw uint16 bits."""
    if dtype == torch.bfloat16:
        return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)
    return torch.from_numpy(a.copy())


def load_golden(name: str):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def make_kv_caches(num_blocks, block_size, num_kv_heads, head_size, dtype, seed=0, device="cpu",
                   cache_dtype="auto"):
    """K [nb, KVH, D/x, bs, x] and V [nb, KVH, D, bs] ~ U(-D^-1/2, D^-1/2) -- the distribution of
    the reference's kernel-test fixture (vllm/utils.py:332-397)."""
    g = torch.Generator().manual_seed(seed)
    scale = head_size ** -0.5
    store = torch.uint8 if cache_dtype == "fp8_e5m2" else dtype
    x = 16 // torch.tensor([], dtype=store).element_size()
    k = (torch.rand(num_blocks, num_kv_heads, head_size // x, block_size, x, generator=g) * 2 - 1) * scale
    v = (torch.rand(num_blocks, num_kv_heads, head_size, block_size, generator=g) * 2 - 1) * scale
    if cache_dtype == "fp8_e5m2":
        from oracle import fp8
        k, v = fp8.to_e5m2(k.to(torch.float16)), fp8.to_e5m2(v.to(torch.float16))
    else:
        k, v = k.to(dtype), v.to(dtype)
    return k.to(device), v.to(device)
