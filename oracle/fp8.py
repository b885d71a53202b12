"""Unscaled fp8-e5m2 <-> {f16, bf16, f32}, bit-exact restatement.

Follows ``csrc/quantization/fp8_e5m2_kvcache/quant_utils.cuh:23-29,73-81,118-126``
(fp8 -> wider: exact, e5m2 is the high byte of an IEEE half) and ``:164-192``
(wider -> fp8 via ``__nv_cvt_{halfraw,bfloat16raw,float}_to_fp8(x, __NV_SATFINITE,
__NV_E5M2)``: ONE round-to-nearest-even step from the source precision, results
beyond the largest finite value 57344 (0x7B) -- including +-Inf -- saturate to it,
NaN maps to sign|0x7F).
"""
import numpy as np
import torch


def f16_bits_to_e5m2(h: np.ndarray) -> np.ndarray:
    h = h.astype(np.uint32)
    sign = (h >> 8) & 0x80
    a = h & 0x7FFF
    r = (a + 0x7F + ((a >> 8) & 1)) >> 8
    r = np.where(r >= 0x7C, 0x7B, r)
    r = np.where(a > 0x7C00, 0x7F, r)
    return (sign | r).astype(np.uint8)


def f32_to_e5m2(f: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(f, dtype=np.float32).view(np.uint32)
    sign = (u >> 24) & 0x80
    a = u & 0x7FFFFFFF
    # normal e5m2 results: rebias (127 -> 15) and round off 21 mantissa bits
    t = (a.astype(np.int64) - 0x38000000)
    rn = (t + 0xFFFFF + ((t >> 21) & 1)) >> 21
    rn = np.where(rn >= 0x7C, 0x7B, rn)
    # subnormal e5m2 results: multiples of 2^-16, round half to even
    with np.errstate(invalid="ignore"):                       # NaN lanes are overwritten below
        rs = np.rint(np.minimum(a, 0x38800000).view(np.float32).astype(np.float64) * 65536.0).astype(np.int64)
    r = np.where(a >= 0x38800000, rn, rs)
    r = np.where(a > 0x7F800000, 0x7F, r)
    return (sign | r.astype(np.uint32)).astype(np.uint8)


def to_e5m2(t: torch.Tensor) -> torch.Tensor:
    """torch tensor (f16 / bf16 / f32) -> uint8 e5m2 codes, same shape."""
    if t.dtype == torch.float16:
        out = f16_bits_to_e5m2(t.contiguous().view(torch.int16).numpy().view(np.uint16))
    elif t.dtype in (torch.bfloat16, torch.float32):
        out = f32_to_e5m2(t.float().contiguous().numpy())   # bf16 -> f32 is exact
    else:
        raise ValueError(f"unsupported dtype {t.dtype}")
    return torch.from_numpy(out.reshape(tuple(t.shape)))


def from_e5m2(codes: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """uint8 e5m2 codes -> dtype (exact for f16, bf16 and f32)."""
    h = torch.from_numpy((codes.contiguous().numpy().astype(np.uint16) << 8).view(np.int16))
    h = h.view(torch.float16).reshape(tuple(codes.shape))
    return h if dtype == torch.float16 else h.float().to(dtype)
