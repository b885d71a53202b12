"""CPU restatement (torch, fp32 math) of the reference's row-wise operators.

TEST INFRASTRUCTURE ONLY.  Each function follows the reference layer's own PyTorch-native
``_forward`` -- the implementation its kernel tests compare the CUDA kernels with:
  rms_norm / fused_add_rms_norm   vllm/model_executor/layers/layernorm.py:26-44
  rotary_embedding                vllm/model_executor/layers/rotary_embedding.py:79-92, 94-140
  silu_and_mul                    vllm/model_executor/layers/activation.py:25-28
Pinned by the tolerances of tests/kernels/test_layernorm.py:50-53 (1e-2 / 1e-2) and
tests/kernels/allclose_default.py (fp16 1e-3 / 1e-3, bf16 1e-3 / 1.6e-2) used by
test_activation.py and test_pos_encoding.py; no golden vectors exist for these operators.
"""
from typing import Optional, Tuple

import torch


def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float,
             residual: Optional[torch.Tensor] = None):
    orig = x.dtype
    x = x.to(torch.float32)
    if residual is not None:
        x = x + residual.to(torch.float32)
        residual = x.to(orig)
    variance = x.pow(2).mean(dim=-1, keepdim=True)
    x = x * torch.rsqrt(variance + eps)
    out = x.to(orig) * weight
    return out if residual is None else (out, residual)


def cos_sin_cache(rotary_dim: int, max_position: int, base: float, dtype: torch.dtype) -> torch.Tensor:
    inv_freq = 1.0 / (base ** (torch.arange(0, rotary_dim, 2, dtype=torch.float) / rotary_dim))
    freqs = torch.einsum("i,j -> ij", torch.arange(max_position, dtype=torch.float), inv_freq)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1).to(dtype)


def _rotate_neox(x):
    x1, x2 = x[..., :x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def _rotate_gptj(x):
    x1, x2 = x[..., ::2], x[..., 1::2]
    return torch.stack((-x2, x1), dim=-1).flatten(-2)


def rotary_embedding(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor, head_size: int,
                     cache: torch.Tensor, is_neox: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    rot = cache.shape[-1]
    q = query.view(*query.shape[:-1], -1, head_size)
    k = key.view(*key.shape[:-1], -1, head_size)
    cos, sin = cache[positions].chunk(2, dim=-1)
    if is_neox:
        cos = cos.repeat(1, 2).unsqueeze(-2)
        sin = sin.repeat(1, 2).unsqueeze(-2)
        rotate = _rotate_neox
    else:
        cos = cos.repeat_interleave(2, dim=-1).unsqueeze(-2)
        sin = sin.repeat_interleave(2, dim=-1).unsqueeze(-2)
        rotate = _rotate_gptj

    def apply(t):
        t_rot, t_pass = t[..., :rot], t[..., rot:]
        t_rot = t_rot * cos + rotate(t_rot) * sin
        return torch.cat((t_rot, t_pass), dim=-1).flatten(-2)

    return apply(q), apply(k)


def silu_and_mul(x: torch.Tensor) -> torch.Tensor:
    d = x.shape[-1] // 2
    return torch.nn.functional.silu(x[..., :d]) * x[..., d:]
