"""RMSNorm, rotary embedding and gated activation layers backed by the HIP operators.

Each layer keeps the reference's two entry points: ``forward`` (native kernel) and
``_forward`` (plain PyTorch, the reference's own test oracle for the kernel):
vllm/model_executor/layers/layernorm.py:10-65, rotary_embedding.py:36-141,
activation.py:15-44.
"""
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import _C


class RMSNorm(nn.Module):
    def __init__(self, hidden_size: int, eps: float = 1e-6) -> None:
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def _forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None):
        orig_dtype = x.dtype
        x = x.to(torch.float32)
        if residual is not None:
            x = x + residual.to(torch.float32)
            residual = x.to(orig_dtype)
        variance = x.pow(2).mean(dim=-1, keepdim=True)
        x = x * torch.rsqrt(variance + self.variance_epsilon)
        x = x.to(orig_dtype) * self.weight
        return x if residual is None else (x, residual)

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None
                ) -> Union[torch.Tensor, Tuple[torch.Tensor, torch.Tensor]]:
        if residual is not None:
            _C.ops.fused_add_rms_norm(x, residual, self.weight.data, self.variance_epsilon)
            return x, residual
        out = torch.empty_like(x)
        _C.ops.rms_norm(out, x, self.weight.data, self.variance_epsilon)
        return out


class RotaryEmbedding(nn.Module):
    """cos / sin table per position, applied in place to q and k."""

    def __init__(self, head_size: int, rotary_dim: int, max_position_embeddings: int, base: float,
                 is_neox_style: bool, dtype: torch.dtype) -> None:
        super().__init__()
        self.head_size = head_size
        self.rotary_dim = rotary_dim
        self.is_neox_style = is_neox_style
        inv_freq = 1.0 / (base ** (torch.arange(0, rotary_dim, 2, dtype=torch.float) / rotary_dim))
        t = torch.arange(max_position_embeddings, dtype=torch.float)
        freqs = torch.einsum("i,j -> ij", t, inv_freq)
        cache = torch.cat((freqs.cos(), freqs.sin()), dim=-1).to(dtype)      # [max_pos, rot_dim]
        self.register_buffer("cos_sin_cache", cache, persistent=False)

    def _forward(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor):
        """PyTorch restatement (rotary_embedding.py:88-122), fp32 products rounded once."""
        shape_q, shape_k = query.shape, key.shape
        cos_sin = self.cos_sin_cache[positions.flatten()].float()
        cos, sin = cos_sin.chunk(2, dim=-1)
        cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)

        def rot(x):
            x = x.reshape(cos.shape[0], -1, self.head_size).float()
            xr, xp = x[..., :self.rotary_dim], x[..., self.rotary_dim:]
            if self.is_neox_style:
                x1, x2 = xr.chunk(2, dim=-1)
                o = torch.cat((x1 * cos - x2 * sin, x2 * cos + x1 * sin), dim=-1)
            else:
                x1, x2 = xr[..., ::2], xr[..., 1::2]
                o = torch.stack((x1 * cos - x2 * sin, x2 * cos + x1 * sin), dim=-1).flatten(-2)
            return torch.cat((o, xp), dim=-1)

        return rot(query).to(query.dtype).reshape(shape_q), rot(key).to(key.dtype).reshape(shape_k)

    def forward(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor):
        _C.ops.rotary_embedding(positions.flatten(), query, key, self.head_size, self.cos_sin_cache,
                                self.is_neox_style)
        return query, key

    def fusable_with_cache_write(self, dtype: torch.dtype) -> bool:
        """Whether ``ops.rotary_embedding_and_cache`` takes this geometry: the rotated part (its halves for
        the neox pairing), the rest of the head and the head itself span whole 16-byte vectors."""
        n = 16 // torch.empty((), dtype=dtype).element_size()
        rot = self.rotary_dim // 2 if self.is_neox_style else self.rotary_dim
        return (self.rotary_dim % 2 == 0 and rot % n == 0 and (self.head_size - self.rotary_dim) % n == 0
                and self.head_size % n == 0)


class SiluAndMul(nn.Module):
    def _forward(self, x: torch.Tensor) -> torch.Tensor:
        d = x.shape[-1] // 2
        return F.silu(x[..., :d]) * x[..., d:]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        d = x.shape[-1] // 2
        out = torch.empty(x.shape[:-1] + (d,), dtype=x.dtype, device=x.device)
        _C.ops.silu_and_mul(out, x)
        return out
