"""Attention restatements on CPU (torch, fp32 arithmetic).

paged_attention  csrc/attention/attention_kernels.cu:83-417: out[s,h] =
    softmax(scale * q.K^T + alibi) . V over the sequence's paged context, GQA by
    kv_head = h // (H / KVH) (:133-134); logits and accumulation in fp32, the
    probabilities rounded to the activation dtype before P.V (:320-321), 1e-6
    added to the denominator (:273).  The reference's own test reference is
    tests/kernels/test_attention.py:58-111 (tolerance :262-269).
prefill_attention  causal self-attention of the prompt run,
    vllm/model_executor/layers/attention/backends/xformers.py:236-261
    (_ref_masked_attention) == tests/kernels/test_attention.py:272-302.
"""
from typing import List, Optional

import torch


def _dequant_cache(cache: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    if cache.dtype == torch.uint8:
        from . import fp8
        return fp8.from_e5m2(cache, dtype)
    return cache


def gather_kv(key_cache: torch.Tensor, value_cache: torch.Tensor, block_table: List[int],
              context_len: int):
    """-> keys [ctx, KVH, D], values [ctx, KVH, D] in cache order."""
    nb, kvh, dx, bs, x = key_cache.shape
    head = dx * x
    nblk = (context_len + bs - 1) // bs
    idx = torch.as_tensor(block_table[:nblk], dtype=torch.long)
    k = key_cache[idx]                                   # [nblk, KVH, D/x, BS, x]
    k = k.permute(0, 3, 1, 2, 4).reshape(nblk * bs, kvh, head)[:context_len]
    v = value_cache[idx]                                 # [nblk, KVH, D, BS]
    v = v.permute(0, 3, 1, 2).reshape(nblk * bs, kvh, head)[:context_len]
    return k, v


def paged_attention(query: torch.Tensor, key_cache: torch.Tensor, value_cache: torch.Tensor,
                    block_tables: torch.Tensor, context_lens: torch.Tensor, scale: float,
                    alibi_slopes: Optional[torch.Tensor] = None) -> torch.Tensor:
    """query [S, H, D] (f16 / bf16 / f32); caches in query dtype or uint8 e5m2 codes."""
    num_seqs, num_heads, head = query.shape
    kvh = value_cache.shape[1]
    g = num_heads // kvh
    dtype = query.dtype
    key_cache = _dequant_cache(key_cache, dtype)
    value_cache = _dequant_cache(value_cache, dtype)
    out = torch.empty_like(query)
    tables = block_tables.tolist()
    lens = context_lens.tolist()
    for i in range(num_seqs):
        ctx = int(lens[i])
        if ctx == 0:
            out[i].zero_()
            continue
        k, v = gather_kv(key_cache, value_cache, tables[i], ctx)
        k = k.float().repeat_interleave(g, dim=1)        # [ctx, H, D]
        v = v.float().repeat_interleave(g, dim=1)
        logits = scale * torch.einsum("hd,khd->hk", query[i].float(), k)
        if alibi_slopes is not None:
            pos = torch.arange(ctx, dtype=torch.float32) - ctx + 1
            logits = logits + alibi_slopes.float().view(-1, 1) * pos.view(1, -1)
        m = logits.max(dim=-1, keepdim=True).values
        e = torch.exp(logits - m)
        p = e / (e.sum(dim=-1, keepdim=True) + 1e-6)
        p = p.to(dtype).float()                          # rounded like the kernel's logits_vec
        out[i] = torch.einsum("hk,khd->hd", p, v).to(dtype)
    return out


def prefill_attention(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                      seq_lens: List[int], scale: float,
                      alibi_slopes: Optional[torch.Tensor] = None,
                      sliding_window: Optional[int] = None) -> torch.Tensor:
    """Packed varlen causal attention: query [T, H, D], key/value [T, KVH, D],
    tokens of sequence i occupy rows sum(seq_lens[:i]) ... ; returns [T, H, D].
    With a sliding window a query sees the keys at distance 0 .. window - 1 behind it (the local
    attention bias of backends/xformers.py:119-121)."""
    num_heads, kvh = query.shape[1], key.shape[1]
    g = num_heads // kvh
    dtype = query.dtype
    out = torch.empty_like(query)
    start = 0
    for n in seq_lens:
        q = query[start:start + n].float()
        k = key[start:start + n].float().repeat_interleave(g, dim=1)
        v = value[start:start + n].float().repeat_interleave(g, dim=1)
        w = scale * torch.einsum("qhd,khd->hqk", q, k)
        if alibi_slopes is not None:
            pos = torch.arange(n, dtype=torch.float32)
            w = w + alibi_slopes.float().view(-1, 1, 1) * (pos.view(1, 1, -1) - pos.view(1, -1, 1))
        mask = torch.triu(torch.ones(n, n, dtype=torch.bool), diagonal=1)
        if sliding_window is not None:
            mask |= torch.tril(torch.ones(n, n, dtype=torch.bool), diagonal=-sliding_window)
        w = w.masked_fill(mask, float("-inf"))
        p = torch.softmax(w, dim=-1).to(dtype).float()
        out[start:start + n] = torch.einsum("hqk,khd->qhd", p, v).to(dtype)
        start += n
    return out


def context_attention(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, key_cache: torch.Tensor,
                      value_cache: torch.Tensor, block_tables: torch.Tensor, new_lens: List[int],
                      ctx_lens: List[int], alibi_slopes: Optional[torch.Tensor] = None,
                      sliding_window: Optional[int] = None) -> torch.Tensor:
    """Prompt attention over a cached prefix: what ``context_attention_fwd`` computes
    (vllm/model_executor/layers/attention/ops/prefix_prefill.py:10-745) and what its test compares
    with (tests/kernels/test_prefix_prefill.py:126-176: causal "from bottom right" attention of the
    new tokens over context + new tokens, scale D^-1/2).  query [T, H, D] and key/value [T, KVH, D]
    hold the NEW tokens of all sequences back to back; the context lives in the paged caches.
    ALiBi bias = slope * (key position - query position), as in _fwd_kernel_alibi (:433-618).
    A sliding window (not in the reference's prefix kernel; the same rule as in prefill_attention) lets a
    query see the keys at distance 0 .. window - 1 behind it, cached or new."""
    num_heads, kvh, head = query.shape[1], key.shape[1], query.shape[2]
    g = num_heads // kvh
    dtype = query.dtype
    scale = head ** -0.5
    out = torch.empty_like(query)
    tables = block_tables.tolist()
    start = 0
    for i, (n, ctx) in enumerate(zip(new_lens, ctx_lens)):
        kc, vc = gather_kv(key_cache, value_cache, tables[i], ctx) if ctx > 0 else (key[:0], value[:0])
        k = torch.cat((kc, key[start:start + n])).float().repeat_interleave(g, dim=1)      # [ctx + n, H, D]
        v = torch.cat((vc, value[start:start + n])).float().repeat_interleave(g, dim=1)
        q = query[start:start + n].float()
        w = scale * torch.einsum("qhd,khd->hqk", q, k)
        qpos = torch.arange(n, dtype=torch.float32) + ctx
        kpos = torch.arange(ctx + n, dtype=torch.float32)
        if alibi_slopes is not None:
            w = w + alibi_slopes.float().view(-1, 1, 1) * (kpos.view(1, 1, -1) - qpos.view(1, -1, 1))
        mask = kpos.view(1, -1) > qpos.view(-1, 1)
        if sliding_window is not None:
            mask = mask | (qpos.view(-1, 1) - kpos.view(1, -1) >= sliding_window)
        w = w.masked_fill(mask.unsqueeze(0), float("-inf"))
        p = torch.softmax(w, dim=-1).to(dtype).float()
        out[start:start + n] = torch.einsum("hqk,khd->qhd", p, v).to(dtype)
        start += n
    return out
