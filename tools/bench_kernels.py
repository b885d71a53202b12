#!/usr/bin/env python3
"""Kernel micro-benchmarks with HIP-event timing: achieved algorithmic GB/s per kernel.

  python tools/bench_kernels.py [paged|reshape|copy|quant|all]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-splitwise_amd"))
from splitwise_amd import _C  # noqa: E402


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        fn()
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) / iters * 1e-3


def bench_paged(num_seqs, ctx, h, kvh, d=128, bs=16, dtype=torch.bfloat16, version="v1", kv="auto"):
    dev = "cuda"
    nblk_seq = (ctx + bs - 1) // bs
    nb = num_seqs * nblk_seq + 1
    store = torch.uint8 if kv == "fp8_e5m2" else dtype
    es = torch.tensor([], dtype=store).element_size()
    x = 16 // es
    kc = torch.empty(nb, kvh, d // x, bs, x, dtype=torch.float16 if store == torch.uint8 else dtype,
                     device=dev).uniform_(-0.1, 0.1)
    vc = torch.empty(nb, kvh, d, bs, dtype=kc.dtype, device=dev).uniform_(-0.1, 0.1)
    if store == torch.uint8:
        kc = kc.view(torch.uint8)[..., ::2].contiguous().view(nb, kvh, d // x, bs, x)
        vc = vc.view(torch.uint8)[..., ::2].contiguous()
    perm = torch.randperm(nb - 1, device=dev).to(torch.int32)
    bt = perm.view(num_seqs, nblk_seq).contiguous()
    cl = torch.full((num_seqs,), ctx, dtype=torch.int32, device=dev)
    q = torch.empty(num_seqs, h, d, dtype=dtype, device=dev).uniform_(-0.1, 0.1)
    out = torch.empty_like(q)
    scale = d ** -0.5
    if version == "v1":
        fn = lambda: _C.ops.paged_attention_v1(out, q, kc, vc, kvh, scale, bt, cl, bs, ctx, None, kv)
    else:
        parts = (ctx + 511) // 512
        tmp = torch.empty(num_seqs, h, parts, d, dtype=dtype, device=dev)
        esum = torch.empty(num_seqs, h, parts, dtype=torch.float32, device=dev)
        ml = torch.empty_like(esum)
        fn = lambda: _C.ops.paged_attention_v2(out, esum, ml, tmp, q, kc, vc, kvh, scale, bt, cl, bs, ctx,
                                               None, kv)
    t = timeit(fn)
    kv_bytes = num_seqs * 2 * ctx * kvh * d * es
    io_bytes = kv_bytes + 2 * num_seqs * h * d * q.element_size() + bt.numel() * 4
    print(f"paged_{version} S={num_seqs} ctx={ctx} H={h} KVH={kvh} D={d} bs={bs} {str(dtype)[6:]} kv={kv}: "
          f"{t * 1e6:8.1f} us  {io_bytes / t / 1e9:8.1f} GB/s  ({io_bytes / 1e6:.1f} MB)")
    return t


def bench_reshape(num_tokens, kvh, d=128, bs=16, dtype=torch.bfloat16, consecutive=True):
    dev = "cuda"
    nb = (num_tokens + bs - 1) // bs + 8
    qkv = torch.randn(num_tokens, 3 * kvh * d, dtype=dtype, device=dev)
    k = qkv[:, kvh * d:2 * kvh * d].view(num_tokens, kvh, d)
    v = qkv[:, 2 * kvh * d:].view(num_tokens, kvh, d)
    x = 16 // qkv.element_size()
    kc = torch.zeros(nb, kvh, d // x, bs, x, dtype=dtype, device=dev)
    vc = torch.zeros(nb, kvh, d, bs, dtype=dtype, device=dev)
    slots = torch.arange(num_tokens, device=dev) if consecutive else torch.randperm(nb * bs, device=dev)[:num_tokens]
    slots = slots.to(torch.long)
    t = timeit(lambda: _C.cache_ops.reshape_and_cache(k, v, kc, vc, slots, "auto"))
    nbytes = num_tokens * 2 * kvh * d * qkv.element_size() * 2 + 8 * num_tokens
    print(f"reshape_and_cache T={num_tokens} KVH={kvh} D={d} consecutive={consecutive}: {t * 1e6:8.1f} us  "
          f"{nbytes / t / 1e9:8.1f} GB/s")


def bench_copy(pairs, layers=32, kvh=32, d=128, bs=16, dtype=torch.bfloat16):
    dev = "cuda"
    nb = 2 * pairs + 2
    x = 8
    kcs = [torch.randn(nb, kvh, d // x, bs, x, dtype=dtype, device=dev) for _ in range(layers)]
    vcs = [torch.randn(nb, kvh, d, bs, dtype=dtype, device=dev) for _ in range(layers)]
    mapping = {i: [pairs + i] for i in range(pairs)}
    t = timeit(lambda: _C.cache_ops.copy_blocks(kcs, vcs, mapping), iters=10)
    nbytes = pairs * layers * 2 * kvh * d * bs * 2 * 2
    print(f"copy_blocks pairs={pairs} layers={layers}: {t * 1e6:8.1f} us  {nbytes / t / 1e9:8.1f} GB/s")


def bench_quant(m, k, n, fmt="gptq", group=128):
    """One of the 13B / TP2 matrices: random int4 words, the stated group size."""
    dev = "cuda"
    a = torch.randn(m, k, dtype=torch.float16, device=dev)
    scales = (torch.rand(k // group, n, device=dev) * 2e-3).to(torch.float16)
    if fmt == "gptq":
        qw = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device=dev)
        qz = torch.randint(-2**31, 2**31 - 1, (k // group, n // 8), dtype=torch.int32, device=dev)
        fn = lambda: _C.ops.gptq_gemm(a, qw, qz, scales, None, True, 4)
        dq = lambda: _C.ops.dequant_gptq(qw, qz, scales, None, 4, True)
    else:
        qw = torch.randint(-2**31, 2**31 - 1, (k, n // 8), dtype=torch.int32, device=dev)
        qz = torch.randint(-2**31, 2**31 - 1, (k // group, n // 8), dtype=torch.int32, device=dev)
        fn = lambda: _C.ops.awq_gemm(a, qw, scales, qz, 8)
        dq = lambda: _C.ops.awq_dequantize(qw, scales, qz, 0, 0, 0)
    t = timeit(fn, iters=50)
    nbytes = k * n // 2 + (k // group) * n * 2 + (k // group) * n // 2 + m * k * 2 + m * n * 2
    flops = 2.0 * m * n * k
    w = dq()
    t_dense = timeit(lambda: torch.matmul(a, w), iters=50)
    print(f"{fmt}_gemm M={m} K={k} N={n} g={group}: {t * 1e6:8.1f} us  {nbytes / t / 1e9:8.1f} GB/s  "
          f"{flops / t / 1e12:7.1f} TFLOP/s   (dense fp16 matmul on the reconstructed weight: {t_dense * 1e6:8.1f} us)")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("paged", "all"):
        for (s, ctx) in [(64, 1024), (256, 1024), (8, 4096), (64, 4096), (1, 1024), (1, 16384)]:
            bench_paged(s, ctx, 32, 32)                       # Llama-2-7B
        for v in ("v2",):
            for (s, ctx) in [(64, 1024), (8, 4096), (1, 16384)]:
                bench_paged(s, ctx, 32, 32, version=v)
        for (s, ctx) in [(64, 1024), (64, 4096), (8, 4096)]:
            bench_paged(s, ctx, 16, 2)                        # Llama-2-70B / TP4
        bench_paged(8, 4096, 64, 8, dtype=torch.float16)      # the reference kernel bench default
        bench_paged(64, 1024, 20, 20, dtype=torch.float16, kv="fp8_e5m2")   # 13B / TP2, fp8 cache
        bench_paged(64, 1024, 32, 32, bs=32)
    if what in ("reshape", "all"):
        bench_reshape(64, 32, consecutive=False)
        bench_reshape(512, 32)
        bench_reshape(2048, 32)
        bench_reshape(49408, 32)
        bench_reshape(49408, 32, consecutive=False)
    if what == "awq":
        for m in (1, 8, 32, 128):
            for (k, n) in [(5120, 13824), (6912, 5120)]:
                bench_quant(m, k, n, fmt="awq")
    if what in ("quant", "all"):
        shapes = [(5120, 7680), (2560, 5120), (5120, 13824), (6912, 5120)]      # qkv, o, gate_up, down of 13B / TP2
        for m in (1, 8, 32, 64, 128, 256):
            for (k, n) in shapes:
                bench_quant(m, k, n)
        for m in (1, 8, 32, 128):
            for (k, n) in shapes[2:]:
                bench_quant(m, k, n, fmt="awq")
    if what in ("copy", "all"):
        bench_copy(8)
        bench_copy(64)
