#!/usr/bin/env python3
"""Decode attention on grouped-query shapes (the 70B / TP4 rank: 16 query heads over 2 KV heads; 64 over 8),
bf16 / f16 activations over 16-bit and fp8 caches: algorithmic GB/s and fraction of the 8 TB/s HBM peak.

  python tools/bench_gqa_decode.py
"""
import contextlib
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402
from bench_kernels import bench_paged  # noqa: E402

CASES = [("70B/TP4 16/2 bf16", 16, 2, torch.bfloat16, "auto"), ("64/8 bf16", 64, 8, torch.bfloat16, "auto"),
         ("64/8 f16", 64, 8, torch.float16, "auto"), ("16/2 bf16 fp8", 16, 2, torch.bfloat16, "fp8_e5m2"),
         ("64/8 f16 fp8", 64, 8, torch.float16, "fp8_e5m2"), ("7B 32/32 bf16", 32, 32, torch.bfloat16, "auto")]
for name, h, kvh, dt, kv in CASES:
    if os.environ.get("GQA_CASES", "") not in name:
        continue
    for s, ctx in ((64, 2048), (64, 4096), (128, 2048), (256, 2048), (256, 4096), (16, 8192), (8, 1024)):
        if s * ctx * kvh > 256 * 4096 * 8:
            continue
        es = 1 if kv != "auto" else 2
        alg = s * 2 * ctx * kvh * 128 * es + 2 * s * h * 128 * 2 + s * (ctx // 16) * 4
        row = []
        for version in ("v1", "v2"):
            with contextlib.redirect_stdout(io.StringIO()):
                t = bench_paged(s, ctx, h, kvh, dtype=dt, version=version, kv=kv)
            row.append(f"{version} {t * 1e6:7.1f} us {alg / t / 1e9:7.0f} GB/s ({alg / t / 8e12:4.2f})")
        print(f"{name:18s} S={s:4d} ctx={ctx:5d}  " + "   ".join(row), flush=True)
