#!/usr/bin/env python3
"""v1 against v2 on small MHA decode launches (a handful of sequences, 500-1500 tokens): the band a ramping
server passes through.  Prints both times and the algorithmic GB/s of the better one."""
import contextlib
import io
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402
from bench_kernels import bench_paged  # noqa: E402

for s in (2, 3, 4, 5, 6, 7, 8, 12, 16):
    for ctx in (520, 640, 768, 800, 900, 1024, 1280, 1536):
        with contextlib.redirect_stdout(io.StringIO()):
            t1 = bench_paged(s, ctx, 32, 32, dtype=torch.bfloat16, version="v1")
            t2 = bench_paged(s, ctx, 32, 32, dtype=torch.bfloat16, version="v2")
        alg = s * 2 * ctx * 32 * 128 * 2
        print(f"mha32 bf16 S={s:3d} wg={s * 32:4d} ctx={ctx:5d} v1={t1 * 1e6:6.1f} v2={t2 * 1e6:6.1f} "
              f"best={'v1' if t1 <= t2 else 'v2'} {alg / min(t1, t2) / 1e9:6.0f} GB/s", flush=True)
