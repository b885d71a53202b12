import sys, os, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from bench_kernels import bench_paged
cases = [("mha32 bf16", 32, 32, torch.bfloat16, "auto"), ("gqa16/2 bf16", 16, 2, torch.bfloat16, "auto"),
         ("gqa64/8 f16", 64, 8, torch.float16, "auto"), ("mha20 f16 fp8", 20, 20, torch.float16, "fp8_e5m2"),
         ("gqa32/8 bf16", 32, 8, torch.bfloat16, "auto"), ("gqa16/2 bf16 fp8", 16, 2, torch.bfloat16, "fp8_e5m2")]
if len(sys.argv) > 1:
    cases = [c for c in cases if any(a in c[0] for a in sys.argv[1:])]
for name, h, kvh, dt, kv in cases:
    for s in (1, 4, 8, 16, 32, 64, 128, 256):
        for ctx in (768, 1024, 2048, 4096, 16384):
            if s * ctx > 300000:
                continue
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                t1 = bench_paged(s, ctx, h, kvh, dtype=dt, version="v1", kv=kv)
                t2 = bench_paged(s, ctx, h, kvh, dtype=dt, version="v2", kv=kv)
            print(f"{name:14s} S={s:4d} ctx={ctx:6d} v1={t1*1e6:8.1f} v2={t2*1e6:8.1f} best={'v1' if t1 <= t2 else 'v2'} ratio={t1/t2:5.2f}", flush=True)
