import sys, os
sys.path.insert(0, "tools")
import torch
from bench_kernels import bench_paged
for s, ctx in ((64, 2048), (256, 1024), (256, 2048)):
    for dt in (torch.bfloat16, torch.float16):
        bench_paged(s, ctx, 32, 32, dtype=dt, version="v1", kv="fp8_e5m2")
    bench_paged(s, ctx, 20, 20, dtype=torch.float16, version="v1", kv="fp8_e5m2")
