#!/usr/bin/env python3
"""What a plain streaming read reaches on this GPU, to put the decode attention kernel's GB/s beside it: reductions
over a 4 GiB buffer (nothing of it stays in the 256 MiB memory-side cache between launches)."""
import torch

n = 4 << 30
buf = torch.empty(n, dtype=torch.uint8, device="cuda")
buf.zero_()
views = {"f32 sum": buf.view(torch.float32), "i64 sum": buf.view(torch.int64), "bf16 amax": buf.view(torch.bfloat16),
         "i32 max": buf.view(torch.int32), "f32 sum rows": buf.view(torch.float32).view(1 << 15, -1)}
ops = {"f32 sum": lambda v: v.sum(), "i64 sum": lambda v: v.sum(), "bf16 amax": lambda v: v.amax(),
       "i32 max": lambda v: v.max(), "f32 sum rows": lambda v: v.sum(dim=1)}
for name, v in views.items():
    for _ in range(2):
        ops[name](v)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        ops[name](v)
    e.record()
    torch.cuda.synchronize()
    t = s.elapsed_time(e) / 5 * 1e-3
    print(f"{name:14s} {t * 1e6:8.1f} us  {n / t / 1e9:7.0f} GB/s ({n / t / 8e12:4.2f} of 8 TB/s)", flush=True)
