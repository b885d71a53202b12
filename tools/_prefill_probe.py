import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-splitwise_amd"))
import torch, torch.nn.functional as F
def timeit(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
shapes = {"qkv": (12288, 4096), "o": (4096, 4096), "gate_up": (22016, 4096), "down": (4096, 11008)}
ws = {k: torch.randn(n, kk, dtype=torch.bfloat16, device="cuda") * 0.02 for k, (n, kk) in shapes.items()}
for m in (772, 1544, 2316, 3088, 3860):
    tot = 0
    for name, (n, k) in shapes.items():
        x = torch.randn(m, k, dtype=torch.bfloat16, device="cuda")
        t = timeit(lambda: F.linear(x, ws[name]))
        tot += t
    flops = 2 * m * sum(n * k for n, k in shapes.values())
    print(f"M={m}: 4 GEMMs {tot:8.1f} us/layer  {flops / tot / 1e6:7.1f} TFLOP/s   per-token {tot / m:6.3f} us")
