import os, sys
ROOT="/root/repo"
for p in (ROOT, ROOT+"/vllm-splitwise_amd", ROOT+"/tools"): sys.path.insert(0,p)
import torch
from splitwise_amd import _C
from bench_elementwise import graph_time
torch.manual_seed(0)
dev, group = "cuda", 128
none = torch.empty((1, 1), device="meta")
for (k, n) in [(5120, 7680), (2560, 5120), (5120, 13824), (6912, 5120)]:
    scales = (torch.rand(k // group, n, device=dev) * 2e-3 + 1e-4).half()
    qz = torch.randint(-2**31, 2**31 - 1, (k // group, n // 8), dtype=torch.int32, device=dev)
    qw = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device=dev)
    for m in (192, 256):
        a = torch.randn(m, k, device=dev).half()
        t = graph_time(lambda: _C.ops.gptq_gemm(a, qw, qz, scales, none, True, 4), reps=32)
        print(f"{os.environ.get('SW_QUANT_KSPLIT','auto'):>4s} K={k} N={n} M={m:3d}: {t:6.1f} us {2.0*m*n*k/t/1e6:6.1f} TFLOP/s", flush=True)
