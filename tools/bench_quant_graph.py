#!/usr/bin/env python3
"""int4 GEMM durations measured inside a replayed HIP graph (32 back-to-back launches), which
leaves the host's per-call cost (ctypes, workspace lookup, output allocation: ~12 us) out of the
figure -- the way the kernels run in a captured decode step.  13B / TP2 matrices."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "vllm-splitwise_amd"), os.path.join(ROOT, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import torch  # noqa: E402
from splitwise_amd import _C  # noqa: E402
from bench_elementwise import graph_time  # noqa: E402


def main():
    torch.manual_seed(0)
    dev, group = "cuda", 128
    none = torch.empty((1, 1), device="meta")
    for (k, n) in [(5120, 7680), (2560, 5120), (5120, 13824), (6912, 5120)]:
        scales = (torch.rand(k // group, n, device=dev) * 2e-3 + 1e-4).half()
        qz = torch.randint(-2**31, 2**31 - 1, (k // group, n // 8), dtype=torch.int32, device=dev)
        qw = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device=dev)
        qwa = torch.randint(-2**31, 2**31 - 1, (k, n // 8), dtype=torch.int32, device=dev)
        for m in (1, 8, 32, 48, 64, 96, 128, 256):
            a = torch.randn(m, k, device=dev).half()
            nbytes = k * n // 2 + (k // group) * n * 2.5 + m * k * 2 + m * n * 2
            t = graph_time(lambda: _C.ops.gptq_gemm(a, qw, qz, scales, none, True, 4), reps=32)
            line = f"K={k} N={n} M={m:3d}: gptq {t:6.1f} us {nbytes / t / 1e3:7.1f} GB/s {2.0 * m * n * k / t / 1e6:6.1f} TFLOP/s"
            if m <= 128:
                t2 = graph_time(lambda: _C.ops.awq_gemm(a, qwa, scales, qz, 8), reps=32)
                line += f" | awq (checkpoint layout) {t2:6.1f} us"
            print(line, flush=True)


if __name__ == "__main__":
    main()
