#!/usr/bin/env python3
"""Dense bf16 GEMM time of the library at the row counts a prompt step can have (7B matrices): how much a
row count just above a multiple of the library's 256-row macro tile costs."""
import sys
sys.path[:0] = ["/root/repo", "/root/repo/vllm-splitwise_amd", "/root/repo/tools"]
import torch
from bench_kernels import timeit

def main():
    dev = "cuda"
    shapes = [(4096, 12288), (4096, 4096), (4096, 22016), (11008, 4096)]
    ws = [torch.randn(n, k, dtype=torch.bfloat16, device=dev) * 0.02 for k, n in shapes]
    for m in (1544, 2304, 2316, 2560, 3072, 3088, 3860, 4096):
        total, parts = 0.0, []
        for (k, n), w in zip(shapes, ws):
            x = torch.randn(m, k, dtype=torch.bfloat16, device=dev)
            t = timeit(lambda: torch.nn.functional.linear(x, w), iters=20)
            total += t
            parts.append(f"{t*1e6:7.1f}")
        flops = sum(2.0 * m * k * n for k, n in shapes)
        print(f"M={m:5d}: {' '.join(parts)} us  sum {total*1e6:8.1f} us  {flops/total/1e12:7.1f} TFLOP/s  {total*1e9/m:6.1f} ns/row", flush=True)

if __name__ == "__main__":
    main()
