#!/usr/bin/env python3
"""Decode-step GEMMs at a few row counts with weights that do not sit in the memory-side cache (16 layers' matrices
in turn): the library on the checkpoint's layout (W [N, K], y = x W^T) against the same product on a transposed copy
(W^T [K, N] contiguous, y = x @ W^T), and against the swapped product (y^T = W x^T).

  python tools/bench_gemm_layouts.py [M ...]
"""
import sys

import torch
import torch.nn.functional as F

LAYERS = 16


def cold_time(fn, ws, iters=6):
    for w in ws:
        fn(w)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        for w in ws:
            fn(w)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (iters * len(ws)) * 1e3


def main():
    ms = [int(a) for a in sys.argv[1:]] or [256]
    shapes = {"qkv": (12288, 4096), "o": (4096, 4096), "gate_up": (22016, 4096), "down": (4096, 11008)}
    for name, (n, k) in shapes.items():
        ws = [torch.randn(n, k, dtype=torch.bfloat16, device="cuda") * 0.02 for _ in range(LAYERS)]
        wts = [w.t().contiguous() for w in ws]
        for m in ms:
            x = torch.randn(m, k, dtype=torch.bfloat16, device="cuda")
            xt = x.t().contiguous()
            out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
            a = cold_time(lambda w: F.linear(x, w), ws)
            b = cold_time(lambda wt: torch.mm(x, wt, out=out), wts)
            c = cold_time(lambda w: torch.mm(w, xt), ws)
            gb = (n * k + m * k + m * n) * 2 / 1e9
            print(f"{name:8s} M={m:4d}: x W^T {a:7.1f} us ({gb / a * 1e6:5.0f} GB/s)   x @ Wt {b:7.1f} us   "
                  f"W x^T {c:7.1f} us", flush=True)
        del ws, wts


if __name__ == "__main__":
    main()
