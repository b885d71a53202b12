#!/usr/bin/env python3
"""Decode-step GEMMs with weights that do NOT sit in the memory-side cache (32 layers' matrices in turn, as in a
model step): the BLAS library's default solution against TunableOp's pick when the tuning runs rotate their buffers.

  python tools/tune_gemm_cold.py <out.csv> [M ...]
"""
import os
import sys

out = sys.argv[1]
os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
os.environ["PYTORCH_TUNABLEOP_TUNING"] = "1"
os.environ["PYTORCH_TUNABLEOP_FILENAME"] = out
os.environ.setdefault("PYTORCH_TUNABLEOP_ROTATING_BUFFER_SIZE", "1024")
os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "100")
os.environ.setdefault("PYTORCH_TUNABLEOP_VERBOSE", "0")

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

LAYERS = 16


def cold_time(x, ws, iters=6):
    for w in ws:
        F.linear(x, w)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        for w in ws:
            F.linear(x, w)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / (iters * len(ws)) * 1e3


def main():
    ms = [int(a) for a in sys.argv[2:]] or [256]
    shapes = {"qkv": (12288, 4096), "o": (4096, 4096), "gate_up": (22016, 4096), "down": (4096, 11008)}
    tun = torch.cuda.tunable
    for name, (n, k) in shapes.items():
        ws = [torch.randn(n, k, dtype=torch.bfloat16, device="cuda") * 0.02 for _ in range(LAYERS)]
        for m in ms:
            x = torch.randn(m, k, dtype=torch.bfloat16, device="cuda")
            tun.enable(False)
            base = cold_time(x, ws)
            tun.enable(True)
            tuned = cold_time(x, ws)
            gb = (n * k + m * k + m * n) * 2 / 1e9
            print(f"{name:8s} M={m:4d}: default {base:7.1f} us ({gb / base * 1e6:6.0f} GB/s)  tuned {tuned:7.1f} us "
                  f"({gb / tuned * 1e6:6.0f} GB/s)  x{base / tuned:4.2f}", flush=True)
        del ws


if __name__ == "__main__":
    main()
