#!/usr/bin/env python3
"""Tunes the ROCm BLAS solution of the decode-step GEMMs (torch TunableOp) and reports the gain.

  python tools/tune_gemm.py <model preset> <out.csv> [M ...]

Measured on the bench's shapes (profiles/r1/gemm_tunableop_v1.log): 1-2 % at 200-256 rows, 3-8 % on the
prompt steps' shapes -- too little to carry a tuned-solution file around, so bench.py does not load one.
"""
import os
import sys

out = sys.argv[2]
os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
os.environ["PYTORCH_TUNABLEOP_TUNING"] = "1"
os.environ["PYTORCH_TUNABLEOP_FILENAME"] = out
os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "60")
os.environ.setdefault("PYTORCH_TUNABLEOP_VERBOSE", "0")

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vllm-splitwise_amd"))
from splitwise_amd.config import MODEL_PRESETS  # noqa: E402


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    cfg = MODEL_PRESETS[sys.argv[1]]
    ms = [int(x) for x in sys.argv[3:]] or [1, 2, 4, 8, 16, 32, 64, 128, 256]
    h, ffn = cfg["hidden_size"], cfg["intermediate_size"]
    heads, kvh = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    d = h // heads
    shapes = {"qkv": ((heads + 2 * kvh) * d, h), "o": (h, h), "gate_up": (2 * ffn, h), "down": (h, ffn),
              "lm_head": (cfg["vocab_size"], h)}
    dtype = torch.bfloat16 if cfg["torch_dtype"] == "bfloat16" else torch.float16
    tun = torch.cuda.tunable
    for name, (n, k) in shapes.items():
        w = torch.randn(n, k, dtype=dtype, device="cuda") * 0.02
        for m in ms:
            x = torch.randn(m, k, dtype=dtype, device="cuda")
            tun.enable(False)
            base = timeit(lambda: F.linear(x, w))
            tun.enable(True)
            tuned = timeit(lambda: F.linear(x, w))
            gb = (n * k + m * k + m * n) * 2 / 1e9
            print(f"{name:8s} M={m:4d} N={n:6d} K={k:6d}: default {base:7.1f} us ({gb / base * 1e6:6.0f} GB/s)  "
                  f"tuned {tuned:7.1f} us ({gb / tuned * 1e6:6.0f} GB/s)  x{base / tuned:4.2f}", flush=True)
    # the results file (<out> with the device ordinal before the extension) is written when the process exits


if __name__ == "__main__":
    main()
