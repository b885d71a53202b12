#!/usr/bin/env python3
"""Durations of the row-wise kernels beside the attention path at the decode shape of the
benchmark (256 rows of Llama-2-7B), each measured as 64 back-to-back launches inside a replayed
HIP graph, the way they run in a decode step."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "vllm-splitwise_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import torch  # noqa: E402
from splitwise_amd import _C  # noqa: E402


def graph_time(fn, reps=64, replays=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(replays):
        g.replay()
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) * 1e3 / (reps * replays)


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    dt, dev = torch.bfloat16, "cuda"
    hidden, heads, d, ffn = 4096, 32, 128, 11008
    x = torch.randn(rows, hidden, dtype=dt, device=dev)
    res = torch.randn_like(x)
    w = torch.ones(hidden, dtype=dt, device=dev)
    out = torch.empty_like(x)
    print(f"rows={rows}")
    t = graph_time(lambda: _C.ops.rms_norm(out, x, w, 1e-5))
    print(f"rms_norm            {t:7.2f} us  ({2 * x.numel() * 2 / t / 1e3:7.1f} GB/s)")
    t = graph_time(lambda: _C.ops.fused_add_rms_norm(x, res, w, 1e-5))
    print(f"fused_add_rms_norm  {t:7.2f} us  ({4 * x.numel() * 2 / t / 1e3:7.1f} GB/s)")
    gu = torch.randn(rows, 2 * ffn, dtype=dt, device=dev)
    act = torch.empty(rows, ffn, dtype=dt, device=dev)
    t = graph_time(lambda: _C.ops.silu_and_mul(act, gu))
    print(f"silu_and_mul        {t:7.2f} us  ({3 * act.numel() * 2 / t / 1e3:7.1f} GB/s)")
    qkv = torch.randn(rows, 3 * hidden, dtype=dt, device=dev)
    q, k, v = qkv.split([hidden, hidden, hidden], dim=-1)
    pos = torch.randint(0, 1000, (rows,), dtype=torch.int64, device=dev)
    cache = torch.randn(2048, d, dtype=dt, device=dev)
    t = graph_time(lambda: _C.ops.rotary_embedding(pos, q, k, d, cache, True))
    print(f"rotary_embedding    {t:7.2f} us  ({4 * rows * hidden * 2 / t / 1e3:7.1f} GB/s)")
    nb = 4096
    kc = torch.empty(nb, heads, d // 8, 16, 8, dtype=dt, device=dev)
    vc = torch.empty(nb, heads, d, 16, dtype=dt, device=dev)
    slots = (torch.randperm(nb, device=dev)[:rows] * 16 + 5).to(torch.int64)
    kk, vv = k.view(rows, heads, d), v.view(rows, heads, d)
    t = graph_time(lambda: _C.cache_ops.reshape_and_cache(kk, vv, kc, vc, slots, "auto"))
    print(f"reshape_and_cache   {t:7.2f} us  ({4 * rows * hidden * 2 / t / 1e3:7.1f} GB/s)")
    t = graph_time(lambda: _C.ops.rotary_embedding_and_cache(pos, q, k, v, kc, vc, slots, d, cache, True, "auto"))
    print(f"rotary + cache, one launch {t:7.2f} us  ({7 * rows * hidden * 2 / t / 1e3:7.1f} GB/s)")


if __name__ == "__main__":
    main()
