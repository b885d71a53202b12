#!/usr/bin/env python3
"""Row-wise kernels on COLD data: every launch of a replayed graph works on its own buffers (more than the
256 MB memory-side cache between two uses of a buffer), as in a model step where the input was written by a GEMM
long ago.  Prints time and bandwidth at `rows` rows of Llama-2-7B."""
import sys

sys.path[:0] = ["/root/repo", "/root/repo/vllm-splitwise_amd", "/root/repo/tools"]
import torch  # noqa: E402
from splitwise_amd import _C  # noqa: E402


def graph_time(fns, replays=10):
    for f in fns:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for f in fns:
            f()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(replays):
        g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (replays * len(fns))


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2316
    sets = max(4, int(600e6 / (rows * 22016 * 2)) + 2)
    dt, dev = torch.bfloat16, "cuda"
    hidden, ffn = 4096, 11008
    w = torch.ones(hidden, dtype=dt, device=dev)
    gus = [torch.randn(rows, 2 * ffn, dtype=dt, device=dev) for _ in range(sets)]
    acts = [torch.empty(rows, ffn, dtype=dt, device=dev) for _ in range(sets)]
    t = graph_time([lambda i=i: _C.ops.silu_and_mul(acts[i], gus[i]) for i in range(sets)])
    print(f"rows={rows}  silu_and_mul        {t:7.2f} us  ({3 * rows * ffn * 2 / t / 1e3:7.1f} GB/s)  {sets} buffer sets")
    sets_n = max(8, int(600e6 / (rows * hidden * 2 * 2)) + 2)
    xs = [torch.randn(rows, hidden, dtype=dt, device=dev) for _ in range(sets_n)]
    rs = [torch.randn(rows, hidden, dtype=dt, device=dev) for _ in range(sets_n)]
    t = graph_time([lambda i=i: _C.ops.fused_add_rms_norm(xs[i], rs[i], w, 1e-5) for i in range(sets_n)])
    print(f"rows={rows}  fused_add_rms_norm  {t:7.2f} us  ({4 * rows * hidden * 2 / t / 1e3:7.1f} GB/s)  {sets_n} buffer sets")


if __name__ == "__main__":
    main()
