#!/usr/bin/env python3
"""Prompt attention over a cached prefix: the unsplit form against the split form (key stages of a query tile
dealt out to several workgroups + a reduce kernel) over the number of parts, and the library's own choice."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vllm-splitwise_amd"), os.path.join(ROOT, "tools")]
import torch  # noqa: E402
from splitwise_amd import _C  # noqa: E402
from bench_kernels import timeit  # noqa: E402


def case(new, ctx, seqs=1, h=32, kvh=32, d=128, bs=16, dtype=torch.bfloat16):
    s = d ** -0.5
    q = torch.empty(new * seqs, h, d, dtype=dtype, device="cuda").uniform_(-s, s)
    k = torch.empty(new * seqs, kvh, d, dtype=dtype, device="cuda").uniform_(-s, s)
    v = torch.empty(new * seqs, kvh, d, dtype=dtype, device="cuda").uniform_(-s, s)
    out = torch.empty_like(q)
    blocks = -(-ctx // bs)
    k_cache = torch.empty(blocks * seqs, kvh, d // 8, bs, 8, dtype=dtype, device="cuda").uniform_(-s, s)
    v_cache = torch.empty(blocks * seqs, kvh, d, bs, dtype=dtype, device="cuda").uniform_(-s, s)
    table = torch.randperm(blocks * seqs, device="cuda").to(torch.int32).view(seqs, -1)
    start = torch.arange(seqs, device="cuda") * new
    total = torch.full((seqs,), new + ctx, device="cuda")
    cl = torch.full((seqs,), ctx, device="cuda")
    flops = 4.0 * h * d * (new * ctx + new * new / 2) * seqs
    auto = _C.load_library().sw_context_attention_split_parts(seqs, new, h, d, bs, table.shape[1] * bs)
    line = f"new={new:5d} ctx={ctx:5d} seqs={seqs} H={h}/{kvh}:"
    for parts in (1, 2, 3, 4, 6, 8):
        t = timeit(lambda: _C.ops.context_attention_fwd(q, k, v, out, k_cache, v_cache, table, start, total, cl, new,
                                                        None, None, parts), iters=30)
        line += f"  p{parts}{'*' if parts == auto else ''} {t * 1e6:6.1f} us ({flops / t / 1e15:4.2f} PF)"
    print(line, flush=True)


if __name__ == "__main__":
    case(512, 3584)
    case(772, 772)
    case(2048, 2048)
    case(256, 4096)
    case(128, 8192)
    case(64, 2048)
    case(512, 3584, seqs=2)
    case(1024, 1024, h=64, kvh=8)
    case(512, 3584, h=16, kvh=2)
