#!/usr/bin/env python3
"""Decode attention over RAGGED batches: the unpartitioned kernel (one workgroup walks a whole context) against
the partitioned one (512-token pieces, balanced by construction), for context lengths with a long tail.
Prints time, achieved bandwidth over the algorithmic bytes, and the ratio max / mean context."""
import sys

sys.path[:0] = ["/root/repo", "/root/repo/vllm-splitwise_amd", "/root/repo/tools"]
import torch  # noqa: E402
from splitwise_amd import _C  # noqa: E402
from bench_kernels import timeit  # noqa: E402


def run(lens, h=32, kvh=32, d=128, bs=16, dtype=torch.bfloat16, order="given"):
    dev = "cuda"
    if order == "longest last":
        lens = sorted(lens)
    elif order == "longest first":
        lens = sorted(lens, reverse=True)
    n, mx = len(lens), max(lens)
    per_seq = (mx + bs - 1) // bs
    blocks = sum((L + bs - 1) // bs for L in lens) + 1
    kc = torch.empty(blocks, kvh, d // 8, bs, 8, dtype=dtype, device=dev).uniform_(-0.1, 0.1)
    vc = torch.empty(blocks, kvh, d, bs, dtype=dtype, device=dev).uniform_(-0.1, 0.1)
    perm = torch.randperm(blocks - 1).tolist()
    bt = torch.zeros(n, per_seq, dtype=torch.int32)
    cur = 0
    for i, L in enumerate(lens):
        nb = (L + bs - 1) // bs
        bt[i, :nb] = torch.tensor(perm[cur:cur + nb], dtype=torch.int32)
        cur += nb
    bt = bt.to(dev)
    cl = torch.tensor(lens, dtype=torch.int32, device=dev)
    q = torch.empty(n, h, d, dtype=dtype, device=dev).uniform_(-0.1, 0.1)
    out = torch.empty_like(q)
    scale = d ** -0.5
    parts = (mx + 511) // 512
    tmp = torch.empty(n, h, parts, d, dtype=dtype, device=dev)
    esum = torch.empty(n, h, parts, dtype=torch.float32, device=dev)
    ml = torch.empty_like(esum)
    t1 = timeit(lambda: _C.ops.paged_attention_v1(out, q, kc, vc, kvh, scale, bt, cl, bs, mx, None, "auto"))
    t2 = timeit(lambda: _C.ops.paged_attention_v2(out, esum, ml, tmp, q, kc, vc, kvh, scale, bt, cl, bs, mx, None, "auto"))
    nbytes = sum(lens) * 2 * kvh * d * 2
    mean = sum(lens) / n
    print(f"S={n:4d} H={h}/{kvh} mean={mean:7.0f} max={mx:6d} (x{mx / mean:4.1f}) {order:13s}: "
          f"v1 {t1 * 1e6:8.1f} us {nbytes / t1 / 1e9:7.0f} GB/s | v2 {t2 * 1e6:8.1f} us {nbytes / t2 / 1e9:7.0f} GB/s", flush=True)


def main():
    g = torch.Generator().manual_seed(0)
    uniform = [int(x) for x in torch.randint(772, 943, (256,), generator=g)]
    # log-normal lengths clipped to 16..8192 (a long tail: a few sequences many times the mean)
    tail = [int(min(8192, max(16, x))) for x in torch.exp(torch.randn(256, generator=g) * 1.0 + 6.3)]
    few_long = [600] * 250 + [6000] * 6
    for lens in (uniform, tail, few_long):
        for order in ("given", "longest last", "longest first"):
            run(lens, order=order)
    run(tail, 16, 2)
    run(tail[:64], 64, 8)


if __name__ == "__main__":
    main()
