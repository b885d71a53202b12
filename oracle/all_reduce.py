"""CPU restatement of the reference's custom all-reduce arithmetic.  TEST INFRASTRUCTURE ONLY.

Follows csrc/custom_all_reduce.cuh: ``packed_reduce`` (:204-212) adds the ranks' values in fp32,
one after the other, and rounds once; ``cross_device_reduce_1stage`` (:214-233) takes the ranks in
the order 0..n-1 everywhere, ``cross_device_reduce_2stage`` (:239-290) lets rank r reduce slice r
of ``size // n`` 16-byte vectors (the last rank also takes the remainder) starting from its own
input: r, r+1, ..., r+n-1 (mod n).  Pinned by the reference's own test
(tests/distributed/test_custom_all_reduce.py:24-57: the result equals the NCCL / torch sum within
rounding) -- the bit patterns themselves are pinned by nothing but this restatement of the source:
PARITY UNPINNED for the exact rounding.
"""
from typing import List

import torch


def _sum_in_order(inputs: List[torch.Tensor], order: List[int]) -> torch.Tensor:
    acc = inputs[order[0]].float().clone()
    for r in order[1:]:
        acc += inputs[r].float()
    return acc.to(inputs[0].dtype)


def one_stage(inputs: List[torch.Tensor]) -> torch.Tensor:
    return _sum_in_order(inputs, list(range(len(inputs))))


def two_stage(inputs: List[torch.Tensor]) -> torch.Tensor:
    n = len(inputs)
    flat = [t.reshape(-1) for t in inputs]
    per_vec = 16 // flat[0].element_size()
    n_vec = flat[0].numel() // per_vec
    part = n_vec // n
    out = torch.empty_like(flat[0])
    for r in range(n):
        lo = r * part * per_vec
        hi = flat[0].numel() if r == n - 1 else (r + 1) * part * per_vec
        order = [(r + i) % n for i in range(n)]
        out[lo:hi] = _sum_in_order([t[lo:hi] for t in flat], order)
    return out.view(inputs[0].shape)
