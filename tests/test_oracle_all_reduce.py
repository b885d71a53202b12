"""The CPU restatement of the reference's all-reduce arithmetic (csrc/custom_all_reduce.cuh:204-290)
on known answers: fp32 accumulation with one rounding, rank order 0..n-1 in one stage, owner-first
order per slice in two stages."""
import torch

import conftest  # noqa: F401
from oracle import all_reduce as oracle


def test_accumulates_in_fp32_and_rounds_once():
    # 2048 + 1 + 1 is 2050 in fp32 and stays 2050 in fp16; adding in fp16 would stick at 2048
    ins = [torch.full((8,), v, dtype=torch.float16) for v in (2048.0, 1.0, 1.0, 0.0)]
    assert oracle.one_stage(ins).tolist() == [2050.0] * 8
    # bf16: 256 + 1 + 1 + 1 = 259 -> rounds to 260 once; step by step in bf16 it would stay 256
    ins = [torch.full((8,), v, dtype=torch.bfloat16) for v in (256.0, 1.0, 1.0, 1.0)]
    assert oracle.one_stage(ins).tolist() == [260.0] * 8


def test_two_stage_order_starts_at_the_slice_owner():
    # fp32: (1e8 + 1) - 1e8 = 0 in rank order 0,1,2,3 but slices owned by other ranks start elsewhere
    vals = (1e8, 1.0, -1e8, 0.0)
    ins = [torch.full((16,), v, dtype=torch.float32) for v in vals]          # 4 vectors of 4 floats: one per rank
    one = oracle.one_stage(ins)
    two = oracle.two_stage(ins)
    assert one.tolist() == [0.0] * 16                       # ((1e8 + 1) - 1e8) + 0: the 1 is absorbed
    # owner 1: 1 - 1e8 + 0 + 1e8 = 0 ; owner 2: -1e8 + 0 + 1e8 + 1 = 1 ; owner 3: 0 + 1e8 + 1 - 1e8 = 0
    assert two.tolist() == [0.0] * 4 + [0.0] * 4 + [1.0] * 4 + [0.0] * 4


def test_two_ranks_agree_in_both_forms_and_last_rank_takes_the_remainder():
    g = torch.Generator().manual_seed(0)
    ins = [torch.randn(8 * 7, generator=g).to(torch.float16) for _ in range(2)]
    assert torch.equal(oracle.one_stage(ins), oracle.two_stage(ins))
    ins = [torch.randn(8 * 11, generator=g).to(torch.bfloat16) for _ in range(4)]      # 11 vectors: slices 2,2,2,5
    two = oracle.two_stage(ins)
    want_tail = oracle._sum_in_order([t[48:] for t in ins], [3, 0, 1, 2])
    assert torch.equal(two[48:], want_tail)
