"""GPU parity: paged_attention_v1 / v2 through the C ABI vs the CPU oracle and vs the
fixtures produced by the reference's own test reference.

Mirrors tests/kernels/test_attention.py:114-269 of the reference: 7 sequences, heads
(40, 40) and (64, 8), every supported head size, block sizes 16 / 32 (and 8), ALiBi on
and off, kv cache "auto" and "fp8_e5m2".  Tolerance is the reference's: atol 1e-3,
rtol 1e-5 (fp8 cache: atol 1e-2), tests/kernels/test_attention.py:262-269.
"""
import glob
import os
import random

import pytest
import torch

from helpers import DTYPES, GOLDEN, from_np, load_golden, make_kv_caches
from oracle import attention as oattn

pytestmark = pytest.mark.gpu

PARTITION_SIZE = 512
NUM_BLOCKS = 1024
PAGED = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "paged_attention_*.npz")))


def _C():
    from splitwise_amd import _C
    return _C


def run_kernel(version, query, kc, vc, num_kv_heads, scale, block_tables, context_lens, block_size,
               alibi, kv_cache_dtype):
    C = _C()
    out = torch.empty_like(query)
    max_ctx = int(context_lens.max().item())
    if version == "v1":
        C.ops.paged_attention_v1(out, query, kc, vc, num_kv_heads, scale, block_tables, context_lens,
                                 block_size, max_ctx, alibi, kv_cache_dtype)
    else:
        num_seqs, num_heads, head_size = query.shape
        parts = (max_ctx + PARTITION_SIZE - 1) // PARTITION_SIZE
        tmp = torch.empty(num_seqs, num_heads, parts, head_size, dtype=query.dtype, device=query.device)
        exp_sums = torch.empty(num_seqs, num_heads, parts, dtype=torch.float32, device=query.device)
        max_logits = torch.empty_like(exp_sums)
        C.ops.paged_attention_v2(out, exp_sums, max_logits, tmp, query, kc, vc, num_kv_heads, scale,
                                 block_tables, context_lens, block_size, max_ctx, alibi, kv_cache_dtype)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("name", PAGED)
def test_against_reference_fixture(version, name):
    z = load_golden(name)
    dtype = DTYPES[str(z["dtype"])]
    q = from_np(z["query"], dtype).cuda()
    kc, vc = from_np(z["key_cache"], dtype).cuda(), from_np(z["value_cache"], dtype).cuda()
    slopes = torch.from_numpy(z["alibi_slopes"]).cuda() if z["alibi_slopes"].size else None
    out = run_kernel(version, q, kc, vc, int(z["num_kv_heads"]), float(z["scale"]),
                     torch.from_numpy(z["block_tables"]).cuda(), torch.from_numpy(z["context_lens"]).cuda(),
                     int(z["block_size"]), slopes, "auto")
    torch.testing.assert_close(out.cpu().float(), from_np(z["out"], dtype).float(), atol=1e-3, rtol=1e-5)


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("num_heads", [(40, 40), (64, 8)])
@pytest.mark.parametrize("head_size", [64, 80, 96, 112, 128, 256])
@pytest.mark.parametrize("use_alibi", [False, True])
@pytest.mark.parametrize("block_size", [16, 32])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("kv_cache_dtype", ["auto", "fp8_e5m2"])
def test_paged_attention(version, num_heads, head_size, use_alibi, block_size, dtype, kv_cache_dtype):
    random.seed(0)
    torch.manual_seed(0)
    num_seqs, max_seq_len = 7, 1500
    num_query_heads, num_kv_heads = num_heads
    scale = float(1.0 / head_size ** 0.5)
    query = torch.empty(num_seqs, num_query_heads, head_size, dtype=dtype).uniform_(-scale, scale)
    alibi = torch.randn(num_query_heads, dtype=torch.float) if use_alibi else None
    context_lens = [random.randint(1, max_seq_len) for _ in range(num_seqs)]
    context_lens[-1] = max_seq_len
    context_lens[0] = 1
    max_blocks = (max_seq_len + block_size - 1) // block_size
    tables = [[random.randint(0, NUM_BLOCKS - 1) for _ in range(max_blocks)] for _ in range(num_seqs)]
    block_tables = torch.tensor(tables, dtype=torch.int)
    context_lens = torch.tensor(context_lens, dtype=torch.int)
    kc, vc = make_kv_caches(NUM_BLOCKS, block_size, num_kv_heads, head_size, dtype, cache_dtype=kv_cache_dtype)

    ref = oattn.paged_attention(query, kc, vc, block_tables, context_lens, scale, alibi)
    out = run_kernel(version, query.cuda(), kc.cuda(), vc.cuda(), num_kv_heads, scale, block_tables.cuda(),
                     context_lens.cuda(), block_size, alibi.cuda() if use_alibi else None, kv_cache_dtype)
    atol = 1e-2 if kv_cache_dtype == "fp8_e5m2" else 1e-3
    torch.testing.assert_close(out.cpu().float(), ref.float(), atol=atol, rtol=1e-5)


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("dtype,block_size,head_size,heads", [
    (torch.float32, 16, 128, (8, 8)), (torch.float32, 32, 80, (8, 2)), (torch.float32, 8, 64, (4, 4)),
    (torch.float16, 8, 128, (8, 4)), (torch.bfloat16, 8, 96, (4, 4))])
@pytest.mark.parametrize("kv_cache_dtype", ["auto", "fp8_e5m2"])
def test_paged_attention_other_layouts(version, dtype, block_size, head_size, heads, kv_cache_dtype):
    random.seed(1)
    torch.manual_seed(1)
    h, kvh = heads
    num_seqs, max_seq_len, nb = 5, 700, 256
    scale = float(1.0 / head_size ** 0.5)
    query = torch.empty(num_seqs, h, head_size, dtype=dtype).uniform_(-scale, scale)
    context_lens = torch.tensor([random.randint(1, max_seq_len) for _ in range(num_seqs - 1)] + [max_seq_len],
                                dtype=torch.int)
    max_blocks = (max_seq_len + block_size - 1) // block_size
    block_tables = torch.tensor([[random.randint(0, nb - 1) for _ in range(max_blocks)] for _ in range(num_seqs)],
                                dtype=torch.int)
    kc, vc = make_kv_caches(nb, block_size, kvh, head_size, dtype, cache_dtype=kv_cache_dtype)
    ref = oattn.paged_attention(query, kc, vc, block_tables, context_lens, scale, None)
    out = run_kernel(version, query.cuda(), kc.cuda(), vc.cuda(), kvh, scale, block_tables.cuda(),
                     context_lens.cuda(), block_size, None, kv_cache_dtype)
    atol = 1e-2 if kv_cache_dtype == "fp8_e5m2" else 1e-3
    torch.testing.assert_close(out.cpu().float(), ref.float(), atol=atol, rtol=1e-5)


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("heads,num_seqs,dtype,kv_cache_dtype", [
    ((16, 16), 64, torch.bfloat16, "auto"), ((16, 16), 72, torch.float16, "fp8_e5m2"),
    ((64, 8), 128, torch.float16, "auto"), ((64, 8), 128, torch.bfloat16, "fp8_e5m2"),
    ((16, 16), 65, torch.bfloat16, "auto")])
def test_large_launches_with_length_sorted_ragged_contexts(version, heads, num_seqs, dtype, kv_cache_dtype):
    """Launches of a thousand workgroups and more renumber their workgroups so that a sequence's heads share an
    XCD (XCD x: sequences x, x + 8, ... when the batch is a multiple of 8; a contiguous range otherwise -- the
    65-sequence case).  Contexts from 1 token to 1200, sorted by length as an age-ordered batch would be, both
    kernels, vector and matrix-core paths, against the oracle."""
    random.seed(2)
    torch.manual_seed(2)
    h, kvh = heads
    head_size, block_size, nb = 128, 16, 2048
    scale = float(head_size ** -0.5)
    query = torch.empty(num_seqs, h, head_size, dtype=dtype).uniform_(-scale, scale)
    lens = sorted([1, 2, 15, 16, 17, 511, 512, 513, 1200] + [random.randint(1, 1200) for _ in range(num_seqs - 9)],
                  reverse=True)
    max_blocks = (max(lens) + block_size - 1) // block_size
    block_tables = torch.tensor([[random.randint(0, nb - 1) for _ in range(max_blocks)] for _ in range(num_seqs)],
                                dtype=torch.int)
    context_lens = torch.tensor(lens, dtype=torch.int)
    kc, vc = make_kv_caches(nb, block_size, kvh, head_size, dtype, cache_dtype=kv_cache_dtype)
    ref = oattn.paged_attention(query, kc, vc, block_tables, context_lens, scale, None)
    out = run_kernel(version, query.cuda(), kc.cuda(), vc.cuda(), kvh, scale, block_tables.cuda(),
                     context_lens.cuda(), block_size, None, kv_cache_dtype)
    atol = 1e-2 if kv_cache_dtype == "fp8_e5m2" else 1e-3
    torch.testing.assert_close(out.cpu().float(), ref.float(), atol=atol, rtol=1e-5)


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("heads", [(16, 16), (64, 8)])
def test_bf16_queries_beyond_the_f16_range_over_an_fp8_cache(version, heads):
    """bf16 activations over an e5m2 cache run their products in f16 on the matrix cores.  bf16 reaches 3e38,
    f16 ends at 65504: queries of magnitude 1e5 (and 1e30) must give the scores the reference's fp32 products
    give (fp8 widened to bf16, quant_utils.cuh:73-81; attention_utils.cuh:29-46), not infinities.  Only some
    heads are large, so a per-head rescaling is exercised next to untouched heads."""
    random.seed(5)
    torch.manual_seed(5)
    h, kvh = heads
    dtype, head_size, block_size, nb, num_seqs = torch.bfloat16, 128, 16, 512, 6
    scale = float(head_size ** -0.5)
    query = torch.empty(num_seqs, h, head_size, dtype=dtype).uniform_(-scale, scale)
    query[:, 1::3] *= 1.2e6                    # |q| up to 1e5: beyond f16
    query[:, 2::5] *= 1e31                     # far beyond
    query[0, 0, :] = 0.0                       # and an all-zero head
    lens = [1, 17, 300, 700, 64, 513]
    max_blocks = (max(lens) + block_size - 1) // block_size
    block_tables = torch.tensor([[random.randint(0, nb - 1) for _ in range(max_blocks)] for _ in range(num_seqs)],
                                dtype=torch.int)
    context_lens = torch.tensor(lens, dtype=torch.int)
    kc, vc = make_kv_caches(nb, block_size, kvh, head_size, dtype, cache_dtype="fp8_e5m2")
    ref = oattn.paged_attention(query, kc, vc, block_tables, context_lens, scale, None)
    assert torch.isfinite(ref.float()).all()
    out = run_kernel(version, query.cuda(), kc.cuda(), vc.cuda(), kvh, scale, block_tables.cuda(),
                     context_lens.cuda(), block_size, None, "fp8_e5m2")
    assert torch.isfinite(out.float()).all()
    torch.testing.assert_close(out.cpu().float(), ref.float(), atol=1e-2, rtol=1e-5)


@pytest.mark.parametrize("version", ["v1", "v2"])
@pytest.mark.parametrize("dtype,heads,block_size", [(torch.bfloat16, (32, 32), 16), (torch.float16, (32, 32), 32),
                                                    (torch.bfloat16, (32, 8), 16)])
def test_both_forms_of_the_vector_kernel_give_the_same_bits(version, dtype, heads, block_size):
    """The vector kernel has two forms, picked by the number of workgroups of the launch (up to 4096: the group's
    table entries are read once and handed from lane to lane; above: read in front of every block's loads).  The
    arithmetic is the same, so a sequence must come out bit-identical whichever launch it is part of: 300 sequences
    in one launch (9600 workgroups per partition; 2400 of the matrix-core kernel in the grouped-query case)
    against the same sequences 50 at a time (1600 per partition), and against the oracle."""
    random.seed(5)
    torch.manual_seed(5)
    h, kvh = heads
    num_seqs, head_size, nb = 300, 128, 1024
    scale = float(head_size ** -0.5)
    query = torch.empty(num_seqs, h, head_size, dtype=dtype).uniform_(-scale, scale)
    lens = [1, 15, 16, 17, 63, 64, 65, 512, 513, 700] + [random.randint(1, 700) for _ in range(num_seqs - 10)]
    max_blocks = (max(lens) + block_size - 1) // block_size
    block_tables = torch.tensor([[random.randint(0, nb - 1) for _ in range(max_blocks)] for _ in range(num_seqs)],
                                dtype=torch.int)
    context_lens = torch.tensor(lens, dtype=torch.int)
    kc, vc = make_kv_caches(nb, block_size, kvh, head_size, dtype)
    q, kcd, vcd, bt, cl = query.cuda(), kc.cuda(), vc.cuda(), block_tables.cuda(), context_lens.cuda()
    whole = run_kernel(version, q, kcd, vcd, kvh, scale, bt, cl, block_size, None, "auto")
    for lo in range(0, num_seqs, 50):
        part = run_kernel(version, q[lo:lo + 50].contiguous(), kcd, vcd, kvh, scale, bt[lo:lo + 50].contiguous(),
                          cl[lo:lo + 50].contiguous(), block_size, None, "auto")
        assert torch.equal(part, whole[lo:lo + 50]), lo
    ref = oattn.paged_attention(query[:40], kc, vc, block_tables[:40], context_lens[:40], scale, None)
    torch.testing.assert_close(whole[:40].cpu().float(), ref.float(), atol=1e-3, rtol=1e-5)


def test_strided_query_and_nan_tail():
    """The query is a slice of a fused qkv tensor (q_stride != H*D), and the unused tail of
    the last block holds NaNs that must not leak (attention_kernels.cu:342-351)."""
    torch.manual_seed(0)
    dtype, h, kvh, d, bs, nb = torch.bfloat16, 8, 8, 128, 16, 64
    scale = d ** -0.5
    qkv = torch.empty(3, 3 * h * d, dtype=dtype).uniform_(-scale, scale)
    query = qkv[:, :h * d].view(3, h, d)
    kc, vc = make_kv_caches(nb, bs, kvh, d, dtype)
    context_lens = torch.tensor([5, 37, 130], dtype=torch.int)
    block_tables = torch.tensor([[1, 0, 0, 0, 0, 0, 0, 0, 0], [2, 3, 4, 0, 0, 0, 0, 0, 0],
                                 [5, 6, 7, 8, 9, 10, 11, 12, 13]], dtype=torch.int)
    for seq, ctx in enumerate(context_lens.tolist()):          # poison what lies past the context
        last = block_tables[seq, (ctx - 1) // bs]
        kc[last, :, :, ctx % bs:, :] = float("nan")
        vc[last, :, :, ctx % bs:] = float("nan")
    ref = oattn.paged_attention(query, kc, vc, block_tables, context_lens, scale)
    assert not torch.isnan(ref).any()
    qd = qkv.cuda()[:, :h * d].view(3, h, d)
    for version in ("v1", "v2"):
        out = run_kernel(version, qd, kc.cuda(), vc.cuda(), kvh, scale, block_tables.cuda(),
                         context_lens.cuda(), bs, None, "auto")
        torch.testing.assert_close(out.cpu().float(), ref.float(), atol=1e-3, rtol=1e-5)


def test_unsupported_shapes_raise():
    C = _C()
    q = torch.zeros(1, 4, 72, dtype=torch.float16, device="cuda")
    kc = torch.zeros(2, 4, 9, 16, 8, dtype=torch.float16, device="cuda")
    vc = torch.zeros(2, 4, 72, 16, dtype=torch.float16, device="cuda")
    bt = torch.zeros(1, 1, dtype=torch.int, device="cuda")
    cl = torch.ones(1, dtype=torch.int, device="cuda")
    with pytest.raises(RuntimeError, match="Unsupported head size"):
        C.ops.paged_attention_v1(torch.empty_like(q), q, kc, vc, 4, 1.0, bt, cl, 16, 1, None, "auto")
    q = torch.zeros(1, 4, 64, dtype=torch.float16, device="cuda")
    kc = torch.zeros(2, 4, 8, 4, 8, dtype=torch.float16, device="cuda")
    vc = torch.zeros(2, 4, 64, 4, dtype=torch.float16, device="cuda")
    with pytest.raises(RuntimeError, match="Unsupported block size"):
        C.ops.paged_attention_v1(torch.empty_like(q), q, kc, vc, 4, 1.0, bt, cl, 4, 1, None, "auto")
