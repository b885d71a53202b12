"""GPU checks at the benchmark's full sizes, where the CPU oracle would take minutes: properties
that hold whatever the size -- exact scaling, invariance under a permutation of the physical
blocks, agreement of the two decode kernels, write/read-back round trips, idempotence of the fp8
codes, consistency of the fused int4 GEMM with the (bit-exactly tested) reconstruction kernel.
Shapes: Llama-2-7B decode at 256 sequences of ~940 tokens (the bench), a 3 x 772-token prompt
step, the Llama-2-13B / TP2 int4 matrices."""
import random

import pytest
import torch

from test_paged_attention_gpu import run_kernel

pytestmark = pytest.mark.gpu

H, D, BS = 32, 128, 16


def _decode_case(num_seqs=256, lo=800, hi=943, dtype=torch.bfloat16, seed=0, heads=H, kv_heads=H):
    g = torch.Generator(device="cuda").manual_seed(seed)
    random.seed(seed)
    ctx = [random.randint(lo, hi) for _ in range(num_seqs)]
    per_seq = -(-hi // BS)
    num_blocks = num_seqs * per_seq
    scale = D ** -0.5
    q = (torch.rand(num_seqs, heads, D, device="cuda", generator=g) * 2 - 1).mul_(scale).to(dtype)
    kc = (torch.rand(num_blocks, kv_heads, D // 8, BS, 8, device="cuda", generator=g) * 2 - 1).mul_(scale).to(dtype)
    vc = (torch.rand(num_blocks, kv_heads, D, BS, device="cuda", generator=g) * 2 - 1).mul_(scale).to(dtype)
    tables = torch.randperm(num_blocks, device="cuda", generator=g).to(torch.int32).view(num_seqs, per_seq)
    lens = torch.tensor(ctx, dtype=torch.int32, device="cuda")
    return q, kc, vc, tables, lens, scale


def test_decode_attention_properties_at_bench_size():
    q, kc, vc, tables, lens, scale = _decode_case()
    args = (H, scale, tables, lens, BS, None, "auto")
    v1 = run_kernel("v1", q, kc, vc, *args)
    v2 = run_kernel("v2", q, kc, vc, *args)
    assert torch.isfinite(v1.float()).all()
    torch.testing.assert_close(v1.float(), v2.float(), atol=1e-3, rtol=1e-2)       # same math, partitioned
    # doubling V doubles the output exactly (powers of two commute with every rounding here)
    assert torch.equal(run_kernel("v1", q, kc, vc * 2, *args), v1 * 2)
    # the physical placement of the blocks is invisible: permute blocks and tables together
    perm = torch.randperm(kc.shape[0], device="cuda")
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(perm.numel(), device="cuda")
    moved = run_kernel("v1", q, kc[perm], vc[perm], H, scale, inv[tables.long()].to(torch.int32), lens, BS, None, "auto")
    assert torch.equal(moved, v1)
    # constant values: the output is that constant times sum(p) / (sum(p) + 1e-6)
    const = torch.full_like(vc, 0.5)
    out = run_kernel("v1", q, kc, const, *args)
    torch.testing.assert_close(out.float(), torch.full_like(out, 0.5).float(), atol=4e-3, rtol=0)


@pytest.mark.parametrize("num_seqs,hi", [(64, 4096), (256, 2048)])
@pytest.mark.parametrize("kv_cache", ["auto", "fp8_e5m2"])
def test_decode_attention_properties_at_the_70b_tp4_shape(num_seqs, hi, kv_cache):
    """One rank of Llama-2-70B under tensor parallelism 4: 16 query heads over 2 KV heads (8 query heads
    share every K / V byte), bf16, 64-256 sequences of 2-4 K tokens.  Size-independent properties, and
    one sequence of the batch against the CPU oracle at the reference's tolerance."""
    from helpers import make_kv_caches  # noqa: F401  (same distribution as the small fixtures)
    from oracle import attention as oattn
    heads, kv_heads = 16, 2
    q, kc, vc, tables, lens, scale = _decode_case(num_seqs, hi // 2, hi, heads=heads, kv_heads=kv_heads, seed=5)
    if kv_cache == "fp8_e5m2":
        from splitwise_amd import _C
        kc8, vc8 = torch.empty_like(kc, dtype=torch.uint8), torch.empty_like(vc, dtype=torch.uint8)
        _C.cache_ops.convert_fp8_e5m2(kc.contiguous(), kc8)
        _C.cache_ops.convert_fp8_e5m2(vc.contiguous(), vc8)
        kc, vc = kc8, vc8
    args = (kv_heads, scale, tables, lens, BS, None, kv_cache)
    v1 = run_kernel("v1", q, kc, vc, *args)
    v2 = run_kernel("v2", q, kc, vc, *args)
    assert torch.isfinite(v1.float()).all()
    torch.testing.assert_close(v1.float(), v2.float(), atol=1e-3, rtol=1e-2)
    # each query head alone (as its own one-head launch over its KV head) gives the grouped launch's row
    for head in (0, 7, 8, 15):
        kv = head // (heads // kv_heads)
        alone = run_kernel("v1", q[:, head:head + 1].contiguous(), kc[:, kv:kv + 1].contiguous(),
                           vc[:, kv:kv + 1].contiguous(), 1, scale, tables, lens, BS, None, kv_cache)
        torch.testing.assert_close(alone[:, 0].float(), v1[:, head].float(), atol=2e-3 if kv_cache == "auto" else 1e-2,
                                   rtol=1e-2)
    # the physical placement of the blocks is invisible
    perm = torch.randperm(kc.shape[0], device="cuda")
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(perm.numel(), device="cuda")
    moved = run_kernel("v1", q, kc[perm], vc[perm], kv_heads, scale, inv[tables.long()].to(torch.int32), lens, BS, None,
                       kv_cache)
    assert torch.equal(moved, v1)
    # a sequence on its own equals the same sequence inside the batch (a launch of few workgroups spreads a
    # sequence over more waves, so the fp32 partial sums meet in another order: equal to the last bit or two)
    pick = num_seqs // 3
    one = run_kernel("v1", q[pick:pick + 1], kc, vc, kv_heads, scale, tables[pick:pick + 1], lens[pick:pick + 1], BS,
                     None, kv_cache)
    torch.testing.assert_close(one[0].float(), v1[pick].float(), atol=1e-4, rtol=1.6e-2)
    # and that sequence against the CPU oracle (gathering its blocks only)
    nblk = -(-int(lens[pick]) // BS)
    blocks = tables[pick, :nblk].long()
    ref = oattn.paged_attention(q[pick:pick + 1].cpu(), kc[blocks].cpu(), vc[blocks].cpu(),
                                torch.arange(nblk, dtype=torch.int32).view(1, -1), lens[pick:pick + 1].cpu(), scale, None)
    torch.testing.assert_close(v1[pick:pick + 1].cpu().float(), ref.float(), atol=1e-3 if kv_cache == "auto" else 1e-2,
                               rtol=1e-5)


def test_reshape_and_cache_round_trip_and_copy_swap_at_prompt_size():
    from splitwise_amd import _C
    torch.manual_seed(1)
    tokens, kvh, layers, blocks = 3 * 784, 32, 4, 400
    for kv_dtype, cache_dtype in (("auto", torch.bfloat16), ("fp8_e5m2", torch.uint8)):
        k = torch.randn(tokens, kvh, D, device="cuda").to(torch.bfloat16)
        v = torch.randn(tokens, kvh, D, device="cuda").to(torch.bfloat16)
        x = 16 // torch.tensor([], dtype=cache_dtype).element_size()
        kc = torch.zeros(blocks, kvh, D // x, BS, x, dtype=cache_dtype, device="cuda")
        vc = torch.zeros(blocks, kvh, D, BS, dtype=cache_dtype, device="cuda")
        table = torch.randperm(blocks, device="cuda")[:tokens // BS]
        slots = (table[:, None] * BS + torch.arange(BS, device="cuda")[None, :]).reshape(-1)
        _C.cache_ops.reshape_and_cache(k, v, kc, vc, slots, kv_dtype)
        k_back = kc[table].permute(0, 3, 1, 2, 4).reshape(tokens, kvh, D)
        v_back = vc[table].permute(0, 3, 1, 2).reshape(tokens, kvh, D)
        if kv_dtype == "auto":
            assert torch.equal(k_back, k) and torch.equal(v_back, v)
        else:
            # the codes decode to values that encode to the same codes (idempotent), within e5m2's step of the input
            dec = torch.empty(tokens, kvh, D, dtype=torch.bfloat16, device="cuda")
            _C.cache_ops.convert_fp8_e5m2(k_back.contiguous(), dec)
            again = torch.empty_like(k_back.contiguous())
            _C.cache_ops.convert_fp8_e5m2(dec, again)
            assert torch.equal(again, k_back.contiguous())
            torch.testing.assert_close(dec.float(), k.float(), atol=2.0 ** -16, rtol=0.125)   # e5m2: 2 mantissa bits, smallest code 2^-16
        untouched = torch.ones(blocks, dtype=torch.bool, device="cuda")
        untouched[table] = False
        assert not kc[untouched].any() and not vc[untouched].any()
    # copy_blocks + swap out / swap in over several layers: bit-exact
    caches = [(torch.randn(blocks, kvh, D // 8, BS, 8, device="cuda").to(torch.bfloat16),
               torch.randn(blocks, kvh, D, BS, device="cuda").to(torch.bfloat16)) for _ in range(layers)]
    before = [(kk.clone(), vv.clone()) for kk, vv in caches]
    mapping = {i: [200 + 2 * i, 201 + 2 * i] for i in range(64)}
    _C.cache_ops.copy_blocks([c[0] for c in caches], [c[1] for c in caches], mapping)
    torch.cuda.synchronize()
    for (kk, vv), (k0, v0) in zip(caches, before):
        for src, dsts in mapping.items():
            for d in dsts:
                assert torch.equal(kk[d], k0[src]) and torch.equal(vv[d], v0[src])
        assert torch.equal(kk[:200], k0[:200]) and torch.equal(kk[328:], k0[328:])
    host = torch.zeros(128, kvh, D, BS, dtype=torch.bfloat16).pin_memory()
    out_map = {3 * i + 1: i for i in range(128)}
    _C.cache_ops.swap_blocks(caches[0][1], host, out_map)
    back = torch.zeros_like(caches[0][1])
    _C.cache_ops.swap_blocks(host, back, {v: k for k, v in out_map.items()})
    torch.cuda.synchronize()
    for src in out_map:
        assert torch.equal(back[src], caches[0][1][src])


def test_prefill_attention_properties_at_prompt_size():
    from splitwise_amd import _C
    torch.manual_seed(2)
    lens = [772, 772, 772]
    n, s = sum(lens), D ** -0.5
    qkv = torch.empty(n, 3 * H, D, dtype=torch.bfloat16, device="cuda").uniform_(-s, s)
    q, k, v = qkv.split([H, H, H], dim=1)
    starts = torch.tensor([0, 772, 1544], dtype=torch.int32, device="cuda")
    ln = torch.tensor(lens, dtype=torch.int32, device="cuda")

    def run(vv):
        out = torch.empty(n, H, D, dtype=torch.bfloat16, device="cuda")
        _C.ops.prefill_attention(out, q, k, vv, starts, ln, 772, s)
        torch.cuda.synchronize()
        return out

    base = run(v)
    assert torch.isfinite(base.float()).all()
    assert torch.equal(run((v * 2).contiguous()), base * 2)                       # exact scaling in V
    # the first token of every sequence attends to itself only: its output is its own value row
    for st in (0, 772, 1544):
        assert torch.equal(base[st], v[st])
    # sequences are independent: running two of them without the first gives the same rows bit for bit ...
    pair = torch.empty(1544, H, D, dtype=torch.bfloat16, device="cuda")
    _C.ops.prefill_attention(pair, q[772:], k[772:], v[772:], starts[:2], ln[:2], 772, s)
    torch.cuda.synchronize()
    assert torch.equal(pair, base[772:])
    # ... and one alone -- a launch of less than one workgroup per CU, whose waves split the keys in pairs and sum in
    # another order -- the same rows to the last bit or two of the bf16 output
    alone = torch.empty(772, H, D, dtype=torch.bfloat16, device="cuda")
    _C.ops.prefill_attention(alone, q[772:1544], k[772:1544], v[772:1544], starts[:1], ln[:1], 772, s)
    torch.cuda.synchronize()
    torch.testing.assert_close(alone.float(), base[772:1544].float(), atol=1e-3, rtol=1e-2)


@pytest.mark.parametrize("fmt", ["gptq", "awq"])
@pytest.mark.parametrize("m", [1, 64, 128, 256])
def test_int4_gemm_consistent_with_reconstruction_at_13b_size(fmt, m):
    """The fused GEMM against a dense GEMM on the weights the reconstruction kernel produces (that
    kernel is bit-exact against the CPU restatement at small sizes), and exact scaling in A."""
    from splitwise_amd import _C
    torch.manual_seed(3)
    k, n, group = 5120, 13824, 128
    a = (torch.randn(m, k, device="cuda") * 0.5).to(torch.float16)
    scales = (torch.rand(k // group, n, device="cuda") * 2e-3 + 1e-4).to(torch.float16)
    qz = torch.randint(-2**31, 2**31 - 1, (k // group, n // 8), dtype=torch.int32, device="cuda")
    if fmt == "gptq":
        qw = torch.randint(-2**31, 2**31 - 1, (k // 8, n), dtype=torch.int32, device="cuda")
        _C.ops.gptq_shuffle(qw, torch.empty((1, 1), device="meta"), 4)
        none = torch.empty((1, 1), device="meta")
        run = lambda x: _C.ops.gptq_gemm(x, qw, qz, scales, none, True, 4)
        w = _C.ops.dequant_gptq(qw, qz, scales, None, 4, True)
    else:
        qw = torch.randint(-2**31, 2**31 - 1, (k, n // 8), dtype=torch.int32, device="cuda")
        run = lambda x: _C.ops.awq_gemm(x, qw, scales, qz, 8)
        w = _C.ops.awq_dequantize(qw, scales, qz, 0, 0, 0)
    out = run(a)
    torch.cuda.synchronize()
    ref = a.float() @ w.float()
    scale = ref.abs().max().item()
    assert (out.float() - ref).abs().max().item() <= 2e-3 * scale + 1e-3          # fp16 rounding of the result
    assert torch.equal(run(a), out)                                               # reproducible: no atomics on data
    # doubling A doubles the result exactly, wherever the result is a normal fp16 number (a
    # subnormal result has already lost the bit that its double keeps)
    twice, normal = run(a * 2), out.abs() >= 2.0 ** -13
    assert torch.equal(twice[normal], (out * 2)[normal])
    assert (twice.float() - 2 * out.float()).abs().max().item() <= 2.0 ** -24
