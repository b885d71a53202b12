"""GPU parity: prompt attention over a cached prefix (sw_context_attention behind the reference's
``context_attention_fwd`` argument list) vs the CPU oracle.  Set-up of the reference's own test
(tests/kernels/test_prefix_prefill.py:31-118): random context and new-token lengths, a shuffled
block table, q / k / v ~ U(-1e-3, 1e-3), MHA and GQA, block size 32 (and 16); that test compares
with xformers (absent here), so the expected values come from the restatement in oracle/attention.py."""
import random

import pytest
import torch

from oracle import attention as oattn

pytestmark = pytest.mark.gpu


def _case(num_heads, num_queries_per_kv, head_size, dtype, block_size, new_lens, ctx_lens, scale_inputs=1e-3, seed=0):
    random.seed(seed)
    torch.manual_seed(seed)
    kvh = num_heads // num_queries_per_kv
    bs = len(new_lens)
    max_blocks = max(-(-c // block_size) for c in ctx_lens) + 2
    cache_blocks = bs * max_blocks + 7
    q = torch.empty(sum(new_lens), num_heads, head_size, dtype=dtype).uniform_(-scale_inputs, scale_inputs)
    k = torch.empty(sum(new_lens), kvh, head_size, dtype=dtype).uniform_(-scale_inputs, scale_inputs)
    v = torch.empty(sum(new_lens), kvh, head_size, dtype=dtype).uniform_(-scale_inputs, scale_inputs)
    # never-written slots hold NaN bit patterns: they must not leak into the result
    k_cache = torch.full((cache_blocks, kvh, head_size // 8, block_size, 8), float("nan"), dtype=dtype)
    v_cache = torch.full((cache_blocks, kvh, head_size, block_size), float("nan"), dtype=dtype)
    table = torch.randperm(cache_blocks)[:bs * max_blocks].view(bs, max_blocks).to(torch.int32)
    for i, ctx in enumerate(ctx_lens):
        kc = torch.empty(ctx, kvh, head_size, dtype=dtype).uniform_(-scale_inputs, scale_inputs)
        vc = torch.empty(ctx, kvh, head_size, dtype=dtype).uniform_(-scale_inputs, scale_inputs)
        for t in range(ctx):
            blk, slot = int(table[i, t // block_size]), t % block_size
            k_cache[blk, :, :, slot, :] = kc[t].view(kvh, head_size // 8, 8)
            v_cache[blk, :, :, slot] = vc[t]
    return q, k, v, k_cache, v_cache, table


def _run(q, k, v, k_cache, v_cache, table, new_lens, ctx_lens, alibi=None, window=None, num_parts=1):
    from splitwise_amd import _C
    starts, acc = [], 0
    for n in new_lens:
        starts.append(acc)
        acc += n
    dev = "cuda"
    out = torch.zeros_like(q, device=dev)
    seq_lens = [a + b for a, b in zip(new_lens, ctx_lens)]
    _C.ops.context_attention_fwd(q.to(dev), k.to(dev), v.to(dev), out, k_cache.to(dev), v_cache.to(dev), table.to(dev),
                                 torch.tensor(starts, dtype=torch.long, device=dev),
                                 torch.tensor(seq_lens, dtype=torch.long, device=dev),
                                 torch.tensor(ctx_lens, dtype=torch.long, device=dev), max(new_lens),
                                 alibi.to(dev) if alibi is not None else None, window, num_parts)
    torch.cuda.synchronize()
    return out.cpu()


@pytest.mark.parametrize("num_heads,num_queries_per_kv", [(64, 1), (64, 8), (12, 1), (12, 3)])
@pytest.mark.parametrize("head_size", [128, 64, 96])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_contexted_kv_attention(num_heads, num_queries_per_kv, head_size, dtype):
    random.seed(1)
    block_size = 32
    new_lens = [random.randint(16, 300) for _ in range(4)]
    ctx_lens = [random.randint(16, 400) for _ in range(4)]
    q, k, v, kc, vc, table = _case(num_heads, num_queries_per_kv, head_size, dtype, block_size, new_lens, ctx_lens)
    out = _run(q, k, v, kc, vc, table, new_lens, ctx_lens)
    ref = oattn.context_attention(q, k, v, kc, vc, table, new_lens, ctx_lens)
    assert not torch.isnan(out.float()).any()
    # the reference's bound on these inputs is atol 1e-6 (test_prefix_prefill.py:176); outputs are ~1e-5
    torch.testing.assert_close(out.float(), ref.float(), atol=1e-6, rtol=1e-2)


@pytest.mark.parametrize("head_size,block_size", [(80, 16), (112, 32), (128, 4), (64, 8), (256, 16)])
def test_both_context_paths(head_size, block_size):
    """The cached context goes through LDS, shared by the workgroup, when a block holds 8 consecutive keys of a
    head element and the head size is a multiple of 32; other shapes read it straight from the cache, wave by
    wave.  Both against the oracle, on caches whose unwritten slots hold NaN."""
    dtype = torch.bfloat16
    new_lens = [70, 1, 129, 40]
    ctx_lens = [130, 77, 0, 513]
    s = head_size ** -0.5
    q, k, v, kc, vc, table = _case(8, 4, head_size, dtype, block_size, new_lens, ctx_lens, scale_inputs=s, seed=7)
    out = _run(q, k, v, kc, vc, table, new_lens, ctx_lens)
    ref = oattn.context_attention(q, k, v, kc, vc, table, new_lens, ctx_lens)
    assert not torch.isnan(out.float()).any()
    torch.testing.assert_close(out.float(), ref.float(), atol=1e-3, rtol=1e-2)


@pytest.mark.parametrize("block_size", [4, 16])
def test_sliding_window_over_the_cached_context(block_size):
    """Windows that end inside the context, inside the new tokens, and beyond everything."""
    dtype, head = torch.float16, 128
    new_lens = [100, 33, 260]
    ctx_lens = [300, 64, 200]
    s = head ** -0.5
    q, k, v, kc, vc, table = _case(4, 2, head, dtype, block_size, new_lens, ctx_lens, scale_inputs=s, seed=9)
    for window in (50, 150, 1000):
        out = _run(q, k, v, kc, vc, table, new_lens, ctx_lens, window=window)
        ref = oattn.context_attention(q, k, v, kc, vc, table, new_lens, ctx_lens, None, window)
        assert not torch.isnan(out.float()).any()
        torch.testing.assert_close(out.float(), ref.float(), atol=1e-3, rtol=1e-3)


@pytest.mark.parametrize("block_size", [16, 32])
@pytest.mark.parametrize("use_alibi", [False, True])
def test_edge_lengths_alibi_and_unit_scale(block_size, use_alibi):
    """No context at all, context of one token, context ending on and off block boundaries, a single
    new token; inputs of the magnitude the model produces (tolerance of the prompt-attention tests)."""
    dtype = torch.bfloat16
    head = 128
    new_lens = [1, 33, 64, 5, 200, 31]
    ctx_lens = [64, 0, 1, 31, 333, 96]
    s = head ** -0.5
    q, k, v, kc, vc, table = _case(8, 2, head, dtype, block_size, new_lens, ctx_lens, scale_inputs=s, seed=3)
    alibi = torch.randn(8, dtype=torch.float) * 0.1 if use_alibi else None
    out = _run(q, k, v, kc, vc, table, new_lens, ctx_lens, alibi)
    ref = oattn.context_attention(q, k, v, kc, vc, table, new_lens, ctx_lens, alibi)
    assert not torch.isnan(out.float()).any()
    torch.testing.assert_close(out.float(), ref.float(), atol=1e-3, rtol=1e-5 if dtype == torch.float16 else 1e-2)


def test_chunked_prompt_equals_whole_prompt():
    """Size-independent property: running a prompt in two chunks (the first chunk written to the
    cache, the second attending to it) gives the second chunk's rows of the one-shot result."""
    from splitwise_amd import _C
    torch.manual_seed(5)
    dtype, h, kvh, d, bs = torch.float16, 16, 4, 128, 16
    total, first = 1000, 592
    s = d ** -0.5
    q = torch.empty(total, h, d, dtype=dtype, device="cuda").uniform_(-s, s)
    k = torch.empty(total, kvh, d, dtype=dtype, device="cuda").uniform_(-s, s)
    v = torch.empty(total, kvh, d, dtype=dtype, device="cuda").uniform_(-s, s)
    whole = torch.zeros_like(q)
    i32 = lambda xs: torch.tensor(xs, dtype=torch.int32, device="cuda")
    _C.ops.prefill_attention(whole, q, k, v, i32([0]), i32([total]), total, s)
    blocks = -(-first // bs)
    k_cache = torch.zeros(blocks + 3, kvh, d // 8, bs, 8, dtype=dtype, device="cuda")
    v_cache = torch.zeros(blocks + 3, kvh, d, bs, dtype=dtype, device="cuda")
    table = torch.randperm(blocks + 3)[:blocks].to(torch.int32)
    slots = torch.tensor([int(table[t // bs]) * bs + t % bs for t in range(first)], dtype=torch.long, device="cuda")
    _C.cache_ops.reshape_and_cache(k[:first], v[:first], k_cache, v_cache, slots, "auto")
    part = torch.zeros(total - first, h, d, dtype=dtype, device="cuda")
    _C.ops.context_attention_fwd(q[first:], k[first:], v[first:], part, k_cache, v_cache, table.view(1, -1).cuda(),
                                 torch.tensor([0], device="cuda"), torch.tensor([total], device="cuda"),
                                 torch.tensor([first], device="cuda"), total - first)
    torch.cuda.synchronize()
    torch.testing.assert_close(part.float(), whole[first:].float(), atol=2e-3, rtol=1e-3)


@pytest.mark.parametrize("seed", range(8))
def test_seeded_random_launches_over_a_cached_prefix(seed):
    """Random launches: 1-5 sequences, new-token runs of 1-400 tokens (query tiles are aligned to the end of each
    run), cached prefixes of 0-600 tokens, block sizes 8 / 16 / 32, grouped-query layouts, ALiBi or a window."""
    rng = random.Random(200 + seed)
    dtype = rng.choice([torch.float16, torch.bfloat16])
    head = rng.choice([64, 128, 128])
    block_size = rng.choice([8, 16, 32])
    per_kv = rng.choice([1, 2, 4])
    heads = per_kv * rng.choice([1, 2, 3])
    n = rng.randint(1, 5)
    pool = [1, 31, 32, 33, 64, 65, 127, 128, 129, 255, 257, 385]
    new_lens = [rng.choice(pool) if rng.random() < 0.6 else rng.randint(1, 400) for _ in range(n)]
    ctx_lens = [rng.choice([0, 1, 63, 64, 65, 200, 511]) if rng.random() < 0.6 else rng.randint(0, 600) for _ in range(n)]
    variant = rng.choice(["plain", "plain", "alibi", "window"])
    s = head ** -0.5
    q, k, v, kc, vc, table = _case(heads, per_kv, head, dtype, block_size, new_lens, ctx_lens, scale_inputs=s,
                                   seed=300 + seed)
    alibi = torch.randn(heads, dtype=torch.float) * 0.05 if variant == "alibi" else None
    window = rng.choice([40, 130, 500]) if variant == "window" else None
    out = _run(q, k, v, kc, vc, table, new_lens, ctx_lens, alibi, window)
    ref = oattn.context_attention(q, k, v, kc, vc, table, new_lens, ctx_lens, alibi, window)
    assert not torch.isnan(out.float()).any()
    torch.testing.assert_close(out.float(), ref.float(), atol=1e-3, rtol=1e-2)


@pytest.mark.parametrize("seed", range(10))
def test_split_form_agrees_with_the_oracle(seed):
    """The key stages of every (query tile, sequence, head) dealt out to 2-8 workgroups and joined by the reduce
    kernel: random launches as above (parts that get no stage at all, parts that hold only context, only new
    tokens, or the boundary between the two), ALiBi or a window, against the oracle and against the unsplit form."""
    rng = random.Random(900 + seed)
    dtype = rng.choice([torch.float16, torch.bfloat16])
    head = rng.choice([64, 128, 128])
    block_size = rng.choice([8, 16, 32])
    per_kv = rng.choice([1, 2, 4])
    heads = per_kv * rng.choice([1, 2, 3])
    n = rng.randint(1, 4)
    new_lens = [rng.choice([1, 31, 64, 65, 128, 129, 257]) if rng.random() < 0.5 else rng.randint(1, 300) for _ in range(n)]
    ctx_lens = [rng.choice([0, 1, 63, 64, 65, 511]) if rng.random() < 0.4 else rng.randint(0, 1500) for _ in range(n)]
    variant = rng.choice(["plain", "plain", "alibi", "window"])
    parts = rng.choice([2, 3, 5, 8])
    s = head ** -0.5
    q, k, v, kc, vc, table = _case(heads, per_kv, head, dtype, block_size, new_lens, ctx_lens, scale_inputs=s,
                                   seed=400 + seed)
    alibi = torch.randn(heads, dtype=torch.float) * 0.05 if variant == "alibi" else None
    window = rng.choice([40, 130, 700]) if variant == "window" else None
    out = _run(q, k, v, kc, vc, table, new_lens, ctx_lens, alibi, window, num_parts=parts)
    whole = _run(q, k, v, kc, vc, table, new_lens, ctx_lens, alibi, window, num_parts=1)
    ref = oattn.context_attention(q, k, v, kc, vc, table, new_lens, ctx_lens, alibi, window)
    assert not torch.isnan(out.float()).any()
    torch.testing.assert_close(out.float(), ref.float(), atol=1e-3, rtol=1e-2)
    torch.testing.assert_close(out.float(), whole.float(), atol=2e-4, rtol=1e-2)     # one more fp32 summation apart


def test_split_form_is_chosen_for_few_new_tokens_over_a_long_context_and_refused_for_odd_shapes():
    from splitwise_amd import _C
    lib = _C.load_library()
    # 512 new tokens on a 3584-token context, 32 heads: 128 workgroups -> 4 parts; a full launch is left alone
    assert lib.sw_context_attention_split_parts(1, 512, 32, 128, 16, 3584) == 4
    assert lib.sw_context_attention_split_parts(1, 64, 32, 128, 16, 8192) == 8
    assert lib.sw_context_attention_split_parts(2, 512, 32, 128, 16, 3584) == 2
    assert lib.sw_context_attention_split_parts(4, 512, 32, 128, 16, 3584) == 1
    assert lib.sw_context_attention_split_parts(1, 512, 32, 128, 16, 512) == 1
    assert lib.sw_context_attention_split_parts(1, 512, 32, 80, 16, 3584) == 1      # head sizes off multiples of 32
    assert lib.sw_context_attention_split_parts(1, 512, 32, 128, 4, 3584) == 1      # no 8 consecutive keys per block
    # the library's own choice (num_parts=None) on such a launch, against the oracle
    dtype, head, bs = torch.bfloat16, 128, 16
    new_lens, ctx_lens = [200], [1500]
    q, k, v, kc, vc, table = _case(8, 2, head, dtype, bs, new_lens, ctx_lens, scale_inputs=head ** -0.5, seed=11)
    assert lib.sw_context_attention_split_parts(1, 200, 8, head, bs, table.shape[1] * bs) > 1
    out = _run(q, k, v, kc, vc, table, new_lens, ctx_lens, num_parts=None)
    ref = oattn.context_attention(q, k, v, kc, vc, table, new_lens, ctx_lens)
    torch.testing.assert_close(out.float(), ref.float(), atol=1e-3, rtol=1e-2)
    # shapes the split form does not take are refused, not silently run unsplit
    q, k, v, kc, vc, table = _case(4, 2, 80, dtype, 16, [40], [300], scale_inputs=80 ** -0.5, seed=12)
    with pytest.raises(RuntimeError, match="cannot be split"):
        _run(q, k, v, kc, vc, table, [40], [300], num_parts=2)
