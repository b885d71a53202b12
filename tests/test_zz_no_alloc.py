"""GPU parity of the allocation-free entry point sw_paged_attention_v2_static, through the C ABI.

It must give exactly what sw_paged_attention_v2 gives (same kernels, the temporaries in the library's
static device scratch instead of caller tensors), on two calls in a row that share the scratch, the
second on a smaller input; the first call is also held to the reference's tolerance against the
CPU oracle (tests/kernels/test_attention.py:262-269 of the reference: atol 1e-3, rtol 1e-5).
"""
import random

import pytest
import torch

from helpers import make_kv_caches
from oracle import attention as oattn

pytestmark = pytest.mark.gpu

PARTITION_SIZE = 512


def _inputs(num_seqs, h, kvh, d, bs, nb, max_len, dtype, seed, cache_dtype="auto"):
    random.seed(seed)
    torch.manual_seed(seed)
    scale = float(d ** -0.5)
    query = torch.empty(num_seqs, h, d, dtype=dtype).uniform_(-scale, scale)
    lens = [random.randint(1, max_len) for _ in range(num_seqs - 1)] + [max_len]
    max_blocks = (max_len + bs - 1) // bs
    tables = torch.tensor([[random.randint(0, nb - 1) for _ in range(max_blocks)] for _ in range(num_seqs)],
                          dtype=torch.int)
    kc, vc = make_kv_caches(nb, bs, kvh, d, dtype, seed=seed, cache_dtype=cache_dtype)
    return query, kc, vc, tables, torch.tensor(lens, dtype=torch.int), scale


def _v2(C, q, kc, vc, kvh, scale, tables, lens, bs, cache_dtype):
    s, h, d = q.shape
    max_ctx = int(lens.max().item())
    parts = (max_ctx + PARTITION_SIZE - 1) // PARTITION_SIZE
    # poisoned temporaries: the kernels must not depend on what they held
    tmp = torch.full((s, h, parts, d), float("nan"), dtype=q.dtype, device=q.device)
    es = torch.full((s, h, parts), float("nan"), dtype=torch.float32, device=q.device)
    ml = torch.full_like(es, float("nan"))
    out = torch.empty_like(q)
    C.ops.paged_attention_v2(out, es, ml, tmp, q, kc, vc, kvh, scale, tables, lens, bs, max_ctx, None, cache_dtype)
    return out


def _static(C, q, kc, vc, kvh, scale, tables, lens, bs, cache_dtype):
    out = torch.empty_like(q)
    C.ops.paged_attention_v2_static(out, q, kc, vc, kvh, scale, tables, lens, bs, int(lens.max().item()), None,
                                    cache_dtype)
    return out


@pytest.mark.parametrize("dtype,cache_dtype", [(torch.bfloat16, "auto"), (torch.float16, "fp8_e5m2")])
def test_static_scratch_matches_v2_on_two_calls(dtype, cache_dtype):
    from splitwise_amd import _C as C
    kvh, bs, nb = 4, 16, 512
    big = _inputs(24, 16, kvh, 128, bs, nb, 1700, dtype, seed=3, cache_dtype=cache_dtype)
    small = _inputs(5, 16, kvh, 128, bs, nb, 600, dtype, seed=4, cache_dtype=cache_dtype)
    first = True
    for query, kc, vc, tables, lens, scale in (big, small):
        dev = [t.cuda() for t in (query, kc, vc, tables, lens)]
        want = _v2(C, dev[0], dev[1], dev[2], kvh, scale, dev[3], dev[4], bs, cache_dtype)
        got = _static(C, dev[0], dev[1], dev[2], kvh, scale, dev[3], dev[4], bs, cache_dtype)
        torch.cuda.synchronize()
        assert torch.equal(got.view(torch.int16), want.view(torch.int16))
        if first:
            ref = oattn.paged_attention(query, kc, vc, tables, lens, scale, None)
            atol = 1e-2 if cache_dtype == "fp8_e5m2" else 1e-3
            torch.testing.assert_close(got.cpu().float(), ref.float(), atol=atol, rtol=1e-5)
            first = False


def test_static_scratch_rejects_what_does_not_fit():
    from splitwise_amd import _C as C
    assert C.load_library().sw_paged_attention_v2_static_max_bytes() == 32 << 20
    # 512 sequences x 64 heads x 8 partitions x 128 x 2 bytes = 64 MiB of partial outputs alone
    q = torch.zeros(512, 64, 128, dtype=torch.bfloat16, device="cuda")
    kc = torch.zeros(4, 8, 16, 16, 8, dtype=torch.bfloat16, device="cuda")
    vc = torch.zeros(4, 8, 128, 16, dtype=torch.bfloat16, device="cuda")
    tables = torch.zeros(512, 256, dtype=torch.int, device="cuda")
    lens = torch.full((512,), 4096, dtype=torch.int, device="cuda")
    with pytest.raises(RuntimeError, match="static scratch"):
        C.ops.paged_attention_v2_static(torch.empty_like(q), q, kc, vc, 8, 0.1, tables, lens, 16, 4096, None, "auto")
