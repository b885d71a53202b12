"""GPU parity: prefill (prompt-run) attention through the C ABI vs the CPU oracle and the
fixtures computed by the reference's ref_multi_query_kv_attention
(tests/kernels/test_attention.py:272-372; tolerance atol 1e-3 rtol 1e-5 from :370-372)."""
import glob
import os
import random

import pytest
import torch

from helpers import DTYPES, GOLDEN, from_np, load_golden
from oracle import attention as oattn

pytestmark = pytest.mark.gpu
PREFILL = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "prefill_attention_*.npz")))


def run(q, k, v, seq_lens, scale, alibi=None, starts=None, window=None):
    from splitwise_amd import _C
    if starts is None:
        starts, acc = [], 0
        for n in seq_lens:
            starts.append(acc)
            acc += n
    out = torch.zeros_like(q)
    _C.ops.prefill_attention(out, q, k, v, torch.tensor(starts, dtype=torch.int32, device="cuda"),
                             torch.tensor(seq_lens, dtype=torch.int32, device="cuda"), max(seq_lens), scale,
                             alibi, window)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("name", PREFILL)
def test_against_reference_fixture(name):
    z = load_golden(name)
    dtype = DTYPES[str(z["dtype"])]
    q, k, v = (from_np(z[n], dtype).cuda() for n in ("query", "key", "value"))
    out = run(q, k, v, z["seq_lens"].tolist(), float(z["scale"]))
    torch.testing.assert_close(out.cpu().float(), from_np(z["out"], dtype).float(), atol=1e-3, rtol=1e-5)


@pytest.mark.parametrize("num_heads", [(40, 40), (64, 8)])
@pytest.mark.parametrize("head_size", [64, 80, 96, 112, 128, 256])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("use_alibi", [False, True])
def test_prefill_attention(num_heads, head_size, dtype, use_alibi):
    random.seed(0)
    torch.manual_seed(0)
    h, kvh = num_heads
    seq_lens = random.sample(range(1, 700), 3)
    n = sum(seq_lens)
    scale = float(1.0 / head_size ** 0.5)
    # q, k, v are slices of one fused tensor, as in the model (strided rows)
    qkv = torch.empty(n, h + 2 * kvh, head_size, dtype=dtype).uniform_(-scale, scale)
    q, k, v = qkv.split([h, kvh, kvh], dim=1)
    alibi = torch.randn(h, dtype=torch.float) * 0.1 if use_alibi else None
    ref = oattn.prefill_attention(q, k, v, seq_lens, scale, alibi)
    qkv_d = qkv.cuda()
    qd, kd, vd = qkv_d.split([h, kvh, kvh], dim=1)
    out = run(qd, kd, vd, seq_lens, scale, alibi.cuda() if use_alibi else None)
    torch.testing.assert_close(out.cpu().float(), ref.float(), atol=1e-3, rtol=1e-5)


def test_padded_batch_layout_and_edge_lengths():
    """The reference pads prompts to [B, max_len] (model_runner.py:216-232): sequence i starts at
    row i * max_len; rows past a prompt's length must stay untouched."""
    torch.manual_seed(0)
    dtype, h, kvh, d = torch.bfloat16, 8, 2, 128
    seq_lens = [1, 32, 33, 129, 128, 31]
    max_len = max(seq_lens)
    scale = d ** -0.5
    q = torch.empty(len(seq_lens) * max_len, h, d, dtype=dtype).uniform_(-scale, scale)
    k = torch.empty(len(seq_lens) * max_len, kvh, d, dtype=dtype).uniform_(-scale, scale)
    v = torch.empty_like(k).uniform_(-scale, scale)
    starts = [i * max_len for i in range(len(seq_lens))]
    out = run(q.cuda(), k.cuda(), v.cuda(), seq_lens, scale, starts=starts).cpu()
    for i, n in enumerate(seq_lens):
        s = starts[i]
        ref = oattn.prefill_attention(q[s:s + n], k[s:s + n], v[s:s + n], [n], scale)
        torch.testing.assert_close(out[s:s + n].float(), ref.float(), atol=1e-3, rtol=1e-5)
        assert torch.count_nonzero(out[s + n:s + max_len]) == 0


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("variant", ["plain", "alibi", "window"])
def test_long_prompts_take_the_256_row_workgroups(dtype, variant):
    """Launches with a prompt of 2048 tokens or more and at least 384 tiles of 256 query rows run workgroups of
    8 waves sharing each K / V stage (here 8 tiles x 12 heads x 4 prompts): lengths around the 256-row and 64-key
    boundaries beside a short prompt in the same launch, with ALiBi and with a sliding window."""
    torch.manual_seed(1)
    h, kvh, d = 12, 4, 128
    seq_lens = [1601, 37, 2048, 1537]
    n = sum(seq_lens)
    scale = float(d ** -0.5)
    qkv = torch.empty(n, h + 2 * kvh, d, dtype=dtype).uniform_(-scale, scale)
    q, k, v = qkv.split([h, kvh, kvh], dim=1)
    alibi = torch.randn(h, dtype=torch.float) * 0.05 if variant == "alibi" else None
    window = 700 if variant == "window" else None
    ref = oattn.prefill_attention(q, k, v, seq_lens, scale, alibi, window)
    qd, kd, vd = qkv.cuda().split([h, kvh, kvh], dim=1)
    out = run(qd, kd, vd, seq_lens, scale, alibi.cuda() if alibi is not None else None, window=window)
    torch.testing.assert_close(out.cpu().float(), ref.float(), atol=1e-3, rtol=1e-5)


@pytest.mark.parametrize("seed", range(12))
def test_seeded_random_ragged_launches(seed):
    """Random launches: 1-7 prompts of 1-900 tokens (lengths around the 128-row tiles, the 64-key stages and
    their remainders), grouped-query layouts, head sizes, ALiBi or a sliding window -- every query tile is aligned
    to the end of its prompt, so each length exercises another split of rows into tiles."""
    rng = random.Random(100 + seed)
    torch.manual_seed(seed)
    dtype = rng.choice([torch.float16, torch.bfloat16])
    head_size = rng.choice([64, 128, 128, 96])
    kvh = rng.choice([1, 2, 4])
    h = kvh * rng.choice([1, 2, 4])
    pool = [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 191, 192, 193, 255, 256, 257, 384, 385, 511, 513, 640, 772, 900]
    seq_lens = [rng.choice(pool) if rng.random() < 0.6 else rng.randint(1, 900) for _ in range(rng.randint(1, 7))]
    variant = rng.choice(["plain", "plain", "alibi", "window"])
    n = sum(seq_lens)
    scale = float(head_size ** -0.5)
    qkv = torch.empty(n, h + 2 * kvh, head_size, dtype=dtype).uniform_(-scale, scale)
    q, k, v = qkv.split([h, kvh, kvh], dim=1)
    alibi = torch.randn(h, dtype=torch.float) * 0.05 if variant == "alibi" else None
    window = rng.choice([17, 64, 100, 300]) if variant == "window" else None
    ref = oattn.prefill_attention(q, k, v, seq_lens, scale, alibi, window)
    qd, kd, vd = qkv.cuda().split([h, kvh, kvh], dim=1)
    out = run(qd, kd, vd, seq_lens, scale, alibi.cuda() if alibi is not None else None, window=window)
    torch.testing.assert_close(out.cpu().float(), ref.float(), atol=1e-3, rtol=1e-5)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("variant", ["plain", "alibi", "window"])
@pytest.mark.parametrize("seq_lens", [[772], [772, 772], [4], [65, 130]])
def test_both_sides_of_the_paired_waves_rule(dtype, variant, seq_lens):
    """A launch of at most 256 workgroups of 128 rows (one 772-token prompt of 32 heads is 224) runs pairs of waves
    that share 32 query rows and split every stage's keys, joined through LDS at the end; two such prompts (448) run
    the plain form.  Same oracle, same tolerance, on both sides of the rule."""
    torch.manual_seed(3)
    h, kvh, d = 32, 8, 128
    n = sum(seq_lens)
    scale = float(d ** -0.5)
    qkv = torch.empty(n, h + 2 * kvh, d, dtype=dtype).uniform_(-scale, scale)
    q, k, v = qkv.split([h, kvh, kvh], dim=1)
    alibi = torch.randn(h, dtype=torch.float) * 0.05 if variant == "alibi" else None
    window = 200 if variant == "window" else None
    ref = oattn.prefill_attention(q, k, v, seq_lens, scale, alibi, window)
    qd, kd, vd = qkv.cuda().split([h, kvh, kvh], dim=1)
    out = run(qd, kd, vd, seq_lens, scale, alibi.cuda() if alibi is not None else None, window=window)
    torch.testing.assert_close(out.cpu().float(), ref.float(), atol=1e-3, rtol=1e-5)


def test_float32_is_rejected():
    from splitwise_amd import _C
    q = torch.zeros(4, 2, 64, device="cuda")
    with pytest.raises(RuntimeError, match="Unsupported data type"):
        _C.ops.prefill_attention(torch.empty_like(q), q, q, q, torch.zeros(1, dtype=torch.int32, device="cuda"),
                                 torch.full((1,), 4, dtype=torch.int32, device="cuda"), 4, 1.0)
