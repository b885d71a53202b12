"""Row-wise operators through the C ABI against the reference's own PyTorch-native ``_forward``
semantics (oracle/elementwise.py), with the tolerances the reference's kernel tests use:
layernorm 1e-2 / 1e-2 (tests/kernels/test_layernorm.py:50-53); activation and rotary the
dtype defaults of tests/kernels/allclose_default.py (fp16 1e-3/1e-3, bf16 1e-3/1.6e-2,
fp32 1e-5/1.3e-6)."""
import pytest
import torch

from oracle import elementwise as oe

pytestmark = pytest.mark.gpu

ATOL = {torch.float16: 1e-3, torch.bfloat16: 1e-3, torch.float32: 1e-5}
RTOL = {torch.float16: 1e-3, torch.bfloat16: 1.6e-2, torch.float32: 1.3e-6}
DTYPES = [torch.float16, torch.bfloat16, torch.float32]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tokens,hidden", [(7, 768), (83, 5120), (1, 8192), (256, 4096), (5, 8200)])
@pytest.mark.parametrize("add_residual", [False, True])
def test_rms_norm(dtype, tokens, hidden, add_residual):
    from splitwise_amd import _C
    torch.manual_seed(0)
    scale = 1 / (2 * hidden)
    x = (torch.randn(tokens, hidden) * scale).to(dtype)
    w = torch.empty(hidden).normal_(mean=1.0, std=0.1).to(dtype)
    res = (torch.randn(tokens, hidden) * scale).to(dtype) if add_residual else None
    ref = oe.rms_norm(x, w, 1e-6, res)
    xd, wd = x.cuda(), w.cuda()
    if add_residual:
        rd = res.cuda()
        _C.ops.fused_add_rms_norm(xd, rd, wd, 1e-6)
        torch.testing.assert_close(xd.cpu(), ref[0], atol=1e-2, rtol=1e-2)
        torch.testing.assert_close(rd.cpu(), ref[1], atol=1e-2, rtol=1e-2)
    else:
        out = torch.empty_like(xd)
        _C.ops.rms_norm(out, xd, wd, 1e-6)
        torch.testing.assert_close(out.cpu(), ref, atol=1e-2, rtol=1e-2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tokens,d", [(7, 512), (83, 13824), (2048, 5504), (1, 11008)])
def test_silu_and_mul(dtype, tokens, d):
    from splitwise_amd import _C
    torch.manual_seed(0)
    x = torch.randn(tokens, 2 * d).to(dtype)
    out = torch.empty(tokens, d, dtype=dtype, device="cuda")
    _C.ops.silu_and_mul(out, x.cuda())
    torch.testing.assert_close(out.cpu(), oe.silu_and_mul(x.float()).to(dtype), atol=ATOL[dtype], rtol=RTOL[dtype])


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("is_neox", [True, False])
@pytest.mark.parametrize("heads,kv_heads,head_size,rot_frac", [(32, 32, 128, 1.0), (16, 2, 128, 1.0), (12, 12, 64, 1.0),
                                                              (8, 8, 80, 1.0), (8, 4, 256, 0.5), (4, 4, 96, 0.5)])
def test_rotary_embedding(dtype, is_neox, heads, kv_heads, head_size, rot_frac):
    from splitwise_amd import _C
    torch.manual_seed(0)
    tokens, max_pos = 71, 8192
    rot = int(head_size * rot_frac)
    cache = oe.cos_sin_cache(rot, max_pos, 10000.0, dtype)
    positions = torch.randint(0, max_pos, (tokens,))
    # q and k are strided slices of one fused projection, as in the model
    qkv = torch.randn(tokens, (heads + 2 * kv_heads) * head_size).to(dtype)
    q_ref, k_ref = oe.rotary_embedding(positions, qkv[:, :heads * head_size].float(),
                                       qkv[:, heads * head_size:(heads + kv_heads) * head_size].float(), head_size,
                                       cache.float(), is_neox)
    dev = qkv.cuda()
    q = dev[:, :heads * head_size]
    k = dev[:, heads * head_size:(heads + kv_heads) * head_size]
    v_before = dev[:, (heads + kv_heads) * head_size:].clone()
    _C.ops.rotary_embedding(positions.cuda(), q, k, head_size, cache.cuda(), is_neox)
    torch.testing.assert_close(q.cpu().float(), q_ref, atol=ATOL[dtype], rtol=RTOL[dtype])
    torch.testing.assert_close(k.cpu().float(), k_ref, atol=ATOL[dtype], rtol=RTOL[dtype])
    assert torch.equal(dev[:, (heads + kv_heads) * head_size:], v_before)          # v is untouched
