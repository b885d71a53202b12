"""GPU parity of the fused decode-step operator: rotary embedding + cache write in one launch.

Bar: bit-exact against the two separate operators (each of which has its own oracle test:
tests/test_elementwise_gpu.py, tests/test_cache_kernels_gpu.py), and the cache contents bit-exact
against the CPU oracle's reshape_and_cache of the rotated key.  Reference semantics:
csrc/pos_encoding_kernels.cu:120-171 followed by csrc/cache_kernels.cu:228-268.
"""
import random

import pytest
import torch

from helpers import make_kv_caches
from oracle import cache_ops as ocache

pytestmark = pytest.mark.gpu


def _bits(t):
    return t.contiguous().view(torch.uint8) if t.dtype != torch.uint8 else t


def _cos_sin(max_pos, rot_dim, dtype):
    inv = 1.0 / (10000.0 ** (torch.arange(0, rot_dim, 2, dtype=torch.float) / rot_dim))
    f = torch.einsum("i,j->ij", torch.arange(max_pos, dtype=torch.float), inv)
    return torch.cat((f.cos(), f.sin()), dim=-1).to(dtype)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("kv_cache_dtype", ["auto", "fp8_e5m2"])
@pytest.mark.parametrize("is_neox", [True, False])
@pytest.mark.parametrize("heads,kv_heads,head_size,rot_dim,block_size",
                         [(32, 32, 128, 128, 16), (16, 2, 128, 128, 16), (12, 4, 64, 32, 32), (5, 5, 80, 64, 8),
                          (8, 1, 256, 128, 16), (7, 7, 96, 96, 16)])
def test_matches_the_two_operators(dtype, kv_cache_dtype, is_neox, heads, kv_heads, head_size, rot_dim, block_size):
    from splitwise_amd import _C
    random.seed(1)
    torch.manual_seed(1)
    num_blocks, tokens, max_pos = 64, 37, 512
    slots = random.sample(range(num_blocks * block_size), tokens)
    slots[5] = -1                                        # a padding row: rotated, not cached
    slots[20] = -1
    slot_mapping = torch.tensor(slots, dtype=torch.long, device="cuda")
    positions = torch.randint(0, max_pos, (tokens,), device="cuda")
    width = (heads + 2 * kv_heads) * head_size
    qkv = torch.randn(tokens, width, dtype=dtype, device="cuda")     # q, k, v are column slices of one tensor
    cos_sin = _cos_sin(max_pos, rot_dim, dtype).cuda()
    kc, vc = make_kv_caches(num_blocks, block_size, kv_heads, head_size, dtype, cache_dtype=kv_cache_dtype, device="cuda")

    def split(t):
        return t[:, :heads * head_size], t[:, heads * head_size:(heads + kv_heads) * head_size], \
            t[:, (heads + kv_heads) * head_size:]

    ref = qkv.clone()
    q_r, k_r, v_r = split(ref)
    kc_r, vc_r = kc.clone(), vc.clone()
    _C.ops.rotary_embedding(positions, q_r, k_r, head_size, cos_sin, is_neox)
    _C.cache_ops.reshape_and_cache(k_r.view(tokens, kv_heads, head_size), v_r.view(tokens, kv_heads, head_size), kc_r,
                                   vc_r, slot_mapping, kv_cache_dtype)

    got = qkv.clone()
    q_g, k_g, v_g = split(got)
    kc_g, vc_g = kc.clone(), vc.clone()
    _C.ops.rotary_embedding_and_cache(positions, q_g, k_g, v_g, kc_g, vc_g, slot_mapping, head_size, cos_sin, is_neox,
                                      kv_cache_dtype)
    torch.cuda.synchronize()
    assert torch.equal(_bits(got), _bits(ref))           # q and k rotated in place, v untouched
    assert torch.equal(_bits(kc_g), _bits(kc_r))
    assert torch.equal(_bits(vc_g), _bits(vc_r))

    # and straight against the CPU oracle's cache write of the rotated key
    kc_o, vc_o = kc.cpu(), vc.cpu()
    ocache.reshape_and_cache(k_g.cpu().view(tokens, kv_heads, head_size), v_g.cpu().view(tokens, kv_heads, head_size),
                             kc_o, vc_o, slot_mapping.cpu(), kv_cache_dtype)
    assert torch.equal(_bits(kc_g.cpu()), _bits(kc_o))
    assert torch.equal(_bits(vc_g.cpu()), _bits(vc_o))


def test_decode_batch_of_the_bench_shape():
    """256 sequences x 32 heads x 128, bf16, distinct blocks: the decode step of the 7B configuration."""
    from splitwise_amd import _C
    torch.manual_seed(2)
    tokens, heads, head_size, block_size, num_blocks = 256, 32, 128, 16, 2048
    slot_mapping = (torch.randperm(num_blocks, device="cuda")[:tokens] * block_size +
                    torch.randint(0, block_size, (tokens,), device="cuda")).long()
    positions = torch.randint(0, 4096, (tokens,), device="cuda")
    qkv = torch.randn(tokens, 3 * heads * head_size, dtype=torch.bfloat16, device="cuda")
    cos_sin = _cos_sin(4096, head_size, torch.bfloat16).cuda()
    kc, vc = make_kv_caches(num_blocks, block_size, heads, head_size, torch.bfloat16, device="cuda")
    ref, got = qkv.clone(), qkv.clone()
    kc_r, vc_r, kc_g, vc_g = kc.clone(), vc.clone(), kc.clone(), vc.clone()
    d = heads * head_size
    _C.ops.rotary_embedding(positions, ref[:, :d], ref[:, d:2 * d], head_size, cos_sin, True)
    _C.cache_ops.reshape_and_cache(ref[:, d:2 * d].view(tokens, heads, head_size),
                                   ref[:, 2 * d:].view(tokens, heads, head_size), kc_r, vc_r, slot_mapping, "auto")
    _C.ops.rotary_embedding_and_cache(positions, got[:, :d], got[:, d:2 * d], got[:, 2 * d:], kc_g, vc_g, slot_mapping,
                                      head_size, cos_sin, True, "auto")
    torch.cuda.synchronize()
    assert torch.equal(got, ref) and torch.equal(kc_g, kc_r) and torch.equal(vc_g, vc_r)


def test_empty_and_rejects():
    from splitwise_amd import _C
    kc, vc = make_kv_caches(4, 16, 2, 64, torch.float16, device="cuda")
    cos_sin = _cos_sin(16, 64, torch.float16).cuda()
    e = torch.empty(0, 128, dtype=torch.float16, device="cuda")
    none = torch.empty(0, dtype=torch.long, device="cuda")
    _C.ops.rotary_embedding_and_cache(none, e, e, e, kc, vc, none, 64, cos_sin, True, "auto")
    x = torch.zeros(1, 128, dtype=torch.float16, device="cuda")
    one = torch.zeros(1, dtype=torch.long, device="cuda")
    with pytest.raises(RuntimeError, match="Unsupported data type of kv cache"):
        _C.ops.rotary_embedding_and_cache(one, x, x.clone(), x.clone(), kc, vc, one, 64, cos_sin, True, "int4")
    with pytest.raises(RuntimeError, match="token count"):
        _C.ops.rotary_embedding_and_cache(one, x, x.clone(), x.clone(), kc, vc, none, 64, cos_sin, True, "auto")
    # rot_dim 12 on f16: neither half (6 elements) nor the rest spans whole 16-byte vectors
    with pytest.raises(RuntimeError, match="16-byte vectors"):
        _C.ops.rotary_embedding_and_cache(one, x, x.clone(), x.clone(), kc, vc, one, 64,
                                          _cos_sin(16, 12, torch.float16).cuda(), True, "auto")
