"""GPU parity: cache kernels through the C ABI vs the CPU oracle (bit-exact).

Mirrors tests/kernels/test_cache.py of the reference (copy_blocks :41-98,
reshape_and_cache :110-161, swap_blocks :174-226) plus the fp8 paths.
"""
import random

import pytest
import torch

from helpers import make_kv_caches
from oracle import cache_ops as ocache
from oracle import fp8 as ofp8

pytestmark = pytest.mark.gpu

DTYPES = [torch.float16, torch.bfloat16, torch.float32]


def _ops():
    from splitwise_amd import _C
    return _C


def _bits(t):
    return t.contiguous().view(torch.uint8) if t.dtype != torch.uint8 else t


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kv_cache_dtype", ["auto", "fp8_e5m2"])
@pytest.mark.parametrize("num_heads,head_size,block_size", [(8, 128, 16), (5, 80, 32), (3, 64, 8), (2, 256, 16)])
@pytest.mark.parametrize("pattern", ["prompt", "decode", "padded"])
def test_reshape_and_cache(dtype, kv_cache_dtype, num_heads, head_size, block_size, pattern):
    C = _ops()
    random.seed(0)
    torch.manual_seed(0)
    num_blocks = 96
    num_tokens = {"prompt": 83, "decode": 42, "padded": 70}[pattern]
    if pattern == "prompt":          # consecutive slots from a block boundary, like one prompt
        blocks = random.sample(range(num_blocks), (num_tokens + block_size - 1) // block_size)
        slots = [blocks[i // block_size] * block_size + i % block_size for i in range(num_tokens)]
    elif pattern == "decode":        # unrelated slots
        slots = random.sample(range(num_blocks * block_size), num_tokens)
    else:                            # two prompts padded to a common length with -1
        slots = []
        for start, n in ((3 * block_size, 29), (40 * block_size, 35)):
            slots += [start + i for i in range(n)] + [-1] * (35 - n)
    slot_mapping = torch.tensor(slots, dtype=torch.long)
    # key / value are slices of a fused qkv tensor: row stride != num_heads * head_size
    qkv = torch.randn(num_tokens, 3, num_heads, head_size, dtype=dtype)
    key, value = qkv[:, 1], qkv[:, 2]
    kc, vc = make_kv_caches(num_blocks, block_size, num_heads, head_size, dtype, cache_dtype=kv_cache_dtype)
    kc_ref, vc_ref = kc.clone(), vc.clone()
    ocache.reshape_and_cache(key, value, kc_ref, vc_ref, slot_mapping, kv_cache_dtype)

    qkv_d = qkv.cuda()
    kc_d, vc_d = kc.cuda(), vc.cuda()
    C.cache_ops.reshape_and_cache(qkv_d[:, 1], qkv_d[:, 2], kc_d, vc_d, slot_mapping.cuda(), kv_cache_dtype)
    torch.cuda.synchronize()
    assert torch.equal(_bits(kc_d.cpu()), _bits(kc_ref))
    assert torch.equal(_bits(vc_d.cpu()), _bits(vc_ref))


def test_reshape_and_cache_empty_and_errors():
    C = _ops()
    kc, vc = make_kv_caches(4, 16, 2, 64, torch.float16, device="cuda")
    key = torch.empty(0, 2, 64, dtype=torch.float16, device="cuda")
    C.cache_ops.reshape_and_cache(key, key, kc, vc, torch.empty(0, dtype=torch.long, device="cuda"), "auto")
    key = torch.zeros(1, 2, 64, dtype=torch.float16, device="cuda")
    with pytest.raises(RuntimeError, match="Unsupported data type of kv cache"):
        C.cache_ops.reshape_and_cache(key, key, kc, vc, torch.zeros(1, dtype=torch.long, device="cuda"), "int4")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32, torch.uint8])
@pytest.mark.parametrize("num_layers", [1, 5])
def test_copy_blocks(dtype, num_layers):
    C = _ops()
    random.seed(0)
    num_blocks, block_size, heads, head = 64, 16, 4, 80
    cache_dtype = "fp8_e5m2" if dtype == torch.uint8 else "auto"
    act = torch.float16 if dtype == torch.uint8 else dtype
    kcs, vcs = [], []
    for layer in range(num_layers):
        k, v = make_kv_caches(num_blocks, block_size, heads, head, act, seed=layer, cache_dtype=cache_dtype)
        kcs.append(k)
        vcs.append(v)
    # every source goes to two destinations, like the reference test (:61-68)
    srcs = random.sample(range(num_blocks), 10)
    rest = random.sample(sorted(set(range(num_blocks)) - set(srcs)), 20)
    mapping = {s: [rest[2 * i], rest[2 * i + 1]] for i, s in enumerate(srcs)}
    kd = [k.cuda() for k in kcs]
    vd = [v.cuda() for v in vcs]
    C.cache_ops.copy_blocks(kd, vd, mapping)
    ocache.copy_blocks(kcs, vcs, mapping)
    torch.cuda.synchronize()
    for a, b in zip(kd + vd, kcs + vcs):
        assert torch.equal(_bits(a.cpu()), _bits(b))
    C.cache_ops.copy_blocks(kd, vd, {})          # empty mapping is a no-op
    C.cache_ops.copy_blocks([], [], {1: [2]})    # no layers is a no-op


@pytest.mark.parametrize("direction", [("cuda", "cpu"), ("cpu", "cuda"), ("cuda", "cuda")])
def test_swap_blocks(direction):
    C = _ops()
    random.seed(0)
    src_dev, dst_dev = direction
    num_blocks, block_size, heads, head = 48, 16, 4, 64
    sk, sv = make_kv_caches(num_blocks, block_size, heads, head, torch.float16, seed=1)
    dk, dv = make_kv_caches(num_blocks, block_size, heads, head, torch.float16, seed=2)
    src_blocks = random.sample(range(num_blocks), 14) + [30, 31, 32]     # includes a consecutive run
    src_blocks = list(dict.fromkeys(src_blocks))
    dst_blocks = random.sample(range(num_blocks), len(src_blocks))
    mapping = dict(zip(src_blocks, dst_blocks))

    def place(t, dev):
        return t.cuda() if dev == "cuda" else t.clone().pin_memory()

    for s, d in ((sk, dk), (sv, dv)):
        s_dev, d_dev = place(s, src_dev), place(d, dst_dev)
        C.cache_ops.swap_blocks(s_dev, d_dev, mapping)
        torch.cuda.synchronize()
        expect = d.clone()
        ocache.swap_blocks(s, expect, mapping)
        assert torch.equal(d_dev.cpu(), expect)
    with pytest.raises(RuntimeError, match="Invalid device combination"):
        C.cache_ops.swap_blocks(sk, dk, mapping)


@pytest.mark.parametrize("dtype", DTYPES)
def test_convert_fp8_e5m2(dtype):
    C = _ops()
    torch.manual_seed(0)
    n = 4 * 1024 * 37 + 5                      # not a multiple of the vector width
    x = torch.cat([torch.randn(n - 12) * 3, torch.tensor(
        [0.0, -0.0, 1e6, -1e6, float("inf"), float("-inf"), 57344.0, 61440.0, 6.1e-5, 1.6e-5, 7.7e-6, 3e-6])])
    x = x.to(dtype)
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    C.cache_ops.convert_fp8_e5m2(x.cuda(), out)
    assert torch.equal(out.cpu(), ofp8.to_e5m2(x))
    back = torch.empty(n, dtype=dtype, device="cuda")
    C.cache_ops.convert_fp8_e5m2(out, back)
    assert torch.equal(_bits(back.cpu()), _bits(ofp8.from_e5m2(out.cpu(), dtype)))
    # all 256 codes decode exactly (NaN payloads aside: compare through isnan)
    codes = torch.arange(256, dtype=torch.uint8).repeat(8)
    dec = torch.empty(codes.numel(), dtype=dtype, device="cuda")
    C.cache_ops.convert_fp8_e5m2(codes.cuda(), dec)
    ref = ofp8.from_e5m2(codes, dtype)
    nan = torch.isnan(ref)
    assert torch.equal(torch.isnan(dec.cpu()), nan)
    assert torch.equal(_bits(dec.cpu()[~nan]), _bits(ref[~nan]))
