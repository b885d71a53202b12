"""Self-consistency of the int4 format restatement (no reference fixture exists for 4-bit GPTQ
or AWQ: parity unpinned, see oracle/quant.py)."""
import numpy as np
import pytest

from oracle import quant


def _gptq_case(k, n, group, seed, act_order=False):
    rng = np.random.default_rng(seed)
    q = rng.integers(0, 16, size=(k, n))
    groups = k // group
    zeros = rng.integers(1, 17, size=(groups, n))
    scales = (rng.uniform(0.5, 1.5, size=(groups, n)) * 1e-2).astype(np.float16)
    g_idx = np.arange(k) // group
    if act_order:
        g_idx = rng.permutation(g_idx)
    qweight, qzeros = quant.gptq_pack(q, zeros)
    w = ((q - zeros[g_idx]).astype(np.float16) * scales[g_idx])
    return q, zeros, scales, g_idx.astype(np.int32), qweight, qzeros, w


@pytest.mark.parametrize("act_order", [False, True])
def test_gptq_plain_reader_recovers_the_weights(act_order):
    q, zeros, scales, g_idx, qweight, qzeros, w = _gptq_case(256, 64, 32, 0, act_order)
    got = quant.gptq_dequant(qweight, qzeros, scales, g_idx, use_exllama=False)
    assert got.dtype == np.float16
    assert np.array_equal(got.view(np.uint16), w.view(np.uint16))


def test_gptq_shuffle_known_word():
    # rows 0..7 hold 0..7: natural word 0x76543210 -> low half rows 0,2,4,6, high half rows 1,3,5,7
    qweight = np.array([[0x76543210]], dtype=np.uint32).view(np.int32)
    assert quant.gptq_shuffle(qweight).view(np.uint32)[0, 0] == 0x75316420


def test_gptq_shuffled_reader_equals_plain_reader():
    q, zeros, scales, g_idx, qweight, qzeros, w = _gptq_case(256, 64, 64, 1)
    shuffled = quant.gptq_shuffle(qweight)
    got = quant.gptq_dequant(shuffled, qzeros, scales, None, use_exllama=True)
    assert np.array_equal(got.view(np.uint16), w.view(np.uint16))


def test_gptq_act_order_shuffle_restores_original_rows():
    q, zeros, scales, g_idx, qweight, qzeros, w = _gptq_case(256, 64, 32, 2, act_order=True)
    perm = np.argsort(g_idx, kind="stable").astype(np.int32)          # gptq.py:206-209
    shuffled = quant.gptq_shuffle(qweight, perm)
    got = quant.gptq_dequant(shuffled, qzeros, scales, perm, use_exllama=True)
    assert np.array_equal(got.view(np.uint16), w.view(np.uint16))
    a = np.random.default_rng(3).standard_normal((5, 256)).astype(np.float16)
    ref = a.astype(np.float32) @ w.astype(np.float32)
    np.testing.assert_allclose(quant.gptq_gemm(a, shuffled, qzeros, scales, perm, True), ref, rtol=1e-6, atol=1e-6)


def test_awq_pack_and_dequantize_round_trip():
    rng = np.random.default_rng(4)
    k, n, group = 128, 64, 32
    q = rng.integers(0, 16, size=(k, n))
    zeros = rng.integers(0, 16, size=(k // group, n))
    scales = (rng.uniform(0.5, 1.5, size=(k // group, n)) * 1e-2).astype(np.float16)
    qweight, qzeros = quant.awq_pack(q, zeros)
    grp = np.arange(k) // group
    want = (q - zeros[grp]).astype(np.float16) * scales[grp]
    got = quant.awq_dequantize(qweight, scales, qzeros)
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))


def test_awq_nibble_order_known_word():
    # logical columns 0..7 hold 0..7: column c sits in nibble [0,4,1,5,2,6,3,7][c]
    qweight, _ = quant.awq_pack(np.arange(8).reshape(1, 8), np.zeros((1, 8), dtype=np.int64))
    assert qweight.view(np.uint32)[0, 0] == 0x75316420


def test_gptq_8bit_pack_shuffle_dequant_round_trip():
    """8-bit GPTQ restatement (qdq_8.cuh, q_gemm.cu:810-918, :1989-2029): packing then reconstruction gives
    (q - zero) * scale with the zero point stored minus one, the "shuffle" leaves the word order alone, and the
    act-order form (rows gathered by the permutation, reconstruction scattered back) equals the plain form."""
    import numpy as np
    from oracle import quant
    rng = np.random.default_rng(0)
    k, n, group = 128, 24, 32
    q = rng.integers(0, 256, size=(k, n))
    zeros = rng.integers(1, 257, size=(k // group, n))
    scales = (rng.uniform(0.5, 1.5, size=(k // group, n)) * 1e-3).astype(np.float16)
    qweight, qzeros = quant.gptq8_pack(q, zeros)
    assert qweight.shape == (k // 4, n) and qzeros.shape == (k // group, n // 4)
    grp = np.arange(k) // group
    want = (q - zeros[grp]).astype(np.float16) * scales[grp]
    assert np.array_equal(quant.gptq8_dequant(quant.gptq8_shuffle(qweight), qzeros, scales, None, True), want)
    g_idx = rng.permutation(grp).astype(np.int32)
    plain = quant.gptq8_dequant(qweight, qzeros, scales, g_idx, False)
    perm = np.argsort(g_idx, kind="stable")
    assert np.array_equal(quant.gptq8_dequant(quant.gptq8_shuffle(qweight, perm), qzeros, scales, perm, True), plain)
    # the extremes of the difference are exact in fp16
    assert float((np.int32(0) - np.int32(256))) == float(np.float16(-256)) and float(np.float16(254)) == 254.0
