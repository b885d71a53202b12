"""int4 dequant-GEMM through the C ABI against the CPU restatement (oracle/quant.py).

Integer and format work (shuffle, reconstruction) is compared bit for bit.  The GEMMs
accumulate in fp32 and round once to fp16; the oracle multiplies the same fp16 weights in fp32,
so the tolerance below is one fp16 rounding of the result (rtol 1.5e-3) plus an absolute term
for cancellation (1e-4 of the largest result).  The reference's own fused kernels are looser
than this: they add fp16 partial sums with atomics (q_gemm.cu:329-336).  No reference fixture
exists for 4-bit GPTQ or AWQ: parity unpinned.
"""
import numpy as np
import pytest
import torch

from oracle import quant

pytestmark = pytest.mark.gpu

RTOL = 1.5e-3
ATOL_FRAC = 1e-4


def _dev(x, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def _close(got: torch.Tensor, ref: np.ndarray):
    got = got.float().cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=RTOL, atol=ATOL_FRAC * float(np.abs(ref).max()) + 1e-7)


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
    return qweight, qzeros, scales, g_idx.astype(np.int32)


@pytest.mark.parametrize("k,n", [(64, 8), (256, 136), (1024, 512)])
def test_gptq_shuffle_bit_exact(k, n):
    from splitwise_amd import _C
    qweight, _, _, _ = _gptq_case(k, n, 32, k + n)
    dev = _dev(qweight)
    _C.ops.gptq_shuffle(dev, torch.empty((1, 1), device="meta"), 4)
    assert np.array_equal(dev.cpu().numpy(), quant.gptq_shuffle(qweight))


def test_gptq_shuffle_with_row_permutation_bit_exact():
    from splitwise_amd import _C
    qweight, _, _, g_idx = _gptq_case(512, 264, 64, 5, act_order=True)
    perm = np.argsort(g_idx, kind="stable").astype(np.int32)
    dev = _dev(qweight)
    _C.ops.gptq_shuffle(dev, _dev(perm), 4)
    assert np.array_equal(dev.cpu().numpy(), quant.gptq_shuffle(qweight, perm))


@pytest.mark.parametrize("group", [32, 128, 512])
@pytest.mark.parametrize("mode", ["exllama", "exllama_act_order", "plain", "plain_act_order"])
def test_gptq_dequant_bit_exact(mode, group):
    from splitwise_amd import _C
    k, n = 512, 200
    act = mode.endswith("act_order")
    qweight, qzeros, scales, g_idx = _gptq_case(k, n, group, 11, act_order=act)
    if mode.startswith("exllama"):
        perm = np.argsort(g_idx, kind="stable").astype(np.int32) if act else None
        w = quant.gptq_shuffle(qweight, perm)
        idx_dev = _dev(perm) if act else torch.empty((1, 1), device="meta")
        got = _C.ops.dequant_gptq(_dev(w), _dev(qzeros), _dev(scales), idx_dev, 4, True)
        ref = quant.gptq_dequant(w, qzeros, scales, perm, True)
    else:
        got = _C.ops.dequant_gptq(_dev(qweight), _dev(qzeros), _dev(scales), _dev(g_idx), 4, False)
        ref = quant.gptq_dequant(qweight, qzeros, scales, g_idx, False)
    assert np.array_equal(got.cpu().numpy().view(np.uint16), ref.view(np.uint16))


GEMM_SHAPES = [          # (M, K, N, group)
    (1, 128, 128, 128), (1, 2560, 5120, 128), (3, 256, 136, 32), (7, 1024, 520, 64), (8, 5120, 7680, 128),
    (16, 512, 384, 512), (32, 6912, 1024, 128), (33, 512, 256, 128), (50, 1024, 384, 64), (64, 2048, 1000, 128),
    (100, 512, 264, 32), (200, 1024, 512, 128), (5, 96, 64, 32), (40, 160, 72, 32), (130, 384, 200, 64),
    # the 128-row tile kernel: ragged rows and columns, one and two row tiles, groups of 64 / 128 / 256 / all of k,
    # k split across workgroups (few column tiles) and not (many), and the wide-matrix case from 97 rows on
    (129, 512, 136, 64), (256, 1024, 264, 128), (255, 768, 1024, 256), (192, 2560, 320, 64), (250, 1280, 72, 1280),
    (160, 256, 3200, 128), (100, 256, 7680, 128), (200, 256, 16648, 128),
    # 33..128 rows: tiles of 64 columns whose 8 waves are 8 k parts -- 64-row tiles up to 64 rows (and, with a short
    # k and few columns, two of them up to 128 rows), 128-row tiles otherwise; k split across workgroups (few
    # columns, long k) and not
    (40, 4096, 72, 128), (64, 256, 520, 64), (65, 3328, 72, 128), (100, 4096, 1000, 128), (128, 5120, 520, 64),
    (96, 512, 16648, 128), (128, 512, 264, 64), (70, 1024, 136, 128), (127, 3072, 8192, 256),
]


@pytest.mark.parametrize("m,k,n,group", GEMM_SHAPES)
def test_gptq_gemm_shuffled(m, k, n, group):
    from splitwise_amd import _C
    qweight, qzeros, scales, _ = _gptq_case(k, n, group, m + k + n)
    a = np.random.default_rng(m).standard_normal((m, k)).astype(np.float16)
    w = quant.gptq_shuffle(qweight)
    got = _C.ops.gptq_gemm(_dev(a), _dev(w), _dev(qzeros), _dev(scales), torch.empty((1, 1), device="meta"), True, 4)
    assert got.shape == (m, n) and got.dtype == torch.float16
    _close(got, quant.gptq_gemm(a, w, qzeros, scales, None, True))


def test_gptq_gemm_tile_kernel_on_seeded_random_shapes():
    """Shapes drawn at random inside the 128-row tile kernel's domain (rows 129..256, k in multiples of 256,
    columns in multiples of 8, groups of 64 / 128 / 256 / all of k), against the CPU restatement."""
    from splitwise_amd import _C
    rng = np.random.default_rng(2024)
    for case in range(10):
        m = int(rng.integers(129, 257))
        k = 256 * int(rng.integers(1, 9))
        n = 8 * int(rng.integers(1, 200))
        group = int(rng.choice([g for g in (64, 128, 256, k) if k % g == 0]))
        qweight, qzeros, scales, _ = _gptq_case(k, n, group, 1000 + case)
        a = rng.standard_normal((m, k)).astype(np.float16)
        w = quant.gptq_shuffle(qweight)
        got = _C.ops.gptq_gemm(_dev(a), _dev(w), _dev(qzeros), _dev(scales), torch.empty((1, 1), device="meta"), True, 4)
        _close(got, quant.gptq_gemm(a, w, qzeros, scales, None, True))


def test_gptq_gemm_64_column_tiles_on_seeded_random_shapes():
    """Shapes drawn at random inside the domain of the 64-column tile kernels (rows 33..128, k in multiples of 256,
    columns in multiples of 8, groups of 64 / 128 / 256 / all of k), against the CPU restatement; every launch
    twice, bit for bit the same (the k parts and the k splits are summed in a fixed order)."""
    from splitwise_amd import _C
    rng = np.random.default_rng(2025)
    for case in range(12):
        m = int(rng.integers(33, 129))
        k = 256 * int(rng.integers(1, 25))
        n = 8 * int(rng.integers(1, 300))
        group = int(rng.choice([g for g in (64, 128, 256, k) if k % g == 0]))
        qweight, qzeros, scales, _ = _gptq_case(k, n, group, 2000 + case)
        a = rng.standard_normal((m, k)).astype(np.float16)
        args = (_dev(a), _dev(quant.gptq_shuffle(qweight)), _dev(qzeros), _dev(scales),
                torch.empty((1, 1), device="meta"), True, 4)
        got = _C.ops.gptq_gemm(*args)
        _close(got, quant.gptq_gemm(a, quant.gptq_shuffle(qweight), qzeros, scales, None, True))
        assert torch.equal(_C.ops.gptq_gemm(*args), got), (m, k, n, group)


@pytest.mark.parametrize("m,k,n,group", [(1, 512, 256, 128), (9, 1024, 520, 64), (48, 512, 128, 32), (70, 512, 136, 128),
                                         (140, 512, 264, 128)])
def test_gptq_gemm_act_order(m, k, n, group):
    from splitwise_amd import _C
    qweight, qzeros, scales, g_idx = _gptq_case(k, n, group, 3 * m + k, act_order=True)
    perm = np.argsort(g_idx, kind="stable").astype(np.int32)
    a = np.random.default_rng(m).standard_normal((m, k)).astype(np.float16)
    w = quant.gptq_shuffle(qweight, perm)
    got = _C.ops.gptq_gemm(_dev(a), _dev(w), _dev(qzeros), _dev(scales), _dev(perm), True, 4)
    ref = quant.gptq_gemm(a, w, qzeros, scales, perm, True)
    _close(got, ref)
    # and the same numbers from the plain-weight path that keeps g_idx
    got2 = _C.ops.gptq_gemm(_dev(a), _dev(qweight), _dev(qzeros), _dev(scales), _dev(g_idx), False, 4)
    _close(got2, ref)


def test_gptq_gemm_is_reproducible_and_reuses_its_workspace():
    from splitwise_amd import _C
    m, k, n, group = 8, 4096, 1024, 128
    qweight, qzeros, scales, _ = _gptq_case(k, n, group, 77)
    a = _dev(np.random.default_rng(0).standard_normal((m, k)).astype(np.float16))
    args = (_dev(quant.gptq_shuffle(qweight)), _dev(qzeros), _dev(scales), None, True, 4)
    first = _C.ops.gptq_gemm(a, *args)
    for _ in range(3):
        assert torch.equal(_C.ops.gptq_gemm(a, *args), first)


def test_gptq_large_m_goes_through_reconstruction():
    from splitwise_amd import _C
    m, k, n, group = _C.GPTQ_FUSED_MAX_ROWS + 44, 512, 256, 128
    qweight, qzeros, scales, _ = _gptq_case(k, n, group, 78)
    a = np.random.default_rng(1).standard_normal((m, k)).astype(np.float16)
    w = quant.gptq_shuffle(qweight)
    got = _C.ops.gptq_gemm(_dev(a), _dev(w), _dev(qzeros), _dev(scales), None, True, 4)
    _close(got, quant.gptq_gemm(a, w, qzeros, scales, None, True))


def test_gptq_rejects_other_bit_widths():
    from splitwise_amd import _C
    qweight, qzeros, scales, _ = _gptq_case(128, 64, 128, 1)
    with pytest.raises(RuntimeError, match="4-bit"):
        _C.ops.gptq_gemm(torch.zeros((1, 256), dtype=torch.float16, device="cuda"), _dev(qweight), _dev(qzeros),
                         _dev(scales), None, True, 2)


# ------------------------------------------------------------------------------------ AWQ
def _awq_case(k, n, group, seed):
    rng = np.random.default_rng(seed)
    q = rng.integers(0, 16, size=(k, n))
    zeros = rng.integers(0, 16, size=(k // group, n))
    scales = (rng.uniform(0.5, 1.5, size=(k // group, n)) * 1e-2).astype(np.float16)
    qweight, qzeros = quant.awq_pack(q, zeros)
    return qweight, qzeros, scales


@pytest.mark.parametrize("k,n,group", [(128, 64, 32), (512, 320, 64), (1024, 1024, 128)])
def test_awq_dequantize_bit_exact(k, n, group):
    from splitwise_amd import _C
    qweight, qzeros, scales = _awq_case(k, n, group, k + n)
    got = _C.ops.awq_dequantize(_dev(qweight), _dev(scales), _dev(qzeros), 0, 0, 0)
    ref = quant.awq_dequantize(qweight, scales, qzeros)
    assert np.array_equal(got.cpu().numpy().view(np.uint16), ref.view(np.uint16))


@pytest.mark.parametrize("m,k,n,group", [(1, 128, 128, 128), (1, 2560, 5120, 128), (5, 256, 64, 32), (8, 5120, 7680, 128),
                                         (17, 1024, 320, 64), (32, 512, 256, 128), (33, 512, 576, 64),
                                         (100, 1024, 512, 128), (255, 256, 128, 128)])
def test_awq_gemm(m, k, n, group):
    from splitwise_amd import _C
    qweight, qzeros, scales = _awq_case(k, n, group, m + k + n)
    a = np.random.default_rng(m).standard_normal((m, k)).astype(np.float16)
    got = _C.ops.awq_gemm(_dev(a), _dev(qweight), _dev(scales), _dev(qzeros), 8)
    assert got.shape == (m, n) and got.dtype == torch.float16
    _close(got, quant.awq_gemm(a, qweight, scales, qzeros))


@pytest.mark.parametrize("k,n,group", [(128, 64, 32), (512, 320, 64), (1024, 1024, 128)])
def test_awq_repacked_to_kmajor_reconstructs_the_same_bits(k, n, group):
    """The one-time repack into the GPTQ kernels' layout loses nothing: the reconstruction from
    the repacked words equals the CPU restatement of the AWQ reconstruction bit for bit (zeros
    0 and 15 included: they are stored as they are, no GPTQ -1)."""
    from splitwise_amd import _C
    qweight, qzeros, scales = _awq_case(k, n, group, k + n)
    qw_k, qz_k = _C.ops.awq_to_kmajor(_dev(qweight), _dev(qzeros))
    assert qw_k.shape == (k // 8, n) and qz_k.shape == (k // group, n // 8)
    got = _C.ops.int4_kmajor_dequant(qw_k, qz_k, _dev(scales), 0)
    ref = quant.awq_dequantize(qweight, scales, qzeros)
    assert np.array_equal(got.cpu().numpy().view(np.uint16), ref.view(np.uint16))


@pytest.mark.parametrize("m,k,n,group", [(1, 2560, 5120, 128), (5, 256, 64, 32), (33, 512, 576, 64),
                                         (100, 1024, 512, 128), (200, 1024, 384, 64), (300, 512, 256, 128)])
def test_awq_gemm_through_the_kmajor_kernels(m, k, n, group):
    from splitwise_amd import _C
    qweight, qzeros, scales = _awq_case(k, n, group, m + k + n)
    a = np.random.default_rng(m).standard_normal((m, k)).astype(np.float16)
    qw_k, qz_k = _C.ops.awq_to_kmajor(_dev(qweight), _dev(qzeros))
    got = _C.ops.int4_kmajor_gemm(_dev(a), qw_k, qz_k, _dev(scales), 0)
    _close(got, quant.awq_gemm(a, qweight, scales, qzeros))


def test_awq_linear_method_repacks_once_and_matches_the_checkpoint_layout_path():
    from splitwise_amd import _C
    from splitwise_amd.model_executor.layers.quantization.awq import AWQConfig, AWQLinearMethod
    k, n, group = 512, 320, 64
    qweight, qzeros, scales = _awq_case(k, n, group, 7)
    x = _dev(np.random.default_rng(3).standard_normal((2, 9, k)).astype(np.float16))
    outs = []
    for repack in (False, True):
        method = AWQLinearMethod(AWQConfig(4, group, True), repack=repack)
        weights = method.create_weights(k, n, k, n, torch.float16)
        weights = {name: torch.nn.Parameter(_dev(t), requires_grad=False)
                   for name, t in (("qweight", qweight), ("qzeros", qzeros), ("scales", scales))}
        outs.append(method.apply_weights(weights, x))
        again = method.apply_weights(weights, x)
        assert torch.equal(again, outs[-1])
        assert bool(weights.get("_kmajor")) == repack
        assert tuple(weights["qweight"].shape) == ((k // 8, n) if repack else (k, n // 8))
    assert outs[0].shape == outs[1].shape == (2, 9, n)
    torch.testing.assert_close(outs[0].float(), outs[1].float(), atol=2e-2, rtol=1e-2)


def test_awq_argument_checks_mirror_the_reference():
    from splitwise_amd import _C
    qweight, qzeros, scales = _awq_case(128, 72, 32, 0)               # N = 72 is not a multiple of 64
    x = torch.zeros((1, 128), dtype=torch.float16, device="cuda")
    with pytest.raises(ValueError, match="cta_N"):
        _C.ops.awq_gemm(x, _dev(qweight), _dev(scales), _dev(qzeros), 8)


def test_int4_layers_keep_their_fp16_weight_for_many_row_calls():
    """Above the fused kernels' row limit a layer reconstructs its weight and runs a dense GEMM; the
    reconstruction is kept (within a per-process budget), so the second call of a layer does no reconstruction
    and gives the same bits as the first and as the operator that reconstructs on every call."""
    from splitwise_amd import _C
    from splitwise_amd.model_executor.layers.quantization import dense_copy
    from splitwise_amd.model_executor.layers.quantization.gptq import ExllamaState, GPTQConfig, GPTQLinearMethod
    k, n, group, m = 512, 384, 128, 300
    qweight, qzeros, scales, _ = _gptq_case(k, n, group, 21)
    x = _dev(np.random.default_rng(5).standard_normal((m, k)).astype(np.float16))
    method = GPTQLinearMethod(GPTQConfig(4, group, False))
    weights = {name: torch.nn.Parameter(_dev(t), requires_grad=False)
               for name, t in (("qweight", qweight), ("qzeros", qzeros), ("scales", scales))}
    weights["g_idx"] = torch.nn.Parameter(torch.arange(k, dtype=torch.int32, device="cuda") // group, requires_grad=False)
    weights["exllama_state"] = ExllamaState.UNINITIALIZED
    used = dense_copy._used_bytes
    first = method.apply_weights(weights, x)
    assert dense_copy._used_bytes == used + k * n * 2 and weights["_dense_copy"].shape == (k, n)
    calls = []
    plain = _C.ops.dequant_gptq
    _C.ops.dequant_gptq = lambda *a: calls.append(1) or plain(*a)
    try:
        second = method.apply_weights(weights, x)
    finally:
        _C.ops.dequant_gptq = plain
    assert not calls and torch.equal(first, second)
    direct = _C.ops.gptq_gemm(x, weights["qweight"], weights["qzeros"], weights["scales"], weights["g_idx"], True, 4)
    assert torch.equal(first, direct)
    small = method.apply_weights(weights, x[:7])                    # few rows: the fused int4 kernel, as before
    torch.testing.assert_close(small.float(), first[:7].float(), atol=2e-2, rtol=1e-2)
    del weights, first, second
    import gc
    gc.collect()
    assert dense_copy._used_bytes == used                           # the budget comes back with the layer


# ----------------------------------------------------------------------------- GPTQ, 8-bit weights
def _gptq8_case(k, n, group, seed, act_order=False):
    rng = np.random.default_rng(seed)
    q = rng.integers(0, 256, size=(k, n))
    groups = k // group
    zeros = rng.integers(1, 257, size=(groups, n))
    scales = (rng.uniform(0.5, 1.5, size=(groups, n)) * 1e-3).astype(np.float16)
    g_idx = np.arange(k) // group
    if act_order:
        g_idx = rng.permutation(g_idx)
    qweight, qzeros = quant.gptq8_pack(q, zeros)
    return qweight, qzeros, scales, g_idx.astype(np.int32)


@pytest.mark.parametrize("mode", ["exllama", "exllama_act_order", "plain"])
def test_gptq_8bit_shuffle_and_dequant_bit_exact(mode):
    """8-bit weights (qdq_8.cuh, q_gemm.cu:810-918, :1989-2029): the row gather and the reconstruction against the
    restatement, bit for bit; zero points over the whole range 1..256 (the difference q - zero spans -256..254)."""
    from splitwise_amd import _C
    k, n, group = 512, 264, 64
    qweight, qzeros, scales, g_idx = _gptq8_case(k, n, group, 21, act_order=mode != "exllama")
    none = torch.empty((1, 1), device="meta")
    if mode == "plain":
        got = _C.ops.dequant_gptq(_dev(qweight), _dev(qzeros), _dev(scales), _dev(g_idx), 8, False)
        ref = quant.gptq8_dequant(qweight, qzeros, scales, g_idx, False)
    else:
        perm = np.argsort(g_idx, kind="stable").astype(np.int32) if mode == "exllama_act_order" else None
        dev = _dev(qweight)
        _C.ops.gptq_shuffle(dev, _dev(perm) if perm is not None else none, 8)
        w = quant.gptq8_shuffle(qweight, perm)
        assert np.array_equal(dev.cpu().numpy(), w)
        got = _C.ops.dequant_gptq(dev, _dev(qzeros), _dev(scales), _dev(perm) if perm is not None else none, 8, True)
        ref = quant.gptq8_dequant(w, qzeros, scales, perm, True)
    assert np.array_equal(got.cpu().numpy().view(np.uint16), ref.view(np.uint16))


@pytest.mark.parametrize("m,k,n,group", [(1, 512, 256, 128), (9, 1024, 520, 64), (24, 512, 132, 32), (70, 512, 136, 128)])
def test_gptq_8bit_gemm(m, k, n, group):
    """The fused kernel (up to 24 rows) and reconstruction + dense GEMM (above), plain and act-order forms."""
    from splitwise_amd import _C
    qweight, qzeros, scales, g_idx = _gptq8_case(k, n, group, 5 * m + n, act_order=True)
    a = np.random.default_rng(m).standard_normal((m, k)).astype(np.float16)
    none = torch.empty((1, 1), device="meta")
    plain_q, plain_z, plain_s, _ = _gptq8_case(k, n, group, 5 * m + n)
    got = _C.ops.gptq_gemm(_dev(a), _dev(plain_q), _dev(plain_z), _dev(plain_s), none, True, 8)
    _close(got, quant.gptq8_gemm(a, plain_q, plain_z, plain_s, None, True))
    perm = np.argsort(g_idx, kind="stable").astype(np.int32)
    w = quant.gptq8_shuffle(qweight, perm)
    got = _C.ops.gptq_gemm(_dev(a), _dev(w), _dev(qzeros), _dev(scales), _dev(perm), True, 8)
    ref = quant.gptq8_gemm(a, w, qzeros, scales, perm, True)
    _close(got, ref)
    got2 = _C.ops.gptq_gemm(_dev(a), _dev(qweight), _dev(qzeros), _dev(scales), _dev(g_idx), False, 8)
    _close(got2, ref)


def test_gptq_8bit_linear_method():
    """GPTQLinearMethod with weight_bits = 8: parameter shapes of the checkpoint format (pack factor 4), the lazy
    shuffle, the fused path and the many-row path."""
    from splitwise_amd.model_executor.layers.quantization.gptq import GPTQConfig
    k, n, group = 512, 256, 128
    method = GPTQConfig(8, group, False).get_linear_method()
    weights = method.create_weights(k, n, k, n, torch.float16)
    assert weights["qweight"].shape == (k // 4, n) and weights["qzeros"].shape == (k // group, n // 4)
    qweight, qzeros, scales, _ = _gptq8_case(k, n, group, 33)
    weights = {key: (val.cuda() if isinstance(val, torch.Tensor) else val) for key, val in weights.items()}
    weights["qweight"].data.copy_(_dev(qweight))
    weights["qzeros"].data.copy_(_dev(qzeros))
    weights["scales"].data.copy_(_dev(scales))
    for m in (3, 100):
        a = np.random.default_rng(m).standard_normal((m, k)).astype(np.float16)
        _close(method.apply_weights(weights, _dev(a)), quant.gptq8_gemm(a, qweight, qzeros, scales, None, True))
    with pytest.raises(ValueError, match="4-bit and 8-bit"):
        GPTQConfig(3, group, False)
