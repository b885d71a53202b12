"""CPU restatement (numpy) of the reference's int4 weight formats and dequant-GEMM semantics.

TEST INFRASTRUCTURE ONLY: imported by tests/, smoke() and the cpu_baseline leg of bench.py,
never by the product path.

PARITY UNPINNED: the reference holds no test or golden vector for 4-bit or 8-bit dense GPTQ or for AWQ (its
only quantization test, tests/kernels/test_dequant.py:60-112, is 8-bit MoE and needs Triton
and a GPU), and its CUDA kernels cannot run here.  Every function below is therefore a
restatement of the cited CUDA source, checked only for self-consistency (pack -> dequantize
recovers the integers; shuffle followed by the shuffled-format reader equals the plain
reader).

Formats
  GPTQ  qweight int32 [K/8, N]: nibble i of word [r, n] is row 8r+i (q_gemm.cu:1549-1558);
        qzeros int32 [G, N/8]: nibble n%8 holds zero-1 (q_gemm.cu:277, 1557);
        scales fp16 [G, N]; g_idx int32 [K] group of each row (act-order) or k // group_size.
  AWQ   qweight int32 [K, N/8], qzeros int32 [K/g, N/8], scales fp16 [K/g, N]; nibble j of a
        word is logical column AWQ_NIBBLE_TO_COL[j] of the word's 8 columns
        (dequantize.cuh:40-56 emits nibbles 0,4,1,5,2,6,3,7 as columns 0..7); zero stored as is.
  Both  W[k, n] = fp16((q - zero) * scale): exact integer difference in fp16, one rounding for
        the product (q_gemm.cu:1020,1029,1558; gemm_kernels.cu:321-328).
"""
import numpy as np

AWQ_COL_TO_NIBBLE = np.array([0, 4, 1, 5, 2, 6, 3, 7])
AWQ_NIBBLE_TO_COL = np.argsort(AWQ_COL_TO_NIBBLE)           # [0, 2, 4, 6, 1, 3, 5, 7]


def _pack_rows(q: np.ndarray) -> np.ndarray:
    """[K, N] values 0..15 -> uint32 [K/8, N], nibble i = row 8r+i."""
    k, n = q.shape
    q = q.astype(np.uint32).reshape(k // 8, 8, n)
    out = np.zeros((k // 8, n), dtype=np.uint32)
    for i in range(8):
        out |= q[:, i, :] << np.uint32(4 * i)
    return out


def _unpack_rows(w: np.ndarray) -> np.ndarray:
    r, n = w.shape
    w = w.astype(np.uint32)
    out = np.empty((r, 8, n), dtype=np.int32)
    for i in range(8):
        out[:, i, :] = (w >> np.uint32(4 * i)) & np.uint32(0xF)
    return out.reshape(r * 8, n)


def _pack_cols(q: np.ndarray, order: np.ndarray) -> np.ndarray:
    """[R, N] values 0..15 -> uint32 [R, N/8]; logical column c of a word sits in nibble order[c]."""
    r, n = q.shape
    q = q.astype(np.uint32).reshape(r, n // 8, 8)
    out = np.zeros((r, n // 8), dtype=np.uint32)
    for c in range(8):
        out |= q[:, :, c] << np.uint32(4 * int(order[c]))
    return out


def _unpack_cols(w: np.ndarray, order: np.ndarray) -> np.ndarray:
    r, nw = w.shape
    w = w.astype(np.uint32)
    out = np.empty((r, nw, 8), dtype=np.int32)
    for c in range(8):
        out[:, :, c] = (w >> np.uint32(4 * int(order[c]))) & np.uint32(0xF)
    return out.reshape(r, nw * 8)


_NATURAL = np.arange(8)


# ------------------------------------------------------------------------------- GPTQ
def gptq_pack(q: np.ndarray, zeros: np.ndarray):
    """q [K, N] in 0..15, zeros [G, N] in 1..16 (the effective zero point) -> (qweight, qzeros)
    as int32 arrays in checkpoint layout."""
    assert zeros.min() >= 1 and zeros.max() <= 16
    return _pack_rows(q).view(np.int32), _pack_cols(zeros - 1, _NATURAL).view(np.int32)


def gptq_shuffle(qweight: np.ndarray, q_perm=None) -> np.ndarray:
    """shuffle_exllama_weight (q_gemm.cu:2032-2093): optional row gather that makes groups
    sequential (make_sequential_4bit_kernel :1812-1853: new row i = old row q_perm[i]), then
    the in-word nibble order q0 q2 q4 q6 | q1 q3 q5 q7 (shuffle_4bit_8, qdq_4.cuh:16-35)."""
    q = _unpack_rows(qweight.view(np.uint32))
    if q_perm is not None:
        q = q[np.asarray(q_perm, dtype=np.int64)]
    k, n = q.shape
    q = q.astype(np.uint32).reshape(k // 8, 8, n)
    out = np.zeros((k // 8, n), dtype=np.uint32)
    for i in range(4):
        out |= q[:, 2 * i, :] << np.uint32(4 * i)
        out |= q[:, 2 * i + 1, :] << np.uint32(4 * i + 16)
    return out.view(np.int32)


def _unshuffle_rows(qweight_shuffled: np.ndarray) -> np.ndarray:
    """Reads a shuffled word the way dequant_4bit_8_gptq does (qdq_4.cuh:110-143): masks
    0x000f000f / 0x00f000f0 pair rows (0,1) and (2,3), then (4,5) and (6,7) after >> 8."""
    w = qweight_shuffled.view(np.uint32).astype(np.uint32)
    r, n = w.shape
    out = np.empty((r, 8, n), dtype=np.int32)
    for half in range(2):
        ww = w >> np.uint32(8 * half)
        out[:, 4 * half + 0, :] = ww & np.uint32(0xF)
        out[:, 4 * half + 1, :] = (ww >> np.uint32(16)) & np.uint32(0xF)
        out[:, 4 * half + 2, :] = (ww >> np.uint32(4)) & np.uint32(0xF)
        out[:, 4 * half + 3, :] = (ww >> np.uint32(20)) & np.uint32(0xF)
    return out.reshape(r * 8, n)


def gptq_dequant(qweight: np.ndarray, qzeros: np.ndarray, scales: np.ndarray, g_idx, use_exllama: bool,
                 zero_offset: int = 1) -> np.ndarray:
    """fp16 [K, N] in ORIGINAL row order.  ``zero_offset``: what is added to the stored zero, 1 for
    GPTQ checkpoints (q_gemm.cu:929); 0 describes AWQ weights repacked into this layout.
    use_exllama: shuffled weights; ``g_idx`` is the row permutation or None; row k' of the
    shuffled matrix belongs to group k' // group_size and lands in output row perm[k']
    (reconstruct_exllama_4bit_kernel, q_gemm.cu:920-1037).
    otherwise: plain weights, ``g_idx[k]`` is the group of row k (reconstruct_gptq_kernel
    :1517-1561)."""
    groups, n = scales.shape
    zeros = _unpack_cols(qzeros.view(np.uint32), _NATURAL) + zero_offset        # [G, N]
    if use_exllama:
        q = _unshuffle_rows(qweight)
        k = q.shape[0]
        grp = np.arange(k) // (k // groups)
    else:
        q = _unpack_rows(qweight.view(np.uint32))
        k = q.shape[0]
        grp = np.asarray(g_idx, dtype=np.int64) if g_idx is not None else np.arange(k) // (k // groups)
    diff = (q - zeros[grp]).astype(np.float16)                                 # exact
    w = diff * scales.astype(np.float16)[grp]                                  # one fp16 rounding
    if use_exllama and g_idx is not None:
        out = np.empty_like(w)
        out[np.asarray(g_idx, dtype=np.int64)] = w
        return out
    return w


def gptq_gemm(a: np.ndarray, qweight, qzeros, scales, g_idx, use_exllama: bool) -> np.ndarray:
    """fp32 result of a . W with W the fp16 reconstruction (what gemm_half_q_half_cuda computes
    above 50 rows, q_gemm.cu:1712-1726, up to summation order)."""
    w = gptq_dequant(qweight, qzeros, scales, g_idx, use_exllama).astype(np.float32)
    return a.astype(np.float32) @ w


# ------------------------------------------------------------------------------- GPTQ, 8-bit
def gptq8_pack(q: np.ndarray, zeros: np.ndarray):
    """q [K, N] in 0..255, zeros [G, N] in 1..256 (the effective zero point) -> (qweight [K/4, N] with byte i of a
    word = row 4r+i, qzeros [G, N/4] with byte n % 4 = zero - 1) as int32 arrays in checkpoint layout."""
    assert zeros.min() >= 1 and zeros.max() <= 256 and q.min() >= 0 and q.max() <= 255
    k, n = q.shape
    g = zeros.shape[0]
    qw = np.zeros((k // 4, n), dtype=np.uint32)
    qq = q.astype(np.uint32).reshape(k // 4, 4, n)
    for i in range(4):
        qw |= qq[:, i, :] << np.uint32(8 * i)
    qz = np.zeros((g, n // 4), dtype=np.uint32)
    zz = (zeros - 1).astype(np.uint32).reshape(g, n // 4, 4)
    for c in range(4):
        qz |= zz[:, :, c] << np.uint32(8 * c)
    return qw.view(np.int32), qz.view(np.int32)


def _unpack8_rows(w: np.ndarray) -> np.ndarray:
    r, n = w.shape
    w = w.view(np.uint32)
    out = np.empty((r, 4, n), dtype=np.int32)
    for i in range(4):
        out[:, i, :] = (w >> np.uint32(8 * i)) & np.uint32(0xFF)
    return out.reshape(r * 4, n)


def gptq8_shuffle(qweight: np.ndarray, q_perm=None) -> np.ndarray:
    """shuffle_exllama_weight for 8-bit weights: the in-word order does not change (shuffle_8bit_4 is empty,
    qdq_8.cuh:14-21); with a permutation new row i = old row q_perm[i] (make_sequential_8bit_kernel,
    q_gemm.cu:1989-2029)."""
    if q_perm is None:
        return qweight.copy()
    q = _unpack8_rows(qweight)[np.asarray(q_perm, dtype=np.int64)]
    k, n = q.shape
    out = np.zeros((k // 4, n), dtype=np.uint32)
    qq = q.astype(np.uint32).reshape(k // 4, 4, n)
    for i in range(4):
        out |= qq[:, i, :] << np.uint32(8 * i)
    return out.view(np.int32)


def gptq8_dequant(qweight: np.ndarray, qzeros: np.ndarray, scales: np.ndarray, g_idx, use_exllama: bool) -> np.ndarray:
    """fp16 [K, N] in ORIGINAL row order: dq_ns(q, zero + 1) * scale (qdq_8.cuh:23-38, qdq_util.cuh:42-46;
    reconstruct_exllama_8bit_kernel q_gemm.cu:810-918, reconstruct_gptq_kernel :1517-1561).  ``g_idx`` as in
    gptq_dequant."""
    groups, n = scales.shape
    z = qzeros.view(np.uint32)
    zeros = np.empty((groups, n // 4, 4), dtype=np.int32)
    for c in range(4):
        zeros[:, :, c] = (z >> np.uint32(8 * c)) & np.uint32(0xFF)
    zeros = zeros.reshape(groups, n) + 1
    q = _unpack8_rows(qweight)
    k = q.shape[0]
    if use_exllama or g_idx is None:
        grp = np.arange(k) // (k // groups)
    else:
        grp = np.asarray(g_idx, dtype=np.int64)
    diff = (q - zeros[grp]).astype(np.float16)                                 # |difference| <= 256: exact
    w = diff * scales.astype(np.float16)[grp]                                  # one fp16 rounding
    if use_exllama and g_idx is not None:
        out = np.empty_like(w)
        out[np.asarray(g_idx, dtype=np.int64)] = w
        return out
    return w


def gptq8_gemm(a: np.ndarray, qweight, qzeros, scales, g_idx, use_exllama: bool) -> np.ndarray:
    return a.astype(np.float32) @ gptq8_dequant(qweight, qzeros, scales, g_idx, use_exllama).astype(np.float32)


# ------------------------------------------------------------------------------- AWQ
def awq_pack(q: np.ndarray, zeros: np.ndarray):
    """q [K, N] and zeros [K/g, N] in 0..15 -> (qweight [K, N/8], qzeros [K/g, N/8]) int32."""
    return _pack_cols(q, AWQ_COL_TO_NIBBLE).view(np.int32), _pack_cols(zeros, AWQ_COL_TO_NIBBLE).view(np.int32)


def awq_dequantize(qweight: np.ndarray, scales: np.ndarray, qzeros: np.ndarray) -> np.ndarray:
    """dequantize_weights (gemm_kernels.cu:274-335): fp16 [K, N]."""
    q = _unpack_cols(qweight.view(np.uint32), AWQ_COL_TO_NIBBLE)
    z = _unpack_cols(qzeros.view(np.uint32), AWQ_COL_TO_NIBBLE)
    k = q.shape[0]
    grp = np.arange(k) // (k // scales.shape[0])
    return (q - z[grp]).astype(np.float16) * scales.astype(np.float16)[grp]


def awq_gemm(x: np.ndarray, qweight, scales, qzeros) -> np.ndarray:
    return x.astype(np.float32) @ awq_dequantize(qweight, scales, qzeros).astype(np.float32)
