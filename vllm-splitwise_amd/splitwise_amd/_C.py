"""ctypes binding of libsplitwise_hip.so with the reference's operator table.

The reference exposes its native operators as ``vllm._C.{ops,cache_ops,cuda_utils}``
(``csrc/pybind.cpp:6-129``).  This module offers the same three namespaces with the
same function names and argument order, so callers written against the reference
(`PagedAttentionImpl`, `CacheEngine`, the kernel tests) run unchanged; underneath,
every call goes through the C ABI declared in ``include/splitwise_hip.h`` with raw
device pointers, explicit strides and torch's current HIP stream.

There is no fallback: if the shared library is missing or an entry point fails,
the call raises ``RuntimeError`` (the reference raises the same type through
``TORCH_CHECK``).
"""
from __future__ import annotations

import ctypes
import os
from types import SimpleNamespace
from typing import Dict, List, Optional

import torch

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsplitwise_hip.so")

SW_F32, SW_F16, SW_BF16, SW_FP8_E5M2 = 0, 1, 2, 3
_TORCH_TO_SW = {torch.float32: SW_F32, torch.float16: SW_F16, torch.bfloat16: SW_BF16}

_c_void_p = ctypes.c_void_p
_c_int = ctypes.c_int
_c_i64 = ctypes.c_int64
_c_float = ctypes.c_float

# name -> argtypes; mirrors include/splitwise_hip.h one to one
_SIGNATURES = {
    "sw_abi_version": [],
    "sw_get_max_shared_memory_per_block": [_c_int, ctypes.POINTER(_c_int)],
    "sw_reshape_and_cache": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int,
                             _c_int, _c_int, _c_int, _c_i64, _c_i64, _c_int, _c_int, _c_void_p],
    "sw_copy_blocks": [_c_void_p, _c_void_p, _c_int, _c_void_p, _c_int, _c_i64, _c_void_p],
    "sw_swap_blocks": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_i64, _c_int, _c_void_p],
    "sw_convert_fp8_e5m2": [_c_void_p, _c_void_p, _c_i64, _c_int, _c_int, _c_void_p],
    "sw_paged_attention_v1": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int,
                              _c_float, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p,
                              _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_void_p],
    "sw_paged_attention_v2": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p,
                              _c_int, _c_int, _c_int, _c_int, _c_float, _c_void_p, _c_void_p, _c_int,
                              _c_int, _c_int, _c_void_p, _c_i64, _c_i64, _c_i64, _c_int, _c_int,
                              _c_void_p],
    "sw_paged_attention_v2_static_max_bytes": [],
    "sw_paged_attention_v2_static": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int,
                                     _c_float, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p,
                                     _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_void_p],
    "sw_prefill_attention": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int,
                             _c_int, _c_int, _c_int, _c_float, _c_void_p, _c_i64, _c_i64, _c_i64, _c_i64,
                             _c_int, _c_int, _c_void_p],
    "sw_rms_norm": [_c_void_p, _c_void_p, _c_void_p, _c_float, _c_i64, _c_int, _c_i64, _c_int, _c_void_p],
    "sw_fused_add_rms_norm": [_c_void_p, _c_void_p, _c_void_p, _c_float, _c_i64, _c_int, _c_i64, _c_int,
                              _c_void_p],
    "sw_rotary_embedding": [_c_void_p, _c_void_p, _c_void_p, _c_i64, _c_int, _c_int, _c_int, _c_void_p, _c_int,
                            _c_int, _c_i64, _c_i64, _c_int, _c_void_p],
    "sw_rotary_embedding_and_cache": [_c_void_p] * 7 + [_c_i64, _c_int, _c_int, _c_int, _c_void_p, _c_int, _c_int, _c_int,
                                      _c_int, _c_i64, _c_i64, _c_i64, _c_int, _c_int, _c_void_p],
    "sw_silu_and_mul": [_c_void_p, _c_void_p, _c_i64, _c_int, _c_i64, _c_int, _c_void_p],
    "sw_ipc_get_mem_handle": [_c_void_p, _c_void_p],
    "sw_mem_address_range": [_c_void_p, ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_i64)],
    "sw_ipc_open_mem_handle": [_c_void_p, ctypes.POINTER(_c_void_p)],
    "sw_ipc_close_mem_handle": [_c_void_p],
    "sw_kv_push": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_void_p],
    "sw_kv_push_mapped": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_void_p],
    "sw_kv_signal": [_c_void_p, _c_int, ctypes.c_int32, _c_void_p],
    "sw_kv_wait": [_c_void_p, _c_int, ctypes.c_int32, _c_i64, _c_void_p, _c_void_p],
    "sw_context_attention": [_c_void_p] * 7 + [_c_int] + [_c_void_p] * 3 + [_c_int] * 6 + [ctypes.c_float, _c_void_p]
                            + [_c_i64] * 6 + [_c_int, _c_int, _c_void_p],
    "sw_context_attention_split": [_c_void_p] * 7 + [_c_int] + [_c_void_p] * 3 + [_c_int] * 6 + [ctypes.c_float, _c_void_p]
                                  + [_c_i64] * 6 + [_c_int, _c_int, _c_i64, _c_int, _c_void_p, _c_i64, _c_void_p],
    "sw_context_attention_split_parts": [_c_int] * 6,
    "sw_context_attention_split_workspace_bytes": [_c_int, _c_i64, _c_int, _c_int],
    "sw_custom_ar_signal_bytes": [],
    "sw_custom_ar_init": [ctypes.POINTER(_c_void_p), _c_int, _c_int, _c_i64, _c_i64, _c_i64, _c_i64,
                          ctypes.POINTER(_c_void_p)],
    "sw_custom_all_reduce": [_c_void_p, _c_void_p, _c_void_p, _c_i64, _c_int, _c_int, _c_void_p],
    "sw_custom_ar_dispose": [_c_void_p],
    "sw_quant_gemm_workspace_bytes": [_c_int, _c_int, _c_int],
    "sw_gptq_shuffle": [_c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p, _c_void_p],
    "sw_gptq_gemm": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int,
                     _c_int, _c_int, _c_int, _c_void_p, _c_i64, _c_void_p],
    "sw_awq_to_kmajor": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p],
    "sw_int4_kmajor_gemm": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_int,
                            _c_void_p, _c_i64, _c_void_p],
    "sw_int4_kmajor_dequant": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int, _c_void_p],
    "sw_gptq_dequant": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int,
                        _c_int, _c_void_p],
    "sw_awq_gemm": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_int,
                    _c_void_p, _c_i64, _c_void_p],
    "sw_awq_dequantize": [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p],
}
_RESTYPES = {"sw_quant_gemm_workspace_bytes": _c_i64, "sw_custom_ar_signal_bytes": _c_i64,
             "sw_paged_attention_v2_static_max_bytes": _c_i64,
             "sw_context_attention_split_workspace_bytes": _c_i64}     # everything else returns an int status

_lib = None


def library_path() -> str:
    return _LIB_PATH


ABI_VERSION = 7                # SW_ABI_VERSION of include/splitwise_hip.h this binding was written against


def load_library() -> ctypes.CDLL:
    """Loads the native library once; raises RuntimeError when it is absent or was built from
    another version of the interface."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            f"native library not built: {_LIB_PATH} is missing "
            "(run `make -C vllm-splitwise_amd` or `python -c 'import __graft_entry__ as g; g.build()'`)")
    lib = ctypes.CDLL(_LIB_PATH)
    lib.sw_last_error.restype = ctypes.c_char_p
    lib.sw_last_error.argtypes = []
    lib.sw_abi_version.argtypes, lib.sw_abi_version.restype = [], _c_int
    if lib.sw_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{_LIB_PATH} implements interface version {lib.sw_abi_version()}, this binding needs "
                           f"{ABI_VERSION}: rebuild it (`make -C vllm-splitwise_amd`)")
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, _c_int)
    _lib = lib
    return lib


def exported_symbols() -> List[str]:
    return ["sw_last_error"] + list(_SIGNATURES)


def _check(status: int) -> None:
    if status != 0:
        msg = load_library().sw_last_error().decode("utf-8", "replace")
        raise RuntimeError(msg or f"splitwise_hip call failed with status {status}")


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _act_dtype(t: torch.Tensor) -> int:
    try:
        return _TORCH_TO_SW[t.dtype]
    except KeyError:
        raise RuntimeError(f"Unsupported data type: {t.dtype}") from None


def _cache_dtype(kv_cache_dtype: str, act: int) -> int:
    if kv_cache_dtype == "auto":
        return act
    if kv_cache_dtype == "fp8_e5m2":
        return SW_FP8_E5M2
    raise RuntimeError(f"Unsupported data type of kv cache: {kv_cache_dtype}")


def _require_cuda(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError("splitwise_hip operators need tensors on a HIP device")


# --------------------------------------------------------------------- cache_ops
def reshape_and_cache(key: torch.Tensor, value: torch.Tensor, key_cache: torch.Tensor,
                      value_cache: torch.Tensor, slot_mapping: torch.Tensor,
                      kv_cache_dtype: str) -> None:
    """``cache_ops.reshape_and_cache`` (csrc/cache_kernels.cu:228-268)."""
    _require_cuda(key, value, key_cache, value_cache, slot_mapping)
    act = _act_dtype(key)
    cache = _cache_dtype(kv_cache_dtype, act)
    num_tokens, num_heads, head_size = key.shape
    block_size, x = key_cache.shape[3], key_cache.shape[4]
    if key.stride(2) != 1 or key.stride(1) != head_size or value.stride(2) != 1 or value.stride(1) != head_size:
        raise RuntimeError("reshape_and_cache: key/value heads must be contiguous")
    if slot_mapping.dtype != torch.int64:
        raise RuntimeError("reshape_and_cache: slot_mapping must be int64")
    with torch.cuda.device(key.device):
        _check(load_library().sw_reshape_and_cache(
            key.data_ptr(), value.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
            slot_mapping.data_ptr(), num_tokens, num_heads, head_size, block_size, x,
            key.stride(0), value.stride(0), act, cache, _stream(key)))


_ptr_table_cache: Dict[tuple, torch.Tensor] = {}


def _ptr_table(tensors: List[torch.Tensor]) -> torch.Tensor:
    """Device array of the tensors' base pointers; uploaded once per distinct list."""
    key = (tensors[0].device.index,) + tuple(t.data_ptr() for t in tensors)
    tab = _ptr_table_cache.get(key)
    if tab is None:
        if len(_ptr_table_cache) > 64:
            _ptr_table_cache.clear()
        tab = torch.tensor([t.data_ptr() for t in tensors], dtype=torch.int64, device=tensors[0].device)
        _ptr_table_cache[key] = tab
    return tab


def copy_blocks(key_caches: List[torch.Tensor], value_caches: List[torch.Tensor],
                block_mapping: Dict[int, List[int]]) -> None:
    """``cache_ops.copy_blocks`` (csrc/cache_kernels.cu:95-150)."""
    num_layers = len(key_caches)
    if num_layers != len(value_caches):
        raise RuntimeError("copy_blocks: key_caches and value_caches differ in length")
    if num_layers == 0:
        return
    _require_cuda(key_caches[0], value_caches[0])
    pairs = [(src, dst) for src, dsts in block_mapping.items() for dst in dsts]
    if not pairs:
        return
    dev = key_caches[0].device
    block_bytes = key_caches[0][0].numel() * key_caches[0].element_size()
    mapping = torch.tensor(pairs, dtype=torch.int64).to(dev, non_blocking=False)
    with torch.cuda.device(dev):
        _check(load_library().sw_copy_blocks(
            _ptr_table(key_caches).data_ptr(), _ptr_table(value_caches).data_ptr(), num_layers,
            mapping.data_ptr(), len(pairs), block_bytes, _stream(key_caches[0])))
        # `mapping` is consumed asynchronously: keep it alive until the stream passes it
        mapping.record_stream(torch.cuda.current_stream(dev))


def swap_blocks(src: torch.Tensor, dst: torch.Tensor, block_mapping: Dict[int, int]) -> None:
    """``cache_ops.swap_blocks`` (csrc/cache_kernels.cu:21-60)."""
    if src.is_cuda and dst.is_cuda:
        if src.device != dst.device:
            raise RuntimeError("src and dst must be on the same GPU")
        kind, dev = 0, src.device
    elif src.is_cuda and not dst.is_cuda:
        kind, dev = 1, src.device
    elif not src.is_cuda and dst.is_cuda:
        kind, dev = 2, dst.device
    else:
        raise RuntimeError("Invalid device combination")
    if not block_mapping:
        return
    items = sorted(block_mapping.items())
    n = len(items)
    src_blocks = (ctypes.c_int64 * n)(*[s for s, _ in items])
    dst_blocks = (ctypes.c_int64 * n)(*[d for _, d in items])
    block_bytes = src[0].numel() * src.element_size()
    with torch.cuda.device(dev):
        _check(load_library().sw_swap_blocks(
            src.data_ptr(), dst.data_ptr(), ctypes.addressof(src_blocks), ctypes.addressof(dst_blocks),
            n, block_bytes, kind, torch.cuda.current_stream(dev).cuda_stream))


def convert_fp8_e5m2(src_cache: torch.Tensor, dst_cache: torch.Tensor) -> None:
    """``cache_ops.convert_fp8_e5m2`` (csrc/cache_kernels.cu:296-320)."""
    _require_cuda(src_cache, dst_cache)
    if not (src_cache.is_contiguous() and dst_cache.is_contiguous()):
        raise RuntimeError("convert_fp8_e5m2: tensors must be contiguous")
    if src_cache.numel() != dst_cache.numel():
        raise RuntimeError("convert_fp8_e5m2: size mismatch")
    sdt = SW_FP8_E5M2 if src_cache.dtype == torch.uint8 else _act_dtype(src_cache)
    ddt = SW_FP8_E5M2 if dst_cache.dtype == torch.uint8 else _act_dtype(dst_cache)
    with torch.cuda.device(src_cache.device):
        _check(load_library().sw_convert_fp8_e5m2(
            src_cache.data_ptr(), dst_cache.data_ptr(), src_cache.numel(), sdt, ddt, _stream(src_cache)))


# --------------------------------------------------------------------------- ops
def _pa_common(query, key_cache, value_cache, block_tables, context_lens, alibi_slopes):
    _require_cuda(query, key_cache, value_cache, block_tables, context_lens)
    if block_tables.dtype != torch.int32 or context_lens.dtype != torch.int32:
        raise RuntimeError("paged_attention: block_tables and context_lens must be int32")
    if not block_tables.is_contiguous() and block_tables.stride(1) != 1:
        raise RuntimeError("paged_attention: block_tables rows must be contiguous")
    if alibi_slopes is not None:
        if alibi_slopes.dtype != torch.float32:
            raise RuntimeError("paged_attention: alibi_slopes must be float32")
        if not alibi_slopes.is_cuda:
            alibi_slopes = alibi_slopes.to(query.device)
    return alibi_slopes


def paged_attention_v1(out: torch.Tensor, query: torch.Tensor, key_cache: torch.Tensor,
                       value_cache: torch.Tensor, num_kv_heads: int, scale: float,
                       block_tables: torch.Tensor, context_lens: torch.Tensor, block_size: int,
                       max_context_len: int, alibi_slopes: Optional[torch.Tensor],
                       kv_cache_dtype: str) -> None:
    """``ops.paged_attention_v1`` (csrc/attention/attention_kernels.cu:711-747)."""
    alibi_slopes = _pa_common(query, key_cache, value_cache, block_tables, context_lens, alibi_slopes)
    act = _act_dtype(query)
    cache = _cache_dtype(kv_cache_dtype, act)
    num_seqs, num_heads, head_size = query.shape
    with torch.cuda.device(query.device):
        _check(load_library().sw_paged_attention_v1(
            out.data_ptr(), query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
            num_seqs, num_heads, num_kv_heads, head_size, float(scale), block_tables.data_ptr(),
            context_lens.data_ptr(), block_size, int(max_context_len), block_tables.stride(0),
            alibi_slopes.data_ptr() if alibi_slopes is not None else None, query.stride(0),
            key_cache.stride(0), key_cache.stride(1), act, cache, _stream(query)))


def paged_attention_v2(out: torch.Tensor, exp_sums: torch.Tensor, max_logits: torch.Tensor,
                       tmp_out: torch.Tensor, query: torch.Tensor, key_cache: torch.Tensor,
                       value_cache: torch.Tensor, num_kv_heads: int, scale: float,
                       block_tables: torch.Tensor, context_lens: torch.Tensor, block_size: int,
                       max_context_len: int, alibi_slopes: Optional[torch.Tensor],
                       kv_cache_dtype: str) -> None:
    """``ops.paged_attention_v2`` (csrc/attention/attention_kernels.cu:901-940)."""
    alibi_slopes = _pa_common(query, key_cache, value_cache, block_tables, context_lens, alibi_slopes)
    act = _act_dtype(query)
    cache = _cache_dtype(kv_cache_dtype, act)
    num_seqs, num_heads, head_size = query.shape
    with torch.cuda.device(query.device):
        _check(load_library().sw_paged_attention_v2(
            out.data_ptr(), exp_sums.data_ptr(), max_logits.data_ptr(), tmp_out.data_ptr(),
            query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(), num_seqs, num_heads,
            num_kv_heads, head_size, float(scale), block_tables.data_ptr(), context_lens.data_ptr(),
            block_size, int(max_context_len), block_tables.stride(0),
            alibi_slopes.data_ptr() if alibi_slopes is not None else None, query.stride(0),
            key_cache.stride(0), key_cache.stride(1), act, cache, _stream(query)))


def paged_attention_v2_static(out: torch.Tensor, query: torch.Tensor, key_cache: torch.Tensor,
                              value_cache: torch.Tensor, num_kv_heads: int, scale: float,
                              block_tables: torch.Tensor, context_lens: torch.Tensor, block_size: int,
                              max_context_len: int, alibi_slopes: Optional[torch.Tensor],
                              kv_cache_dtype: str) -> None:
    """``paged_attention_v2`` without the three temporaries: they live in the library's static device
    scratch (32 MiB per GPU), so the call allocates nothing.  The argument list is
    ``paged_attention_v1``'s.  Raises RuntimeError when the temporaries would not fit; calls on one
    GPU must not overlap."""
    alibi_slopes = _pa_common(query, key_cache, value_cache, block_tables, context_lens, alibi_slopes)
    act = _act_dtype(query)
    cache = _cache_dtype(kv_cache_dtype, act)
    num_seqs, num_heads, head_size = query.shape
    with torch.cuda.device(query.device):
        _check(load_library().sw_paged_attention_v2_static(
            out.data_ptr(), query.data_ptr(), key_cache.data_ptr(), value_cache.data_ptr(),
            num_seqs, num_heads, num_kv_heads, head_size, float(scale), block_tables.data_ptr(),
            context_lens.data_ptr(), block_size, int(max_context_len), block_tables.stride(0),
            alibi_slopes.data_ptr() if alibi_slopes is not None else None, query.stride(0),
            key_cache.stride(0), key_cache.stride(1), act, cache, _stream(query)))


def prefill_attention(out: torch.Tensor, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                      seq_starts: torch.Tensor, seq_lens: torch.Tensor, max_seq_len: int, scale: float,
                      alibi_slopes: Optional[torch.Tensor] = None, sliding_window: Optional[int] = None) -> None:
    """Causal prompt attention over (start, length) sequences; stands where the reference calls
    xformers / flash-attn (backends/xformers.py:155-164).  query [T, H, D], key/value [T, KVH, D]."""
    _require_cuda(out, query, key, value, seq_starts, seq_lens)
    if seq_starts.dtype != torch.int32 or seq_lens.dtype != torch.int32:
        raise RuntimeError("prefill_attention: seq_starts and seq_lens must be int32")
    for t in (out, query, key, value):
        if t.stride(2) != 1 or t.stride(1) != t.shape[2]:
            raise RuntimeError("prefill_attention: heads must be contiguous")
    act = _act_dtype(query)
    if alibi_slopes is not None and not alibi_slopes.is_cuda:
        alibi_slopes = alibi_slopes.to(query.device)
    with torch.cuda.device(query.device):
        _check(load_library().sw_prefill_attention(
            out.data_ptr(), query.data_ptr(), key.data_ptr(), value.data_ptr(), seq_starts.data_ptr(),
            seq_lens.data_ptr(), seq_lens.numel(), int(max_seq_len), query.shape[1], key.shape[1],
            query.shape[2], float(scale), alibi_slopes.data_ptr() if alibi_slopes is not None else None,
            query.stride(0), key.stride(0), value.stride(0), out.stride(0),
            int(sliding_window) if sliding_window else 0, act, _stream(query)))


def context_attention_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, o: torch.Tensor, k_cache: torch.Tensor,
                          v_cache: torch.Tensor, b_loc: torch.Tensor, b_start_loc: torch.Tensor,
                          b_seq_len: torch.Tensor, b_ctx_len: torch.Tensor, max_input_len: int,
                          alibi_slopes: Optional[torch.Tensor] = None, sliding_window: Optional[int] = None,
                          num_parts: Optional[int] = None) -> None:
    """Prompt attention over a cached prefix plus new tokens, with the argument list of the
    reference's Triton entry point (ops/prefix_prefill.py:621-632): q / o [T, H, D], k / v [T, KVH, D]
    (new tokens only), k_cache [blocks, KVH, D/8, bs, 8], v_cache [blocks, KVH, D, bs], b_loc the
    block table [B, max_blocks], b_start_loc the first new-token row of each sequence, b_seq_len
    context + new tokens, b_ctx_len the cached tokens.  scale = D^-1/2 (:641).
    ``num_parts``: workgroups that share the keys of one (query tile, sequence, head); None lets the library
    decide from the launch's shape (few new tokens over a long context leave most of the chip idle otherwise)."""
    _require_cuda(o, q, k, v, k_cache, v_cache, b_loc, b_start_loc, b_seq_len, b_ctx_len)
    for t in (o, q, k, v):
        if t.stride(2) != 1 or t.stride(1) != t.shape[2]:
            raise RuntimeError("context_attention_fwd: heads must be contiguous")
    if k_cache.dtype != q.dtype or v_cache.dtype != q.dtype:
        raise RuntimeError("context_attention_fwd: the cache must hold the activation dtype")
    if k_cache.dim() != 5 or k_cache.shape[4] != 8 or not k_cache.is_contiguous() or not v_cache.is_contiguous():
        raise RuntimeError("context_attention_fwd: key cache [blocks, KVH, D/8, bs, 8] and value cache "
                           "[blocks, KVH, D, bs], both contiguous")
    head = q.shape[2]
    block_size = v_cache.shape[3]
    starts = b_start_loc.to(torch.int32)
    ctx = b_ctx_len.to(torch.int32)
    new = (b_seq_len.to(torch.int32) - ctx).contiguous()
    table = b_loc.to(torch.int32).contiguous()
    act = _act_dtype(q)
    if alibi_slopes is not None:
        alibi_slopes = alibi_slopes.to(device=q.device, dtype=torch.float32)
    lib = load_library()
    if num_parts is None:
        num_parts = lib.sw_context_attention_split_parts(new.numel(), int(max_input_len), q.shape[1], head, block_size,
                                                         table.shape[1] * block_size)
    with torch.cuda.device(q.device):
        args = (o.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
                table.data_ptr(), table.shape[1], starts.data_ptr(), new.data_ptr(), ctx.data_ptr(), new.numel(),
                int(max_input_len), q.shape[1], k.shape[1], head, block_size, float(head) ** -0.5,
                alibi_slopes.data_ptr() if alibi_slopes is not None else None, q.stride(0), k.stride(0), v.stride(0),
                o.stride(0), v_cache.stride(0), v_cache.stride(1), int(sliding_window) if sliding_window else 0, act)
        if num_parts > 1:
            nbytes = int(lib.sw_context_attention_split_workspace_bytes(num_parts, q.shape[0], q.shape[1], head))
            workspace = torch.empty(nbytes // 4, dtype=torch.float32, device=q.device)
            _check(lib.sw_context_attention_split(*args, q.shape[0], int(num_parts), workspace.data_ptr(), nbytes,
                                                  _stream(q)))
        else:
            _check(lib.sw_context_attention(*args, _stream(q)))


def _rows(t: torch.Tensor):
    """(number of rows, row stride in elements) of a tensor whose last dim is contiguous and whose
    leading dims collapse to one stride."""
    if t.stride(-1) != 1:
        raise RuntimeError("last dimension must be contiguous")
    width = t.shape[-1]
    rows = t.numel() // max(width, 1)
    if t.dim() == 1:
        return 1, width
    stride = t.stride(-2)
    expect = stride
    for d in range(t.dim() - 3, -1, -1):
        expect *= t.shape[d + 1]
        if t.shape[d] != 1 and t.stride(d) != expect:
            raise RuntimeError("leading dimensions must collapse to a single row stride")
    return rows, stride


def rms_norm(out: torch.Tensor, input: torch.Tensor, weight: torch.Tensor, epsilon: float) -> None:
    """``ops.rms_norm`` (csrc/layernorm_kernels.cu:70-92)."""
    _require_cuda(out, input, weight)
    if weight.dtype != input.dtype or out.dtype != input.dtype:
        raise RuntimeError("rms_norm: out, input and weight must share one dtype")
    rows, stride = _rows(input)
    if not out.is_contiguous():
        raise RuntimeError("rms_norm: out must be contiguous")
    with torch.cuda.device(input.device):
        _check(load_library().sw_rms_norm(out.data_ptr(), input.data_ptr(), weight.data_ptr(), float(epsilon),
                                          rows, input.shape[-1], stride, _act_dtype(input), _stream(input)))


def fused_add_rms_norm(input: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor,
                       epsilon: float) -> None:
    """``ops.fused_add_rms_norm`` (csrc/layernorm_kernels.cu:94-118), in place on both."""
    _require_cuda(input, residual, weight)
    if weight.dtype != input.dtype or residual.dtype != input.dtype:
        raise RuntimeError("fused_add_rms_norm: input, residual and weight must share one dtype")
    rows, stride = _rows(input)
    if not residual.is_contiguous():
        raise RuntimeError("fused_add_rms_norm: residual must be contiguous")
    with torch.cuda.device(input.device):
        _check(load_library().sw_fused_add_rms_norm(input.data_ptr(), residual.data_ptr(), weight.data_ptr(),
                                                    float(epsilon), rows, input.shape[-1], stride,
                                                    _act_dtype(input), _stream(input)))


def rotary_embedding(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor, head_size: int,
                     cos_sin_cache: torch.Tensor, is_neox: bool) -> None:
    """``ops.rotary_embedding`` (csrc/pos_encoding_kernels.cu:120-171), in place on query / key."""
    _require_cuda(positions, query, key, cos_sin_cache)
    if positions.dtype != torch.int64 or not positions.is_contiguous():
        raise RuntimeError("rotary_embedding: positions must be contiguous int64")
    if cos_sin_cache.dtype != query.dtype:
        raise RuntimeError("rotary_embedding: cos_sin_cache must have the dtype of query")
    q_rows, q_stride = _rows(query)
    k_rows, k_stride = _rows(key)
    if q_rows != positions.numel() or k_rows != q_rows:
        raise RuntimeError("rotary_embedding: positions / query / key disagree on the token count")
    with torch.cuda.device(query.device):
        _check(load_library().sw_rotary_embedding(
            positions.data_ptr(), query.data_ptr(), key.data_ptr(), q_rows, query.shape[-1] // head_size,
            key.shape[-1] // head_size, head_size, cos_sin_cache.data_ptr(), cos_sin_cache.shape[1],
            1 if is_neox else 0, q_stride, k_stride, _act_dtype(query), _stream(query)))


def rotary_embedding_and_cache(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                               key_cache: torch.Tensor, value_cache: torch.Tensor, slot_mapping: torch.Tensor,
                               head_size: int, cos_sin_cache: torch.Tensor, is_neox: bool, kv_cache_dtype: str) -> None:
    """``ops.rotary_embedding`` then ``cache_ops.reshape_and_cache`` on the same tokens, one launch
    (decode steps); no single counterpart in the reference, results identical to the two calls."""
    _require_cuda(positions, query, key, value, key_cache, value_cache, slot_mapping, cos_sin_cache)
    if positions.dtype != torch.int64 or not positions.is_contiguous():
        raise RuntimeError("rotary_embedding_and_cache: positions must be contiguous int64")
    if slot_mapping.dtype != torch.int64 or not slot_mapping.is_contiguous():
        raise RuntimeError("rotary_embedding_and_cache: slot_mapping must be contiguous int64")
    if cos_sin_cache.dtype != query.dtype:
        raise RuntimeError("rotary_embedding_and_cache: cos_sin_cache must have the dtype of query")
    q_rows, q_stride = _rows(query)
    k_rows, k_stride = _rows(key)
    v_rows, v_stride = _rows(value)
    if not (q_rows == k_rows == v_rows == positions.numel() == slot_mapping.numel()):
        raise RuntimeError("rotary_embedding_and_cache: positions / slots / query / key / value disagree on the "
                           "token count")
    act = _act_dtype(query)
    with torch.cuda.device(query.device):
        _check(load_library().sw_rotary_embedding_and_cache(
            positions.data_ptr(), query.data_ptr(), key.data_ptr(), value.data_ptr(), key_cache.data_ptr(),
            value_cache.data_ptr(), slot_mapping.data_ptr(), q_rows, query.shape[-1] // head_size,
            key.shape[-1] // head_size, head_size, cos_sin_cache.data_ptr(), cos_sin_cache.shape[1],
            1 if is_neox else 0, key_cache.shape[3], key_cache.shape[4], q_stride, k_stride, v_stride, act,
            _cache_dtype(kv_cache_dtype, act), _stream(query)))


def silu_and_mul(out: torch.Tensor, input: torch.Tensor) -> None:
    """``ops.silu_and_mul`` (csrc/activation_kernels.cu:66-84)."""
    _require_cuda(out, input)
    rows, stride = _rows(input)
    if not out.is_contiguous():
        raise RuntimeError("silu_and_mul: out must be contiguous")
    with torch.cuda.device(input.device):
        _check(load_library().sw_silu_and_mul(out.data_ptr(), input.data_ptr(), rows, input.shape[-1] // 2,
                                              stride, _act_dtype(input), _stream(input)))


# ------------------------------------------------------------------ KV transport
def ipc_get_mem_handle(dev_ptr: int) -> bytes:
    buf = ctypes.create_string_buffer(64)
    _check(load_library().sw_ipc_get_mem_handle(dev_ptr, buf))
    return buf.raw


def mem_address_range(dev_ptr: int):
    """(base address, size in bytes) of the device allocation that contains ``dev_ptr``."""
    base, size = _c_void_p(), _c_i64()
    _check(load_library().sw_mem_address_range(dev_ptr, ctypes.byref(base), ctypes.byref(size)))
    return base.value, size.value


def ipc_open_mem_handle(handle: bytes) -> int:
    out = _c_void_p()
    _check(load_library().sw_ipc_open_mem_handle(ctypes.c_char_p(handle), ctypes.byref(out)))
    return out.value


def ipc_close_mem_handle(dev_ptr: int) -> None:
    _check(load_library().sw_ipc_close_mem_handle(dev_ptr))


# ------------------------------------------------------------------ all-reduce by peer reads
def custom_ar_signal_bytes() -> int:
    return int(load_library().sw_custom_ar_signal_bytes())


def custom_ar_init(peer_bases: List[int], rank: int, signal_off: int, scratch_off: int, staging_off: int,
                   max_bytes: int) -> int:
    arr = (_c_void_p * len(peer_bases))(*peer_bases)
    state = _c_void_p()
    _check(load_library().sw_custom_ar_init(arr, rank, len(peer_bases), signal_off, scratch_off, staging_off,
                                            max_bytes, ctypes.byref(state)))
    return state.value


def custom_all_reduce(state: int, inp: torch.Tensor, out: torch.Tensor, algorithm: int = 0) -> None:
    if not (inp.is_contiguous() and out.is_contiguous()) or inp.shape != out.shape or inp.dtype != out.dtype:
        raise RuntimeError("custom_all_reduce: input and output must be contiguous tensors of one shape and dtype")
    _check(load_library().sw_custom_all_reduce(state, inp.data_ptr(), out.data_ptr(), inp.numel() * inp.element_size(),
                                               _act_dtype(inp), algorithm, _stream(inp)))


def custom_ar_dispose(state: int) -> None:
    _check(load_library().sw_custom_ar_dispose(state))


def kv_push(dst_base: int, src_base: int, offsets: List[int], nbytes: List[int], stream: int,
            max_workgroups: int = 0) -> None:
    n = len(offsets)
    if n == 0:
        return
    off = (ctypes.c_int64 * n)(*offsets)
    nb = (ctypes.c_int64 * n)(*nbytes)
    _check(load_library().sw_kv_push(dst_base, src_base, ctypes.addressof(off), ctypes.addressof(nb), n,
                                     int(max_workgroups), stream))


def kv_push_mapped(dst_base: int, src_base: int, dst_offsets: List[int], src_offsets: List[int], nbytes: List[int],
                   stream: int, max_workgroups: int = 0) -> None:
    """kv_push with a destination offset of its own per range."""
    n = len(dst_offsets)
    if n == 0:
        return
    if len(src_offsets) != n or len(nbytes) != n:
        raise RuntimeError("kv_push_mapped: offset and size lists differ in length")
    dst = (ctypes.c_int64 * n)(*dst_offsets)
    src = (ctypes.c_int64 * n)(*src_offsets)
    nb = (ctypes.c_int64 * n)(*nbytes)
    _check(load_library().sw_kv_push_mapped(dst_base, src_base, ctypes.addressof(dst), ctypes.addressof(src),
                                            ctypes.addressof(nb), n, int(max_workgroups), stream))


def kv_signal(flags_ptr: int, slot: int, epoch: int, stream: int) -> None:
    _check(load_library().sw_kv_signal(flags_ptr, slot, epoch, stream))


def kv_wait(flags_ptr: int, slot: int, epoch: int, max_spins: int, status_ptr: int, stream: int) -> None:
    _check(load_library().sw_kv_wait(flags_ptr, slot, epoch, max_spins, status_ptr, stream))



# ------------------------------------------------------------- int4 linear layers
# Above this many rows a dense fp16 GEMM on the reconstructed weight beats streaming the int4
# weight once per 64 rows; the reference switches at 50 rows (q_gemm.cu:1707, MAX_Q_GEMM_ROWS)
# and, for AWQ, at 256 tokens (awq.py:161).
GPTQ_FUSED_MAX_ROWS = 256
_WORKSPACE_MIN_BYTES = 64 << 20
_workspaces: Dict[tuple, List[torch.Tensor]] = {}


def _quant_workspace(ref: torch.Tensor, m: int, n: int, k: int) -> torch.Tensor:
    """Zero-initialised scratch for the fused GEMMs, one per (device, stream); replaced by a
    larger one on demand (older ones stay alive: captured graphs may still name them)."""
    want = min(int(load_library().sw_quant_gemm_workspace_bytes(m, n, k)), 1 << 30)
    key = (ref.device.index, torch.cuda.current_stream(ref.device).cuda_stream)
    pool = _workspaces.setdefault(key, [])
    if not pool or pool[-1].numel() < want:
        pool.append(torch.zeros(max(want, _WORKSPACE_MIN_BYTES), dtype=torch.uint8, device=ref.device))
    return pool[-1]


def _optional_index(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """The reference passes a meta-device tensor where it means "no index" (gptq.py:206-214)."""
    if t is None or t.device.type == "meta" or t.numel() == 0:
        return None
    if t.dtype != torch.int32:
        raise RuntimeError("index tensors must be int32")
    return t


def _require_half(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t.dtype != torch.float16:
            raise RuntimeError(f"int4 linear layers support float16 only, got {t.dtype}")


def gptq_shuffle(q_weight: torch.Tensor, q_perm: Optional[torch.Tensor], bit: int) -> None:
    """``ops.gptq_shuffle`` (csrc/quantization/gptq/q_gemm.cu:2748-2769): in place."""
    _require_cuda(q_weight)
    perm = _optional_index(q_perm)
    scratch = torch.empty_like(q_weight) if perm is not None else None
    k, n = q_weight.size(0) * 32 // bit, q_weight.size(1)
    _check(load_library().sw_gptq_shuffle(q_weight.data_ptr(), perm.data_ptr() if perm is not None else None, k, n,
                                          int(bit), scratch.data_ptr() if scratch is not None else None,
                                          _stream(q_weight)))


def gptq_dequant(b_q_weight: torch.Tensor, b_gptq_qzeros: torch.Tensor, b_gptq_scales: torch.Tensor,
                 b_g_idx: Optional[torch.Tensor], bits: int, use_exllama: bool) -> torch.Tensor:
    """``ops.dequant_gptq`` (q_gemm.cu:2820-2861): the fp16 weight [K, N]."""
    _require_cuda(b_q_weight, b_gptq_qzeros, b_gptq_scales)
    _require_half(b_gptq_scales)
    idx = _optional_index(b_g_idx)
    k, n = b_q_weight.size(0) * 32 // bits, b_q_weight.size(1)
    out = torch.empty((k, n), dtype=torch.float16, device=b_q_weight.device)
    _check(load_library().sw_gptq_dequant(out.data_ptr(), b_q_weight.data_ptr(), b_gptq_qzeros.data_ptr(),
                                          b_gptq_scales.data_ptr(), idx.data_ptr() if idx is not None else None, k, n,
                                          b_gptq_qzeros.size(0), int(bool(use_exllama)), int(bits),
                                          _stream(b_q_weight)))
    return out


def gptq_gemm(a: torch.Tensor, b_q_weight: torch.Tensor, b_gptq_qzeros: torch.Tensor, b_gptq_scales: torch.Tensor,
              b_g_idx: Optional[torch.Tensor], use_exllama: bool, bit: int) -> torch.Tensor:
    """``ops.gptq_gemm`` (q_gemm.cu:2712-2746): a [M, K] fp16 -> [M, N] fp16."""
    _require_cuda(a, b_q_weight, b_gptq_qzeros, b_gptq_scales)
    _require_half(a, b_gptq_scales)
    if not a.is_contiguous():
        a = a.contiguous()
    m, k = a.shape
    n = b_q_weight.size(1)
    if k != b_q_weight.size(0) * 32 // bit:
        raise RuntimeError(f"gptq_gemm: input has {k} columns, weight has {b_q_weight.size(0) * 32 // bit} rows")
    # 8-bit weights: reconstruction + dense GEMM above 24 rows, as the reference (MAX_Q_GEMM_ROWS_8BIT, q_gemm.cu:1707)
    if m > (GPTQ_FUSED_MAX_ROWS if bit != 8 else 24):
        return torch.matmul(a, gptq_dequant(b_q_weight, b_gptq_qzeros, b_gptq_scales, b_g_idx, bit, use_exllama))
    idx = _optional_index(b_g_idx)
    c = torch.empty((m, n), dtype=torch.float16, device=a.device)
    ws = _quant_workspace(a, m, n, k)
    _check(load_library().sw_gptq_gemm(c.data_ptr(), a.data_ptr(), b_q_weight.data_ptr(), b_gptq_qzeros.data_ptr(),
                                       b_gptq_scales.data_ptr(), idx.data_ptr() if idx is not None else None, m, n, k,
                                       b_gptq_qzeros.size(0), int(bool(use_exllama)), int(bit), ws.data_ptr(),
                                       ws.numel(), _stream(a)))
    return c


def awq_dequantize(qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor, split_k_iters: int = 0,
                   thx: int = 0, thy: int = 0) -> torch.Tensor:
    """``ops.awq_dequantize`` (csrc/quantization/awq/gemm_kernels.cu:585-640); the three launch
    hints of the reference are accepted and ignored."""
    del split_k_iters, thx, thy
    _require_cuda(qweight, scales, qzeros)
    _require_half(scales)
    k, n = qweight.size(0), qweight.size(1) * 8
    out = torch.empty((k, n), dtype=torch.float16, device=qweight.device)
    _check(load_library().sw_awq_dequantize(out.data_ptr(), qweight.data_ptr(), scales.data_ptr(), qzeros.data_ptr(),
                                            k, n, k // scales.size(0), _stream(qweight)))
    return out


def awq_gemm(x: torch.Tensor, qweight: torch.Tensor, scales: torch.Tensor, qzeros: torch.Tensor,
             split_k_iters: int) -> torch.Tensor:
    """``ops.awq_gemm`` (gemm_kernels.cu:648-705): x [M, K] fp16 -> [M, N] fp16.  ``split_k_iters``
    sized the reference's partial-output tensor; the split is chosen natively here."""
    del split_k_iters
    _require_cuda(x, qweight, scales, qzeros)
    _require_half(x, scales)
    if not x.is_contiguous():
        x = x.contiguous()
    m, k = x.shape
    n = qweight.size(1) * 8
    group_size = k // scales.size(0)
    # the reference throws std::invalid_argument for these (gemm_kernels.cu:671-678)
    if n % 64 != 0:
        raise ValueError("OC is not multiple of cta_N = 64")
    if group_size % 32 != 0:
        raise ValueError("Group size should be a multiple of 32")
    if n % group_size != 0:
        raise ValueError("OC is not multiple of Group size")
    c = torch.empty((m, n), dtype=torch.float16, device=x.device)
    ws = _quant_workspace(x, m, n, k)
    _check(load_library().sw_awq_gemm(c.data_ptr(), x.data_ptr(), qweight.data_ptr(), scales.data_ptr(),
                                      qzeros.data_ptr(), m, k, n, group_size, ws.data_ptr(), ws.numel(), _stream(x)))
    return c


def awq_to_kmajor(qweight: torch.Tensor, qzeros: torch.Tensor):
    """AWQ tensors (qweight [K, N/8], qzeros [groups, N/8]) -> the k-major pair
    (qweight [K/8, N], qzeros [groups, N/8] in plain column order) that ``int4_kmajor_gemm`` and
    ``int4_kmajor_dequant`` take with ``zero_offset=0``.  No counterpart in the reference."""
    _require_cuda(qweight, qzeros)
    k, n = qweight.size(0), qweight.size(1) * 8
    if k % 8 != 0:
        raise RuntimeError(f"awq_to_kmajor: {k} rows are not a multiple of 8")
    qweight_k = torch.empty((k // 8, n), dtype=torch.int32, device=qweight.device)
    qzeros_k = torch.empty_like(qzeros)
    _check(load_library().sw_awq_to_kmajor(qweight_k.data_ptr(), qzeros_k.data_ptr(), qweight.contiguous().data_ptr(),
                                           qzeros.contiguous().data_ptr(), k, n, qzeros.size(0), _stream(qweight)))
    return qweight_k, qzeros_k


def int4_kmajor_dequant(qweight: torch.Tensor, qzeros: torch.Tensor, scales: torch.Tensor,
                        zero_offset: int) -> torch.Tensor:
    """fp16 weight [K, N] of k-major int4 tensors (GPTQ after gptq_shuffle: zero_offset 1; AWQ
    after awq_to_kmajor: zero_offset 0)."""
    _require_cuda(qweight, qzeros, scales)
    _require_half(scales)
    k, n = qweight.size(0) * 8, qweight.size(1)
    out = torch.empty((k, n), dtype=torch.float16, device=qweight.device)
    _check(load_library().sw_int4_kmajor_dequant(out.data_ptr(), qweight.data_ptr(), qzeros.data_ptr(),
                                                 scales.data_ptr(), k, n, qzeros.size(0), int(zero_offset),
                                                 _stream(qweight)))
    return out


def int4_kmajor_gemm(a: torch.Tensor, qweight: torch.Tensor, qzeros: torch.Tensor, scales: torch.Tensor,
                     zero_offset: int) -> torch.Tensor:
    """a [M, K] fp16 times k-major int4 weights -> [M, N] fp16; above the fused kernels' row limit
    the weight is reconstructed and a dense GEMM runs, as in ``gptq_gemm``."""
    _require_cuda(a, qweight, qzeros, scales)
    _require_half(a, scales)
    if not a.is_contiguous():
        a = a.contiguous()
    m, k = a.shape
    n = qweight.size(1)
    if k != qweight.size(0) * 8:
        raise RuntimeError(f"int4_kmajor_gemm: input has {k} columns, weight has {qweight.size(0) * 8} rows")
    if m > GPTQ_FUSED_MAX_ROWS:
        return torch.matmul(a, int4_kmajor_dequant(qweight, qzeros, scales, zero_offset))
    c = torch.empty((m, n), dtype=torch.float16, device=a.device)
    ws = _quant_workspace(a, m, n, k)
    _check(load_library().sw_int4_kmajor_gemm(c.data_ptr(), a.data_ptr(), qweight.data_ptr(), qzeros.data_ptr(),
                                              scales.data_ptr(), m, n, k, qzeros.size(0), int(zero_offset),
                                              ws.data_ptr(), ws.numel(), _stream(a)))
    return c


# ------------------------------------------------------------------- cuda_utils
def get_max_shared_memory_per_block_device_attribute(gpu: int) -> int:
    """``cuda_utils.get_max_shared_memory_per_block_device_attribute``
    (csrc/cuda_utils_kernels.cu:16-35)."""
    out = ctypes.c_int(0)
    _check(load_library().sw_get_max_shared_memory_per_block(int(gpu), ctypes.byref(out)))
    return out.value


ops = SimpleNamespace(
    paged_attention_v1=paged_attention_v1,
    paged_attention_v2=paged_attention_v2,
    paged_attention_v2_static=paged_attention_v2_static,
    prefill_attention=prefill_attention,
    context_attention_fwd=context_attention_fwd,
    rms_norm=rms_norm,
    fused_add_rms_norm=fused_add_rms_norm,
    rotary_embedding=rotary_embedding,
    rotary_embedding_and_cache=rotary_embedding_and_cache,
    silu_and_mul=silu_and_mul,
    gptq_shuffle=gptq_shuffle,
    gptq_gemm=gptq_gemm,
    dequant_gptq=gptq_dequant,
    awq_gemm=awq_gemm,
    awq_to_kmajor=awq_to_kmajor,
    int4_kmajor_gemm=int4_kmajor_gemm,
    int4_kmajor_dequant=int4_kmajor_dequant,
    awq_dequantize=awq_dequantize,
)
cache_ops = SimpleNamespace(
    reshape_and_cache=reshape_and_cache,
    copy_blocks=copy_blocks,
    swap_blocks=swap_blocks,
    convert_fp8_e5m2=convert_fp8_e5m2,
)
custom_ar = SimpleNamespace(
    signal_bytes=custom_ar_signal_bytes,
    init=custom_ar_init,
    all_reduce=custom_all_reduce,
    dispose=custom_ar_dispose,
)
kv_ops = SimpleNamespace(
    ipc_get_mem_handle=ipc_get_mem_handle,
    mem_address_range=mem_address_range,
    ipc_open_mem_handle=ipc_open_mem_handle,
    ipc_close_mem_handle=ipc_close_mem_handle,
    kv_push=kv_push,
    kv_push_mapped=kv_push_mapped,
    kv_signal=kv_signal,
    kv_wait=kv_wait,
)
cuda_utils = SimpleNamespace(
    get_max_shared_memory_per_block_device_attribute=get_max_shared_memory_per_block_device_attribute,
)
