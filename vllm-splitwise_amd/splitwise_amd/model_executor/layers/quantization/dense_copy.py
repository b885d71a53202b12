"""fp16 copies of int4 weights for the many-row (prompt) path.

Above the fused kernels' row limit an int4 layer is reconstructed to fp16 and multiplied by the BLAS library, as
in the reference (vllm/model_executor/layers/quantization/gptq.py:203-222, awq.py:153-172 call the same two
steps on every forward pass).  With 288 GB of HBM per GPU the reconstruction does not have to be repeated: the
first many-row call of a layer keeps its fp16 weight, up to a budget shared by the process, and later calls go
straight to the GEMM (a 13B model: 26 GB kept, 34 us x 4 matrices x 40 layers saved per prompt step).  The values
are those the reconstruction kernel writes either way; decode steps keep reading the int4 weights.
"""
import os
import weakref
from typing import Any, Callable, Dict, Optional

import torch

# GiB of fp16 copies a process may keep (0 turns the cache off)
_BUDGET_BYTES = int(float(os.environ.get("SPLITWISE_INT4_DENSE_COPY_GIB", "48")) * (1 << 30))
_used_bytes = 0


def _release(nbytes: int) -> None:
    global _used_bytes
    _used_bytes -= nbytes


def dense_weight(weights: Dict[str, Any], rows: int, cols: int,
                 reconstruct: Callable[[], torch.Tensor]) -> Optional[torch.Tensor]:
    """The layer's [rows, cols] fp16 weight: the kept copy, or a new one that is kept if the budget allows;
    None when the budget is spent (the caller reconstructs per call)."""
    global _used_bytes
    kept = weights.get("_dense_copy")
    if kept is not None:
        return kept
    nbytes = rows * cols * 2
    if weights.get("_dense_copy_refused") or _used_bytes + nbytes > _BUDGET_BYTES:
        weights["_dense_copy_refused"] = True
        return None
    kept = reconstruct()
    weights["_dense_copy"] = kept
    _used_bytes += nbytes
    weakref.finalize(kept, _release, nbytes)
    return kept
