"""Linear layers with a pluggable weight format.

The plug-in point mirrors vllm/model_executor/layers/linear.py:29-76: a ``LinearMethodBase``
creates the parameters of a layer (``create_weights``) and applies them (``apply_weights``);
quantised formats (GPTQ, AWQ) register their own method.  Tensor parallelism is expressed by
the layer kind: column-parallel layers shard the output dimension, row-parallel layers the
input dimension and all-reduce the result over the TP group (:580-633).
"""
from typing import Any, Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..parallel_state import (get_tensor_model_parallel_rank, get_tensor_model_parallel_world_size,
                              tensor_model_parallel_all_reduce)


def set_weight_attrs(weight: torch.Tensor, attrs: Optional[Dict[str, Any]]) -> None:
    """Metadata a checkpoint loader reads to shard a parameter (input_dim / output_dim /
    packed_dim / pack_factor), as vllm/model_executor/utils.py:20-35."""
    if attrs is None:
        return
    for key, value in attrs.items():
        assert not hasattr(weight, key), f"Overwriting existing tensor attribute: {key}"
        setattr(weight, key, value)


class LinearMethodBase:
    def create_weights(self, input_size_per_partition: int, output_size_per_partition: int, input_size: int,
                       output_size: int, params_dtype: torch.dtype) -> Dict[str, Any]:
        raise NotImplementedError

    def apply_weights(self, weights: Dict[str, torch.Tensor], x: torch.Tensor,
                      bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        raise NotImplementedError


class UnquantizedLinearMethod(LinearMethodBase):
    """Dense weights [out, in]; the GEMM is the ROCm BLAS library's (hipBLASLt via torch)."""

    def create_weights(self, input_size_per_partition, output_size_per_partition, input_size, output_size,
                       params_dtype):
        weight = nn.Parameter(torch.empty(output_size_per_partition, input_size_per_partition,
                                          dtype=params_dtype), requires_grad=False)
        set_weight_attrs(weight, {"input_dim": 1, "output_dim": 0})
        return {"weight": weight}

    def apply_weights(self, weights, x, bias=None):
        return F.linear(x, weights["weight"], bias)


class _LinearBase(nn.Module):
    def __init__(self, in_per_part: int, out_per_part: int, input_size: int, output_size: int, bias: bool,
                 params_dtype: torch.dtype, linear_method: Optional[LinearMethodBase]) -> None:
        super().__init__()
        self.input_size = input_size
        self.output_size = output_size
        self.linear_method = linear_method or UnquantizedLinearMethod()
        self.linear_weights = self.linear_method.create_weights(in_per_part, out_per_part, input_size,
                                                                output_size, params_dtype)
        for name, w in self.linear_weights.items():
            if isinstance(w, torch.Tensor):
                self.register_parameter(name, w if isinstance(w, nn.Parameter) else nn.Parameter(w, False))
        self.bias = nn.Parameter(torch.zeros(out_per_part, dtype=params_dtype), requires_grad=False) if bias else None

    # ---- placing a FULL (unsharded) tensor into this rank's shard: the job of the per-layer
    # ``weight_loader`` of vllm/model_executor/layers/linear.py:200-227, 291-360, 433-520, 598-617.
    # Sizes are taken in the parameter's own units, so packed int4 dimensions need no special case
    # beyond dividing a fused layer's part sizes by the pack factor.
    def _parts(self, param: torch.Tensor, dim: int):
        """(offset in the full tensor, offset in the shard, size) runs along ``dim``."""
        tp, rank = get_tensor_model_parallel_world_size(), get_tensor_model_parallel_rank()
        size = param.shape[dim]
        return [(rank * size, 0, size)], size * tp

    def _shard_dim(self, param: torch.Tensor) -> Optional[int]:
        return getattr(param, "output_dim", None)

    def full_shape(self, param: torch.Tensor):
        dim = self._shard_dim(param)
        if dim is None or get_tensor_model_parallel_world_size() == 1:
            return tuple(param.shape)
        shape = list(param.shape)
        shape[dim] = self._parts(param, dim)[1]
        return tuple(shape)

    def weight_loader(self, param: torch.Tensor, full: torch.Tensor) -> None:
        dim = self._shard_dim(param)
        if dim is None or get_tensor_model_parallel_world_size() == 1:
            assert param.shape == full.shape, (tuple(param.shape), tuple(full.shape))
            param.data.copy_(full)
            return
        for src, dst, size in self._parts(param, dim)[0]:
            param.data.narrow(dim, dst, size).copy_(full.narrow(dim, src, size))

    def _weights(self) -> Dict[str, Any]:
        # the dict itself: a method may keep per-layer state in it (GPTQ's shuffle flag).  Module.to()
        # moves parameters in place, so the Parameter objects named here stay the live ones.
        return self.linear_weights


class ColumnParallelLinear(_LinearBase):
    """Y = X W^T with W sharded along the output dimension."""

    def __init__(self, input_size: int, output_size: int, bias: bool = False,
                 params_dtype: torch.dtype = torch.float16, linear_method: Optional[LinearMethodBase] = None):
        tp = get_tensor_model_parallel_world_size()
        assert output_size % tp == 0
        super().__init__(input_size, output_size // tp, input_size, output_size, bias, params_dtype, linear_method)
        if self.bias is not None:
            set_weight_attrs(self.bias, {"output_dim": 0})

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.linear_method.apply_weights(self._weights(), x, self.bias)


class MergedColumnParallelLinear(ColumnParallelLinear):
    """Several column-parallel projections of the same input fused into one GEMM (gate_up_proj)."""

    def __init__(self, input_size: int, output_sizes: List[int], bias: bool = False,
                 params_dtype: torch.dtype = torch.float16, linear_method: Optional[LinearMethodBase] = None):
        self.output_sizes = output_sizes
        super().__init__(input_size, sum(output_sizes), bias, params_dtype, linear_method)

    def _parts(self, param: torch.Tensor, dim: int):
        tp, rank = get_tensor_model_parallel_world_size(), get_tensor_model_parallel_rank()
        pack = getattr(param, "pack_factor", 1) if getattr(param, "packed_dim", None) == dim else 1
        runs, src, dst = [], 0, 0
        for full_size in self.output_sizes:
            full_size //= pack
            shard = full_size // tp
            runs.append((src + rank * shard, dst, shard))
            src += full_size
            dst += shard
        return runs, src


class QKVParallelLinear(ColumnParallelLinear):
    """Fused q / k / v projection; kv heads are replicated when there are fewer than TP ranks
    (vllm/model_executor/layers/linear.py:380-420)."""

    def __init__(self, hidden_size: int, head_size: int, total_num_heads: int,
                 total_num_kv_heads: Optional[int] = None, bias: bool = False,
                 params_dtype: torch.dtype = torch.float16, linear_method: Optional[LinearMethodBase] = None):
        tp = get_tensor_model_parallel_world_size()
        total_num_kv_heads = total_num_heads if total_num_kv_heads is None else total_num_kv_heads
        self.num_heads = total_num_heads // tp
        if tp >= total_num_kv_heads:
            self.num_kv_heads, self.num_kv_head_replicas = 1, tp // total_num_kv_heads
        else:
            self.num_kv_heads, self.num_kv_head_replicas = total_num_kv_heads // tp, 1
        out = (self.num_heads + 2 * self.num_kv_heads) * tp * head_size
        self.head_size = head_size
        self.total_num_heads = total_num_heads
        self.total_num_kv_heads = total_num_kv_heads
        super().__init__(hidden_size, out, bias, params_dtype, linear_method)

    def _parts(self, param: torch.Tensor, dim: int):
        # the full tensor is [q: all heads | k: all kv heads | v: all kv heads]; a rank takes its
        # query heads and the kv heads it serves (replicated when there are fewer than ranks)
        rank = get_tensor_model_parallel_rank()
        pack = getattr(param, "pack_factor", 1) if getattr(param, "packed_dim", None) == dim else 1
        d = self.head_size
        q_full, kv_full = self.total_num_heads * d // pack, self.total_num_kv_heads * d // pack
        q_shard, kv_shard = self.num_heads * d // pack, self.num_kv_heads * d // pack
        kv_rank = rank // self.num_kv_head_replicas
        runs = [(rank * q_shard, 0, q_shard),
                (q_full + kv_rank * kv_shard, q_shard, kv_shard),
                (q_full + kv_full + kv_rank * kv_shard, q_shard + kv_shard, kv_shard)]
        return runs, q_full + 2 * kv_full


class RowParallelLinear(_LinearBase):
    """Y = X W^T with W sharded along the input dimension; partial sums are all-reduced."""

    def __init__(self, input_size: int, output_size: int, bias: bool = False,
                 params_dtype: torch.dtype = torch.float16, linear_method: Optional[LinearMethodBase] = None,
                 reduce_results: bool = True):
        tp = get_tensor_model_parallel_world_size()
        assert input_size % tp == 0
        self.reduce_results = reduce_results
        self.tp_size = tp
        super().__init__(input_size // tp, output_size, input_size, output_size, bias, params_dtype, linear_method)

    def _shard_dim(self, param: torch.Tensor) -> Optional[int]:
        return getattr(param, "input_dim", None)          # the bias is replicated

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        out = self.linear_method.apply_weights(self._weights(), x, None)
        if self.reduce_results and self.tp_size > 1:
            out = tensor_model_parallel_all_reduce(out)
        if self.bias is not None:
            out = out + self.bias
        return out
