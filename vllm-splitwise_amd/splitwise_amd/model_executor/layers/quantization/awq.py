"""AWQ linear layers (vllm/model_executor/layers/quantization/awq.py:12-172)."""
from typing import Any, Dict, List, Optional

import torch
from torch.nn.parameter import Parameter

from .... import _C
from ..linear import LinearMethodBase, set_weight_attrs
from .base_config import QuantizationConfig
from .dense_copy import dense_weight


class AWQConfig(QuantizationConfig):
    def __init__(self, weight_bits: int, group_size: int, zero_point: bool) -> None:
        self.weight_bits = weight_bits
        self.group_size = group_size
        self.zero_point = zero_point
        if weight_bits != 4:
            raise ValueError(f"Currently, only 4-bit weight quantization is supported for AWQ, but got {weight_bits} "
                             "bits.")
        self.pack_factor = 32 // weight_bits

    def __repr__(self) -> str:
        return f"AWQConfig(weight_bits={self.weight_bits}, group_size={self.group_size}, zero_point={self.zero_point})"

    def get_name(self) -> str:
        return "awq"

    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        return [torch.half]

    @staticmethod
    def get_config_filenames() -> List[str]:
        return ["quant_config.json", "quantize_config.json"]

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "AWQConfig":
        return cls(cls.get_from_keys(config, ["w_bit", "bits"]), cls.get_from_keys(config, ["q_group_size", "group_size"]),
                   cls.get_from_keys(config, ["zero_point"]))

    def get_linear_method(self) -> "AWQLinearMethod":
        return AWQLinearMethod(self)

    def get_scaled_act_names(self) -> List[str]:
        return ["gelu", "gelu_fast", "gelu_new", "gelu_pytorch_tanh"]


class AWQLinearMethod(LinearMethodBase):
    """``repack`` (default on; ``SPLITWISE_AWQ_REPACK=0`` turns it off): at the first use of a layer
    its checkpoint tensors are rewritten once into the k-major layout of the GPTQ kernels, which
    then serve it (same fp16 weights bit for bit, a faster path to the matrix cores at every row
    count).  Like GPTQ's shuffle (gptq.py:205-212) this happens in place: afterwards ``qweight`` is
    [K/8, N] and ``qzeros`` holds plain column order, and ``weights["_kmajor"]`` is set.  Off, every call goes through ``ops.awq_gemm`` /
    ``ops.awq_dequantize`` on the checkpoint layout, as in the reference (awq.py:153-172)."""

    def __init__(self, quant_config: AWQConfig, repack: Optional[bool] = None) -> None:
        import os
        self.quant_config = quant_config
        self.repack = os.environ.get("SPLITWISE_AWQ_REPACK", "1") != "0" if repack is None else bool(repack)

    def create_weights(self, input_size_per_partition: int, output_size_per_partition: int, input_size: int,
                       output_size: int, params_dtype: torch.dtype) -> Dict[str, Any]:
        cfg = self.quant_config
        if input_size_per_partition % cfg.group_size != 0:
            raise ValueError("The input size is not aligned with the quantized weight shape. This can be caused "
                             "by too large tensor parallel size.")
        if output_size_per_partition % cfg.pack_factor != 0:
            raise ValueError("The output size is not aligned with the quantized weight shape. This can be caused "
                             "by too large tensor parallel size.")
        qweight = Parameter(torch.empty(input_size_per_partition, output_size_per_partition // cfg.pack_factor,
                                        dtype=torch.int32), requires_grad=False)
        set_weight_attrs(qweight, {"input_dim": 0, "output_dim": 1, "packed_dim": 1, "pack_factor": cfg.pack_factor})
        qzeros = Parameter(torch.empty(input_size_per_partition // cfg.group_size,
                                       output_size_per_partition // cfg.pack_factor, dtype=torch.int32),
                           requires_grad=False)
        set_weight_attrs(qzeros, {"input_dim": 0, "output_dim": 1, "packed_dim": 1, "pack_factor": cfg.pack_factor})
        scales = Parameter(torch.empty(input_size_per_partition // cfg.group_size, output_size_per_partition,
                                       dtype=params_dtype), requires_grad=False)
        set_weight_attrs(scales, {"input_dim": 0, "output_dim": 1})
        return {"qweight": qweight, "qzeros": qzeros, "scales": scales}

    def apply_weights(self, weights: Dict[str, Any], x: torch.Tensor,
                      bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        qweight, scales, qzeros = weights["qweight"], weights["scales"], weights["qzeros"]
        pack_factor = self.quant_config.pack_factor
        reshaped_x = x.reshape(-1, x.shape[-1])
        if self.repack and x.is_cuda:
            if not weights.get("_kmajor"):
                qweight.data, qzeros.data = _C.ops.awq_to_kmajor(qweight.data, qzeros.data)
                weights["_kmajor"] = True
            dense = None
            if reshaped_x.shape[0] > _C.GPTQ_FUSED_MAX_ROWS:
                # many rows: reconstruct + dense GEMM, as ops.int4_kmajor_gemm does -- with the reconstruction kept
                dense = dense_weight(weights, reshaped_x.shape[-1], qweight.shape[-1],
                                     lambda: _C.ops.int4_kmajor_dequant(qweight, qzeros, scales, 0))
            if dense is not None:
                out = torch.matmul(reshaped_x, dense)
            else:
                out = _C.ops.int4_kmajor_gemm(reshaped_x, qweight, qzeros, scales, 0)
            if bias is not None:
                out = out + bias
            return out.reshape(x.shape[:-1] + (qweight.shape[-1],))
        out_shape = x.shape[:-1] + (qweight.shape[-1] * pack_factor,)
        if x.shape[:-1].numel() >= 256:                   # many tokens: dense GEMM on the fp16 weight
            out = torch.matmul(reshaped_x, _C.ops.awq_dequantize(qweight, scales, qzeros, 0, 0, 0))
        else:
            out = _C.ops.awq_gemm(reshaped_x, qweight, scales, qzeros, pack_factor)
        if bias is not None:
            out = out + bias
        return out.reshape(out_shape)
