"""GPTQ linear layers (vllm/model_executor/layers/quantization/gptq.py:18-222).

4-bit (matrix-core kernels) and 8-bit (plain kernels): the native library provides no 2/3-bit kernels.  Parameter shapes, the lazy
shuffle on the first forward pass and the act-order rules are the reference's.
"""
import enum
from typing import Any, Dict, List, Optional

import torch
from torch.nn.parameter import Parameter

from .... import _C
from ..linear import LinearMethodBase, set_weight_attrs
from .base_config import QuantizationConfig
from .dense_copy import dense_weight


class GPTQConfig(QuantizationConfig):
    def __init__(self, weight_bits: int, group_size: int, desc_act: bool) -> None:
        self.weight_bits = weight_bits
        self.group_size = group_size
        self.desc_act = desc_act
        if weight_bits not in (2, 3, 4, 8):
            raise ValueError("Currently, only 2/3/4/8-bit weight quantization is supported for GPTQ, but got "
                             f"{weight_bits} bits.")
        if weight_bits not in (4, 8):
            raise ValueError(f"this build provides 4-bit and 8-bit GPTQ kernels only, got {weight_bits} bits")
        self.pack_factor = 32 // weight_bits

    def __repr__(self) -> str:
        return f"GPTQConfig(weight_bits={self.weight_bits}, group_size={self.group_size}, desc_act={self.desc_act})"

    def get_name(self) -> str:
        return "gptq"

    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        return [torch.half]

    @staticmethod
    def get_config_filenames() -> List[str]:
        return ["quantize_config.json"]

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "GPTQConfig":
        return cls(cls.get_from_keys(config, ["bits"]), cls.get_from_keys(config, ["group_size"]),
                   cls.get_from_keys(config, ["desc_act"]))

    def get_linear_method(self) -> "GPTQLinearMethod":
        return GPTQLinearMethod(self)

    def get_scaled_act_names(self) -> List[str]:
        return []


class ExllamaState(enum.Enum):
    UNUSED = enum.auto()
    UNINITIALIZED = enum.auto()
    READY = enum.auto()


class GPTQLinearMethod(LinearMethodBase):
    def __init__(self, quant_config: GPTQConfig) -> None:
        self.quant_config = quant_config

    def create_weights(self, input_size_per_partition: int, output_size_per_partition: int, input_size: int,
                       output_size: int, params_dtype: torch.dtype) -> Dict[str, Any]:
        del output_size
        cfg = self.quant_config
        if cfg.group_size != -1 and input_size_per_partition % cfg.group_size != 0:
            raise ValueError("The input size is not aligned with the quantized weight shape. This can be caused "
                             "by too large tensor parallel size.")
        if output_size_per_partition % cfg.pack_factor != 0:
            raise ValueError("The output size is not aligned with the quantized weight shape. This can be caused "
                             "by too large tensor parallel size.")
        group_size = cfg.group_size if cfg.group_size != -1 else input_size
        exllama_state = ExllamaState.UNINITIALIZED
        scale_and_zero_size = input_size // group_size
        scale_and_zero_input_dim = None
        if input_size != input_size_per_partition and cfg.group_size != -1:
            if cfg.desc_act:
                # act-order rows of a row-parallel shard may belong to any group: keep g_idx
                exllama_state = ExllamaState.UNUSED
            else:
                scale_and_zero_size = input_size_per_partition // group_size
                scale_and_zero_input_dim = 0
        qweight = Parameter(torch.empty(input_size_per_partition // cfg.pack_factor, output_size_per_partition,
                                        dtype=torch.int32), requires_grad=False)
        set_weight_attrs(qweight, {"input_dim": 0, "output_dim": 1, "packed_dim": 0, "pack_factor": cfg.pack_factor})
        g_idx = Parameter(torch.arange(input_size_per_partition, dtype=torch.int32) // group_size, requires_grad=False)
        set_weight_attrs(g_idx, {"input_dim": 0, "ignore_warning": True})
        qzeros = Parameter(torch.empty(scale_and_zero_size, output_size_per_partition // cfg.pack_factor,
                                       dtype=torch.int32), requires_grad=False)
        set_weight_attrs(qzeros, {"input_dim": scale_and_zero_input_dim, "output_dim": 1, "packed_dim": 1,
                                  "pack_factor": cfg.pack_factor})
        scales = Parameter(torch.empty(scale_and_zero_size, output_size_per_partition, dtype=params_dtype),
                           requires_grad=False)
        set_weight_attrs(scales, {"input_dim": scale_and_zero_input_dim, "output_dim": 1})
        return {"qweight": qweight, "g_idx": g_idx, "qzeros": qzeros, "scales": scales,
                "exllama_state": exllama_state}

    def apply_weights(self, weights: Dict[str, Any], x: torch.Tensor,
                      bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        qweight = weights["qweight"]
        out_shape = x.shape[:-1] + (qweight.shape[-1],)
        reshaped_x = x.reshape(-1, x.shape[-1])
        if weights["exllama_state"] == ExllamaState.UNINITIALIZED:
            # the kernels read a shuffled weight: done once, on the first forward pass
            if self.quant_config.desc_act:
                weights["g_idx"] = torch.argsort(weights["g_idx"]).to(torch.int)
            else:
                weights["g_idx"] = torch.empty((1, 1), device="meta")
            weights["exllama_state"] = ExllamaState.READY
            _C.ops.gptq_shuffle(weights["qweight"], weights["g_idx"], self.quant_config.weight_bits)
        ready = weights["exllama_state"] == ExllamaState.READY
        dense = None
        fused_rows = 24 if self.quant_config.weight_bits == 8 else _C.GPTQ_FUSED_MAX_ROWS     # q_gemm.cu:1707
        if reshaped_x.shape[0] > fused_rows and reshaped_x.is_cuda:
            # many rows: reconstruct + dense GEMM, as ops.gptq_gemm does -- with the reconstruction kept
            dense = dense_weight(weights, reshaped_x.shape[-1], qweight.shape[-1],
                                 lambda: _C.ops.dequant_gptq(weights["qweight"], weights["qzeros"], weights["scales"],
                                                             weights["g_idx"], self.quant_config.weight_bits, ready))
        if dense is not None:
            output = torch.matmul(reshaped_x, dense)
        else:
            output = _C.ops.gptq_gemm(reshaped_x, weights["qweight"], weights["qzeros"], weights["scales"],
                                      weights["g_idx"], ready, self.quant_config.weight_bits)
        if bias is not None:
            output = output + bias
        return output.reshape(out_shape)
