"""Registry of weight-quantisation schemes (vllm/model_executor/layers/quantization/__init__.py:10-21)."""
from typing import Any, Dict, Type

from .awq import AWQConfig, AWQLinearMethod
from .base_config import QuantizationConfig
from .gptq import GPTQConfig, GPTQLinearMethod

_QUANTIZATION_CONFIG_REGISTRY: Dict[str, Type[QuantizationConfig]] = {"awq": AWQConfig, "gptq": GPTQConfig}
# what a checkpoint's quantize_config.json would say; used for the random-weight presets
_DEFAULTS: Dict[str, Dict[str, Any]] = {
    "gptq": {"bits": 4, "group_size": 128, "desc_act": False},
    "awq": {"w_bit": 4, "q_group_size": 128, "zero_point": True},
}


def get_quantization_config(quantization: str) -> Type[QuantizationConfig]:
    if quantization not in _QUANTIZATION_CONFIG_REGISTRY:
        raise ValueError(f"Invalid quantization method: {quantization}")
    return _QUANTIZATION_CONFIG_REGISTRY[quantization]


def get_linear_method(quantization: str, hf_config: Dict[str, Any]):
    """The linear method of a model: its ``quantization_config`` entry when the geometry carries
    one (HF ``config.json``), the scheme's usual settings otherwise."""
    cls = get_quantization_config(quantization)
    raw = dict(_DEFAULTS[quantization])
    raw.update(hf_config.get("quantization_config") or {})
    return cls.from_config(raw).get_linear_method()


__all__ = ["QuantizationConfig", "GPTQConfig", "GPTQLinearMethod", "AWQConfig", "AWQLinearMethod",
           "get_quantization_config", "get_linear_method"]
