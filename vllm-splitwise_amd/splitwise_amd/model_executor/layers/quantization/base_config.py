"""Base class of a weight-quantisation scheme (vllm/model_executor/layers/quantization/base_config.py:8-68)."""
from typing import Any, Dict, List

import torch


class QuantizationConfig:
    def get_name(self) -> str:
        raise NotImplementedError

    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        raise NotImplementedError

    @staticmethod
    def get_config_filenames() -> List[str]:
        raise NotImplementedError

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "QuantizationConfig":
        raise NotImplementedError

    @staticmethod
    def get_from_keys(config: Dict[str, Any], keys: List[str]) -> Any:
        for key in keys:
            if key in config:
                return config[key]
        raise ValueError(f"Cannot find any of {keys} in the model's quantization config.")

    def get_linear_method(self):
        raise NotImplementedError

    def get_scaled_act_names(self) -> List[str]:
        raise NotImplementedError
