"""Builds a model of the configured geometry.

There is no network and no checkpoint in this environment, so weights are seeded random
values of the architecture's shapes (a local directory holding ``*.safetensors`` files is
loaded when present).  Initialisation runs on the target device, tensor by tensor.
"""
import glob
import os
import zlib
from typing import Optional

import torch
import torch.nn as nn

from ..config import ModelConfig
from .models.llama import LlamaForCausalLM
from .models.opt import OPTForCausalLM

_ARCH = {"llama": LlamaForCausalLM, "opt": OPTForCausalLM}


def _linear_method(model_config: ModelConfig):
    if model_config.quantization is None:
        return None
    from .layers.quantization import get_linear_method
    method = get_linear_method(model_config.quantization, model_config.hf_config)
    supported = method.quant_config.get_supported_act_dtypes()
    if model_config.dtype not in supported:              # vllm/model_executor/model_loader.py:57-62
        raise ValueError(f"{model_config.dtype} is not supported for quantization method "
                         f"{model_config.quantization}. Supported dtypes: {supported}")
    return method


def _random_full(name: str, shape, dtype: torch.dtype, device: torch.device, seed: int) -> torch.Tensor:
    """The FULL (unsharded) value of one parameter: a function of the seed and the parameter's name
    only, so every tensor-parallel degree shards the same model."""
    g = torch.Generator(device=device).manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 63))
    shape = tuple(shape)
    if name.endswith("g_idx"):
        raise AssertionError("g_idx keeps its constructed value")
    if not dtype.is_floating_point:
        # packed int4 words: uniformly random bits
        return torch.randint(-2 ** 31, 2 ** 31 - 1, shape, dtype=torch.int64, device=device, generator=g).to(dtype)
    if "norm" in name and name.endswith("weight"):
        return torch.ones(shape, dtype=dtype, device=device)
    if name.endswith("bias"):
        return torch.zeros(shape, dtype=dtype, device=device)
    if name.endswith("scales"):
        return (torch.rand(shape, device=device, generator=g) * 2e-3 + 1e-4).to(dtype)
    # N(0, 0.02): chunked through fp32 to bound the temporary
    out = torch.empty(shape, dtype=dtype, device=device)
    flat = out.view(-1)
    step = 1 << 26
    for s in range(0, flat.numel(), step):
        n = min(step, flat.numel() - s)
        flat[s:s + n].copy_((torch.randn(n, device=device, generator=g) * 0.02).to(dtype))
    return out


@torch.no_grad()
def _random_init(model: nn.Module, seed: int) -> None:
    owner = {}
    for mod_name, mod in model.named_modules():
        if hasattr(mod, "weight_loader") and hasattr(mod, "full_shape"):     # sharded layers place their own slice
            for pname, p in mod._parameters.items():
                if p is not None:
                    owner[id(p)] = mod
    for name, p in model.named_parameters():
        if name.endswith("g_idx"):
            continue                                  # keeps its sequential group index
        layer = owner.get(id(p))
        if layer is None:                             # replicated on every rank
            p.copy_(_random_full(name, p.shape, p.dtype, p.device, seed))
        else:
            layer.weight_loader(p, _random_full(name, layer.full_shape(p), p.dtype, p.device, seed))


def get_model(model_config: ModelConfig, device: torch.device) -> nn.Module:
    cls = _ARCH[model_config.architecture]
    # parameters created without an explicit dtype (norm weights) take the model dtype
    # (vllm/model_executor/model_loader.py:24-31 sets the default dtype the same way)
    old_dtype = torch.get_default_dtype()
    torch.set_default_dtype(model_config.dtype)
    try:
        with torch.device(device):
            model = cls(model_config.hf_config, model_config.dtype, _linear_method(model_config))
    finally:
        torch.set_default_dtype(old_dtype)
    files = sorted(glob.glob(os.path.join(model_config.model, "*.safetensors"))) if os.path.isdir(
        model_config.model) else []
    if files:
        from safetensors.torch import load_file
        state = {}
        for f in files:
            state.update(load_file(f, device=str(device)))
        missing, unexpected = model.load_state_dict(state, strict=False)
        if missing:
            raise ValueError(f"checkpoint {model_config.model} lacks parameters: {missing[:8]} ...")
    else:
        _random_init(model, model_config.seed)
    return model.eval()
