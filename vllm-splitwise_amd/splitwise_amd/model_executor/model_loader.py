"""Builds a model of the configured geometry.

There is no network and no checkpoint in this environment, so weights are seeded random
values of the architecture's shapes (a local directory holding ``*.safetensors`` files is
loaded when present).  Initialisation runs on the target device, tensor by tensor.
"""
import glob
import os
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


@torch.no_grad()
def _random_init(model: nn.Module, seed: int) -> None:
    gen = {}
    for name, p in model.named_parameters():
        dev = p.device
        if dev not in gen:
            gen[dev] = torch.Generator(device=dev).manual_seed(seed)
        g = gen[dev]
        if name.endswith("g_idx"):
            continue                                  # keeps its sequential group index
        if not p.is_floating_point():
            # packed int4 words: uniformly random bits
            p.copy_(torch.randint(-2 ** 31, 2 ** 31 - 1, p.shape, dtype=torch.int64, device=dev,
                                  generator=g).to(p.dtype))
        elif "norm" in name and name.endswith("weight"):
            p.fill_(1.0)
        elif name.endswith("bias"):
            p.zero_()
        elif name.endswith("scales"):
            p.copy_((torch.rand(p.shape, device=dev, generator=g) * 2e-3 + 1e-4).to(p.dtype))
        else:
            # N(0, 0.02): chunked through fp32 to bound the temporary
            flat = p.view(-1)
            step = 1 << 26
            for s in range(0, flat.numel(), step):
                n = min(step, flat.numel() - s)
                flat[s:s + n].copy_((torch.randn(n, device=dev, generator=g) * 0.02).to(p.dtype))


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
