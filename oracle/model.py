"""Whole-model CPU restatement (fp32, no KV cache): logits of every position of a token
sequence for the Llama and OPT architectures, from a state dict of the engine's model.

Follows vllm/model_executor/models/llama.py:53-331 and opts.py:47-300 (layer order, NeoX
rotary, RMSNorm with fused residual, SiLU-gated MLP; OPT: learned positions with offset 2,
pre-LayerNorm, ReLU MLP, tied embedding).  Used to check greedy decoding: the token the
engine picked must be (within a tolerance) the arg-max of these logits.
"""
from typing import Dict, List

import torch
import torch.nn.functional as F


def _rms(x, w, eps):
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w


def _causal_attention(q, k, v, num_heads, num_kv_heads):
    n, _ = q.shape
    d = q.shape[1] // num_heads
    q = q.view(n, num_heads, d).transpose(0, 1)
    k = k.view(n, num_kv_heads, d).transpose(0, 1).repeat_interleave(num_heads // num_kv_heads, dim=0)
    v = v.view(n, num_kv_heads, d).transpose(0, 1).repeat_interleave(num_heads // num_kv_heads, dim=0)
    w = (q @ k.transpose(1, 2)) * d ** -0.5
    w = w.masked_fill(torch.triu(torch.ones(n, n, dtype=torch.bool), 1), float("-inf"))
    return (torch.softmax(w, -1) @ v).transpose(0, 1).reshape(n, num_heads * d)


def _rope_neox(x, num_heads, positions, theta):
    n = x.shape[0]
    d = x.shape[1] // num_heads
    inv = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float) / d))
    f = positions.float()[:, None] * inv[None, :]
    cos, sin = f.cos()[:, None, :], f.sin()[:, None, :]
    x = x.view(n, num_heads, d)
    x1, x2 = x[..., :d // 2], x[..., d // 2:]
    return torch.cat((x1 * cos - x2 * sin, x2 * cos + x1 * sin), -1).reshape(n, num_heads * d)


def _through_e5m2(t: torch.Tensor, act_dtype: torch.dtype) -> torch.Tensor:
    """What a decode step reads back from an fp8-e5m2 cache: the 16-bit activation rounded once
    more to e5m2 (csrc/cache_kernels.cu:201-214 writes, attention_kernels.cu:222-232 reads)."""
    from . import fp8
    return fp8.from_e5m2(fp8.to_e5m2(t.to(act_dtype)), torch.float32)


def llama_logits(sd: Dict[str, torch.Tensor], cfg: dict, tokens: List[int], prompt_len: int = None,
                 kv_cache_dtype: str = "auto", act_dtype: torch.dtype = torch.float16) -> torch.Tensor:
    """``kv_cache_dtype="fp8_e5m2"``: positions from ``prompt_len`` on are decode steps and attend
    to keys and values as the cache holds them (e5m2); prompt positions attend to the exact
    ones, as the prompt step computes attention before anything is read back from the cache."""
    sd = {k: v.float() for k, v in sd.items()}
    fp8_cache = kv_cache_dtype == "fp8_e5m2"
    h, kvh = cfg["num_attention_heads"], cfg.get("num_key_value_heads", cfg["num_attention_heads"])
    d = cfg["hidden_size"] // h
    eps = cfg.get("rms_norm_eps", 1e-6)
    pos = torch.arange(len(tokens))
    x = sd["embed_tokens.weight"][torch.tensor(tokens)]
    for i in range(cfg["num_hidden_layers"]):
        p = f"layers.{i}."
        y = _rms(x, sd[p + "input_layernorm.weight"], eps)
        qkv = y @ sd[p + "self_attn.qkv_proj.weight"].t()
        q, k, v = qkv.split([h * d, kvh * d, kvh * d], -1)
        q = _rope_neox(q, h, pos, cfg.get("rope_theta", 10000.0))
        k = _rope_neox(k, kvh, pos, cfg.get("rope_theta", 10000.0))
        a = _causal_attention(q, k, v, h, kvh)
        if fp8_cache:
            a_dec = _causal_attention(q, _through_e5m2(k, act_dtype), _through_e5m2(v, act_dtype), h, kvh)
            a = torch.cat((a[:prompt_len], a_dec[prompt_len:]))
        x = x + a @ sd[p + "self_attn.o_proj.weight"].t()
        y = _rms(x, sd[p + "post_attention_layernorm.weight"], eps)
        gu = y @ sd[p + "mlp.gate_up_proj.weight"].t()
        g, u = gu.chunk(2, -1)
        x = x + (F.silu(g) * u) @ sd[p + "mlp.down_proj.weight"].t()
    x = _rms(x, sd["norm.weight"], eps)
    return x @ sd["lm_head.weight"].t()


def opt_logits(sd: Dict[str, torch.Tensor], cfg: dict, tokens: List[int]) -> torch.Tensor:
    sd = {k: v.float() for k, v in sd.items()}
    h = cfg["num_attention_heads"]
    hid = cfg["hidden_size"]
    pos = torch.arange(len(tokens)) + 2
    x = sd["embed_tokens.weight"][torch.tensor(tokens)] + sd["embed_positions.weight"][pos]
    for i in range(cfg["num_hidden_layers"]):
        p = f"layers.{i}."
        y = F.layer_norm(x, (hid,), sd[p + "self_attn_layer_norm.weight"], sd[p + "self_attn_layer_norm.bias"])
        qkv = y @ sd[p + "self_attn.qkv_proj.weight"].t() + sd[p + "self_attn.qkv_proj.bias"]
        q, k, v = qkv.chunk(3, -1)
        a = _causal_attention(q, k, v, h, h)
        x = x + a @ sd[p + "self_attn.out_proj.weight"].t() + sd[p + "self_attn.out_proj.bias"]
        y = F.layer_norm(x, (hid,), sd[p + "final_layer_norm.weight"], sd[p + "final_layer_norm.bias"])
        y = F.relu(y @ sd[p + "fc1.weight"].t() + sd[p + "fc1.bias"])
        x = x + y @ sd[p + "fc2.weight"].t() + sd[p + "fc2.bias"]
    x = F.layer_norm(x, (hid,), sd["final_layer_norm.weight"], sd["final_layer_norm.bias"])
    return x @ sd["embed_tokens.weight"].t()


def logits(architecture: str, sd, cfg, tokens, **kv):
    return llama_logits(sd, cfg, tokens, **kv) if architecture == "llama" else opt_logits(sd, cfg, tokens)
