"""Llama decoder on the HIP operator table.

Structure of vllm/model_executor/models/llama.py:53-331: per layer fused-residual RMSNorm ->
fused qkv projection -> NeoX rotary on q, k -> Attention -> o_proj (row-parallel + all-reduce)
-> RMSNorm -> merged gate/up projection -> SiluAndMul -> down_proj; final RMSNorm; logits from
the (untied) lm_head.  q, k, v reach the attention operator as strided views of the qkv output.
"""
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from ..input_metadata import InputMetadata
from ..layers.attention import Attention
from ..layers.basic import RMSNorm, RotaryEmbedding, SiluAndMul
from ..layers.linear import (LinearMethodBase, MergedColumnParallelLinear, QKVParallelLinear, RowParallelLinear)
from ..layers.sampler import Sampler, SamplingMetadata
from ..layers.vocab_parallel_embedding import ParallelLMHead, VocabParallelEmbedding
from ..parallel_state import get_tensor_model_parallel_world_size

KVCache = Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]


class LlamaMLP(nn.Module):
    def __init__(self, hidden: int, intermediate: int, dtype, linear_method) -> None:
        super().__init__()
        self.gate_up_proj = MergedColumnParallelLinear(hidden, [intermediate] * 2, False, dtype, linear_method)
        self.down_proj = RowParallelLinear(intermediate, hidden, False, dtype, linear_method)
        self.act_fn = SiluAndMul()

    def forward(self, x):
        return self.down_proj(self.act_fn(self.gate_up_proj(x)))


class LlamaAttention(nn.Module):
    def __init__(self, cfg: dict, dtype, linear_method) -> None:
        super().__init__()
        tp = get_tensor_model_parallel_world_size()
        hidden = cfg["hidden_size"]
        total_heads = cfg["num_attention_heads"]
        total_kv = cfg.get("num_key_value_heads", total_heads)
        self.num_heads = total_heads // tp
        self.num_kv_heads = max(1, total_kv // tp)
        self.head_dim = hidden // total_heads
        self.q_size = self.num_heads * self.head_dim
        self.kv_size = self.num_kv_heads * self.head_dim
        self.qkv_proj = QKVParallelLinear(hidden, self.head_dim, total_heads, total_kv, False, dtype, linear_method)
        self.o_proj = RowParallelLinear(total_heads * self.head_dim, hidden, False, dtype, linear_method)
        self.rotary_emb = RotaryEmbedding(self.head_dim, self.head_dim, cfg.get("max_position_embeddings", 8192),
                                          cfg.get("rope_theta", 10000.0), True, dtype)
        self.attn = Attention(self.num_heads, self.head_dim, self.head_dim ** -0.5, self.num_kv_heads,
                              sliding_window=cfg.get("sliding_window"))

    def forward(self, positions, hidden_states, kv_cache: KVCache, input_metadata: InputMetadata):
        qkv = self.qkv_proj(hidden_states)
        q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
        # decode steps rotate q / k and write the cache in one launch; other steps rotate first
        attn = self.attn(q, k, v, kv_cache[0], kv_cache[1], input_metadata, rotary=self.rotary_emb, positions=positions)
        return self.o_proj(attn)


class LlamaDecoderLayer(nn.Module):
    def __init__(self, cfg: dict, dtype, linear_method) -> None:
        super().__init__()
        self.self_attn = LlamaAttention(cfg, dtype, linear_method)
        self.mlp = LlamaMLP(cfg["hidden_size"], cfg["intermediate_size"], dtype, linear_method)
        eps = cfg.get("rms_norm_eps", 1e-6)
        self.input_layernorm = RMSNorm(cfg["hidden_size"], eps)
        self.post_attention_layernorm = RMSNorm(cfg["hidden_size"], eps)

    def forward(self, positions, hidden_states, kv_cache, input_metadata, residual):
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states, kv_cache, input_metadata)
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        hidden_states = self.mlp(hidden_states)
        return hidden_states, residual


class LlamaForCausalLM(nn.Module):
    def __init__(self, cfg: dict, dtype: torch.dtype, linear_method: Optional[LinearMethodBase] = None) -> None:
        super().__init__()
        self.config = cfg
        self.embed_tokens = VocabParallelEmbedding(cfg["vocab_size"], cfg["hidden_size"], dtype)
        self.layers = nn.ModuleList([LlamaDecoderLayer(cfg, dtype, linear_method)
                                     for _ in range(cfg["num_hidden_layers"])])
        self.norm = RMSNorm(cfg["hidden_size"], cfg.get("rms_norm_eps", 1e-6))
        self.lm_head = ParallelLMHead(cfg["vocab_size"], cfg["hidden_size"], dtype)
        self.sampler = Sampler(cfg["vocab_size"])

    def forward(self, input_ids, positions, kv_caches: List[KVCache], input_metadata: InputMetadata):
        hidden_states = self.embed_tokens(input_ids)
        residual = None
        for i, layer in enumerate(self.layers):
            hidden_states, residual = layer(positions, hidden_states, kv_caches[i], input_metadata, residual)
        hidden_states, _ = self.norm(hidden_states, residual)
        return hidden_states

    def sample(self, hidden_states, sampling_metadata: SamplingMetadata):
        return self.sampler(self.lm_head.weight, hidden_states, sampling_metadata)
