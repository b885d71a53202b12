"""OPT decoder (the plumbing configuration's model) on the HIP operator table.

Structure of vllm/model_executor/models/opt.py:47-300: learned positions with offset 2,
LayerNorm before attention and before the MLP (do_layer_norm_before, as in opt-125m), biases
on every projection, ReLU MLP, final LayerNorm, logits from the tied token embedding.
"""
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..input_metadata import InputMetadata
from ..layers.attention import Attention
from ..layers.linear import ColumnParallelLinear, LinearMethodBase, QKVParallelLinear, RowParallelLinear
from ..layers.sampler import Sampler, SamplingMetadata
from ..layers.vocab_parallel_embedding import VocabParallelEmbedding
from ..parallel_state import get_tensor_model_parallel_world_size

KVCache = Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]


class OPTAttention(nn.Module):
    def __init__(self, hidden: int, total_heads: int, dtype, linear_method) -> None:
        super().__init__()
        tp = get_tensor_model_parallel_world_size()
        self.num_heads = total_heads // tp
        self.head_dim = hidden // total_heads
        self.qkv_proj = QKVParallelLinear(hidden, self.head_dim, total_heads, None, True, dtype, linear_method)
        self.out_proj = RowParallelLinear(hidden, hidden, True, dtype, linear_method)
        self.attn = Attention(self.num_heads, self.head_dim, self.head_dim ** -0.5)

    def forward(self, hidden_states, kv_cache: KVCache, input_metadata: InputMetadata):
        q, k, v = self.qkv_proj(hidden_states).chunk(3, dim=-1)
        return self.out_proj(self.attn(q, k, v, kv_cache[0], kv_cache[1], input_metadata))


class OPTDecoderLayer(nn.Module):
    def __init__(self, cfg: dict, dtype, linear_method) -> None:
        super().__init__()
        hidden = cfg["hidden_size"]
        self.self_attn = OPTAttention(hidden, cfg["num_attention_heads"], dtype, linear_method)
        self.self_attn_layer_norm = nn.LayerNorm(hidden, dtype=dtype)
        self.fc1 = ColumnParallelLinear(hidden, cfg["intermediate_size"], True, dtype, linear_method)
        self.fc2 = RowParallelLinear(cfg["intermediate_size"], hidden, True, dtype, linear_method)
        self.final_layer_norm = nn.LayerNorm(hidden, dtype=dtype)

    def forward(self, hidden_states, kv_cache, input_metadata):
        residual = hidden_states
        hidden_states = self.self_attn(self.self_attn_layer_norm(hidden_states), kv_cache, input_metadata)
        hidden_states = residual + hidden_states
        residual = hidden_states
        hidden_states = self.fc2(F.relu(self.fc1(self.final_layer_norm(hidden_states))))
        return residual + hidden_states


class OPTForCausalLM(nn.Module):
    POSITION_OFFSET = 2     # vllm/model_executor/models/opt.py:47-57

    def __init__(self, cfg: dict, dtype: torch.dtype, linear_method: Optional[LinearMethodBase] = None) -> None:
        super().__init__()
        self.config = cfg
        hidden = cfg["hidden_size"]
        self.embed_tokens = VocabParallelEmbedding(cfg["vocab_size"], hidden, dtype)
        self.embed_positions = nn.Embedding(cfg["max_position_embeddings"] + self.POSITION_OFFSET, hidden,
                                            dtype=dtype)
        self.layers = nn.ModuleList([OPTDecoderLayer(cfg, dtype, linear_method)
                                     for _ in range(cfg["num_hidden_layers"])])
        self.final_layer_norm = nn.LayerNorm(hidden, dtype=dtype)
        self.sampler = Sampler(cfg["vocab_size"])

    def forward(self, input_ids, positions, kv_caches: List[KVCache], input_metadata: InputMetadata):
        hidden_states = self.embed_tokens(input_ids) + self.embed_positions(positions + self.POSITION_OFFSET)
        for i, layer in enumerate(self.layers):
            hidden_states = layer(hidden_states, kv_caches[i], input_metadata)
        return self.final_layer_norm(hidden_states)

    def sample(self, hidden_states, sampling_metadata: SamplingMetadata):
        return self.sampler(self.embed_tokens.weight, hidden_states, sampling_metadata)
