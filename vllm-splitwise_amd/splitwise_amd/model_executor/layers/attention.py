"""Attention layer: KV-cache write, Splitwise KV push, prompt / decode attention.

Order of operations restated from the reference's backends
(vllm/model_executor/layers/attention/backends/xformers.py:70-191, flash_attn.py:77-125):
  1. reshape_and_cache of the step's keys / values (skipped during the profiling run,
     where no cache exists yet);
  2. prompt run on a prompt-pool worker: for every hand-off slot and block run in
     ``blocks_to_nw``, push this layer's K and V blocks to the paired token worker
     (ordered after step 1 on the same stream, before the layer's attention finishes);
  3. prompt run -> causal self-attention over the prompt tokens, decode run -> paged
     attention over the block tables (v1 / v2 choice: ops/paged_attn.py:39-111).
"""
from typing import List, Optional

import torch
import torch.nn as nn

from ... import _C
from ..input_metadata import InputMetadata

_PARTITION_SIZE = 512          # must match the partition size of sw_paged_attention_v2
_SUPPORTED_HEAD_SIZES = [64, 80, 96, 112, 128, 256]


class PagedAttentionImpl:
    """Static helpers with the reference's names (ops/paged_attn.py:15-138)."""

    @staticmethod
    def get_supported_head_sizes() -> List[int]:
        return list(_SUPPORTED_HEAD_SIZES)

    @staticmethod
    def reshape_and_cache(key, value, key_cache, value_cache, input_metadata: InputMetadata) -> None:
        _C.cache_ops.reshape_and_cache(key, value, key_cache, value_cache, input_metadata.slot_mapping.flatten(),
                                       input_metadata.kv_cache_dtype)

    @staticmethod
    def query_group_width(num_heads: int, num_kv_heads: int, head_size: int, block_size: int,
                          dtype: torch.dtype, kv_cache_dtype: str) -> int:
        """Query heads one workgroup of the decode kernel serves from a single pass over K and V (the
        dispatch rule of csrc/paged_attention.cuh).  f16 / bf16, head sizes 64 / 128, blocks of 16 / 32:
        groups with a divisor of 4 or more go to the matrix-core kernel, which takes the largest divisor
        up to 16; smaller groups to the vector kernels (2 heads for an even group; 1 for fp32)."""
        if head_size not in (64, 128) or block_size < 16:
            return 1
        group = num_heads // num_kv_heads
        if dtype in (torch.float16, torch.bfloat16) and block_size in (16, 32):
            width = min(group, 16)
            while group % width:
                width -= 1
            # (bf16 over an fp8 cache has no packed vector path: the matrix-core kernel takes every group size)
            if width >= 4 or (dtype == torch.bfloat16 and kv_cache_dtype == "fp8_e5m2"):
                return width
        cap = {torch.float16: 8, torch.bfloat16: 8}.get(dtype, 1)
        width = 8 if group % 8 == 0 else 4 if group % 4 == 0 else 2 if group % 2 == 0 else 1
        return min(width, cap)

    @staticmethod
    def use_v1(num_seqs: int, num_heads: int, num_kv_heads: int, max_context_len: int, query_group: int = 1,
               mean_context_len: Optional[float] = None) -> bool:
        """The unpartitioned kernel launches one workgroup per (sequence, group of ``query_group``
        query heads); it fills the 256 CUs only with enough of them, below that the context is
        split into 512-token partitions (v2).  Thresholds measured on MI355X for MHA, GQA and fp8
        shapes (1..256 sequences, 768..16384 tokens; within 1 % of the better kernel summed over that sweep,
        at most 18 % off in any one case).  (The reference's rule, ops/paged_attn.py:62-63,
        is tuned for 128-thread blocks per query head and a shared-memory cap that does not exist
        here.)"""
        if max_context_len <= _PARTITION_SIZE:
            return True
        workgroups = num_seqs * num_heads // max(query_group, 1)
        # A ragged batch (longest context at least twice the mean, when the caller knows the mean): an
        # unpartitioned workgroup walks the longest context alone while the rest of a launch of a few hundred
        # workgroups has long finished; 512-token partitions spread it (16/2 heads, 256 sequences, mean 1009, max
        # 8192 tokens: 119 us unpartitioned, 67 us partitioned).  Launches of thousands of workgroups keep the
        # unpartitioned kernel: there the long ones overlap with the many short ones that follow.
        if (mean_context_len is not None and max_context_len >= 2 * mean_context_len
                and max_context_len > 2 * _PARTITION_SIZE and workgroups < 2048):
            return False
        if query_group >= 4:
            # the matrix-core kernel (8 waves per workgroup below 256 workgroups): unpartitioned from 128
            # workgroups, and for short contexts whatever the launch size
            return workgroups >= 128 or max_context_len <= 2 * _PARTITION_SIZE
        if workgroups >= 512:
            return True
        if workgroups >= 256:
            return max_context_len < 4096
        if workgroups >= 128:
            return max_context_len <= _PARTITION_SIZE + _PARTITION_SIZE // 2
        return workgroups >= 16 and max_context_len <= 2 * _PARTITION_SIZE

    @staticmethod
    def forward_prefix(query, key, value, key_cache, value_cache, input_metadata: InputMetadata,
                       alibi_slopes=None):
        """Prompt attention of new tokens over a prefix that is already in the cache
        (ops/paged_attn.py:113-138): query / key / value [T, heads, head_size] of the new tokens;
        the metadata carries the block tables, the first row of each sequence (start_loc), the
        total lengths (prompt_lens) and the cached lengths (context_lens)."""
        output = torch.empty_like(query)
        _C.ops.context_attention_fwd(query, key, value, output, key_cache, value_cache, input_metadata.block_tables,
                                     input_metadata.start_loc, input_metadata.prompt_lens,
                                     input_metadata.context_lens, input_metadata.max_seq_len, alibi_slopes)
        return output

    @staticmethod
    def forward_decode(query, key_cache, value_cache, input_metadata: InputMetadata, num_kv_heads: int,
                       scale: float, alibi_slopes: Optional[torch.Tensor], workspace=None) -> torch.Tensor:
        output = torch.empty_like(query)
        block_size = value_cache.shape[3]
        num_seqs, num_heads, head_size = query.shape
        max_ctx = input_metadata.max_context_len
        width = PagedAttentionImpl.query_group_width(num_heads, num_kv_heads, head_size, block_size, query.dtype,
                                                     input_metadata.kv_cache_dtype)
        forced = getattr(input_metadata, "decode_kernel", None)
        if forced is not None:
            unpartitioned = forced == "v1"
        else:
            unpartitioned = PagedAttentionImpl.use_v1(num_seqs, num_heads, num_kv_heads, max_ctx, width,
                                                      getattr(input_metadata, "mean_context_len", None))
        if unpartitioned:
            _C.ops.paged_attention_v1(output, query, key_cache, value_cache, num_kv_heads, scale,
                                      input_metadata.block_tables, input_metadata.context_lens, block_size,
                                      max_ctx, alibi_slopes, input_metadata.kv_cache_dtype)
        else:
            parts = (max_ctx + _PARTITION_SIZE - 1) // _PARTITION_SIZE
            tmp_output = torch.empty((num_seqs, num_heads, parts, head_size), dtype=output.dtype,
                                     device=output.device)
            exp_sums = torch.empty((num_seqs, num_heads, parts), dtype=torch.float32, device=output.device)
            max_logits = torch.empty_like(exp_sums)
            _C.ops.paged_attention_v2(output, exp_sums, max_logits, tmp_output, query, key_cache, value_cache,
                                      num_kv_heads, scale, input_metadata.block_tables,
                                      input_metadata.context_lens, block_size, max_ctx, alibi_slopes,
                                      input_metadata.kv_cache_dtype)
        return output


class Attention(nn.Module):
    """The class name matters: the worker finds the layers to hook the KV transport into by it
    (vllm/worker/worker.py:160-169)."""

    def __init__(self, num_heads: int, head_size: int, scale: float, num_kv_heads: Optional[int] = None,
                 alibi_slopes: Optional[List[float]] = None, sliding_window: Optional[int] = None) -> None:
        super().__init__()
        self.num_heads = num_heads
        self.head_size = head_size
        self.scale = float(scale)
        self.num_kv_heads = num_heads if num_kv_heads is None else num_kv_heads
        self.sliding_window = sliding_window
        self.alibi_slopes = (torch.tensor(alibi_slopes, dtype=torch.float32) if alibi_slopes is not None else None)
        assert self.num_heads % self.num_kv_heads == 0
        if head_size not in _SUPPORTED_HEAD_SIZES:
            raise ValueError(f"Head size {head_size} is not supported by PagedAttention. "
                             f"Supported head sizes are: {_SUPPORTED_HEAD_SIZES}.")
        self.kvcache_comm_manager = None
        self.layer_id: Optional[int] = None

    def set_kvcache_comm_manager(self, manager) -> None:
        self.kvcache_comm_manager = manager

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor,
                key_cache: Optional[torch.Tensor], value_cache: Optional[torch.Tensor],
                input_metadata: InputMetadata, rotary=None, positions: Optional[torch.Tensor] = None) -> torch.Tensor:
        """query [..., H*D], key / value [..., KVH*D] (views of the fused qkv output are fine).

        ``rotary`` / ``positions``: a model that has NOT applied its rotary embedding yet hands the
        module and the positions over; a decode step then rotates q / k and writes the cache in one
        launch, any other step applies ``rotary`` first and goes on as the reference does."""
        out_shape = query.shape
        fused_cache_write = False
        if rotary is not None:
            if (not input_metadata.is_prompt and key_cache is not None and value_cache is not None
                    and rotary.fusable_with_cache_write(query.dtype)):
                _C.ops.rotary_embedding_and_cache(positions.flatten(), query, key, value, key_cache, value_cache,
                                                  input_metadata.slot_mapping.flatten(), self.head_size,
                                                  rotary.cos_sin_cache, rotary.is_neox_style,
                                                  input_metadata.kv_cache_dtype)
                fused_cache_write = True
            else:
                query, key = rotary(positions, query, key)
        query = query.reshape(-1, self.num_heads, self.head_size) if query.dim() != 2 else \
            query.view(-1, self.num_heads, self.head_size)
        key = key.view(-1, self.num_kv_heads, self.head_size) if key.dim() == 2 else \
            key.reshape(-1, self.num_kv_heads, self.head_size)
        value = value.view(-1, self.num_kv_heads, self.head_size) if value.dim() == 2 else \
            value.reshape(-1, self.num_kv_heads, self.head_size)
        if self.alibi_slopes is not None and self.alibi_slopes.device != query.device:
            self.alibi_slopes = self.alibi_slopes.to(query.device)

        if key_cache is not None and value_cache is not None and not fused_cache_write:
            PagedAttentionImpl.reshape_and_cache(key, value, key_cache, value_cache, input_metadata)

        if input_metadata.is_prompt:
            if input_metadata.blocks_to_nw and self.kvcache_comm_manager is not None:
                comm = self.kvcache_comm_manager
                if hasattr(comm, "put_many"):
                    # every run of every hand-off slot, K and V, in one launch per layer
                    runs = [r for slot_runs in input_metadata.blocks_to_nw.values() for r in slot_runs]
                    peer = getattr(input_metadata, "nw_peer", 0)
                    if peer:
                        comm.put_many(self.layer_id, runs, peer)
                    else:
                        comm.put_many(self.layer_id, runs)
                else:
                    for sem_id, runs in input_metadata.blocks_to_nw.items():
                        for block_start, num_blocks in runs:
                            comm.put(sem_id, self.layer_id, block_start, num_blocks)
            tables = input_metadata.block_tables
            if key_cache is not None and tables is not None and tables.numel() > 0:
                # some prompt of the step continues a prefix that is in the cache already: the new tokens
                # attend to the paged context and to each other (backends/xformers.py:166-177)
                output = PagedAttentionImpl.forward_prefix(query, key, value, key_cache, value_cache, input_metadata,
                                                           self.alibi_slopes)
            else:
                output = torch.empty_like(query)
                _C.ops.prefill_attention(output, query, key, value, input_metadata.start_loc,
                                         input_metadata.prompt_lens, input_metadata.max_seq_len, self.scale,
                                         self.alibi_slopes, self.sliding_window)
        else:
            output = PagedAttentionImpl.forward_decode(query, key_cache, value_cache, input_metadata,
                                                       self.num_kv_heads, self.scale, self.alibi_slopes)
        return output.view(out_shape[:-1] + (self.num_heads * self.head_size,))
