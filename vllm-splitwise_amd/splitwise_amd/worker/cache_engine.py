"""Owns the paged KV cache of one worker.

Same layout and accounting as vllm/worker/cache_engine.py:15-178: per layer a key cache
[num_blocks, kv_heads, head_size/x, block_size, x] (x = 16 bytes of cache elements) and a
value cache [num_blocks, kv_heads, head_size, block_size]; pinned host mirrors for swapping;
swap / copy run on a side stream with one event per layer.  This layout is also the wire
format of the KV push between the prompt and token pools, so it is kept bit for bit.
"""
from typing import Dict, List, Optional, Tuple

import torch

from .. import _C
from ..config import CacheConfig, ModelConfig, ParallelConfig
from ..utils import STR_DTYPE_TO_TORCH_DTYPE

KVCache = Tuple[torch.Tensor, torch.Tensor]


def _cache_torch_dtype(cache_dtype: str, model_dtype: torch.dtype) -> torch.dtype:
    if cache_dtype == "auto":
        return model_dtype
    if cache_dtype == "fp8_e5m2":
        return torch.uint8
    return STR_DTYPE_TO_TORCH_DTYPE[cache_dtype]


class CacheEngine:
    def __init__(self, cache_config: CacheConfig, model_config: ModelConfig, parallel_config: ParallelConfig,
                 device: torch.device) -> None:
        self.cache_config = cache_config
        self.head_size = model_config.get_head_size()
        self.num_layers = model_config.get_num_layers(parallel_config)
        self.num_heads = model_config.get_num_kv_heads(parallel_config)
        self.block_size = cache_config.block_size
        self.num_gpu_blocks = cache_config.num_gpu_blocks
        self.num_cpu_blocks = cache_config.num_cpu_blocks
        self.dtype = _cache_torch_dtype(cache_config.cache_dtype, model_config.dtype)
        self.device = device
        self.gpu_slab: Optional[torch.Tensor] = None
        # only a worker of a prompt/token pair exports its slab to another process
        self.exported_to_peer = bool(getattr(parallel_config, "sep_prompt_token", False))
        self.gpu_cache = self._allocate_gpu_slab(self.num_gpu_blocks, device)
        self.cpu_cache = self._allocate(self.num_cpu_blocks, torch.device("cpu"), pinned=device.type == "cuda")
        self.cache_stream = torch.cuda.Stream(device) if device.type == "cuda" else None
        self.events = ([torch.cuda.Event() for _ in range(self.num_layers)] if device.type == "cuda" else [])

    def get_key_block_shape(self) -> Tuple[int, int, int, int]:
        x = 16 // torch.tensor([], dtype=self.dtype).element_size()
        return (self.num_heads, self.head_size // x, self.block_size, x)

    def get_value_block_shape(self) -> Tuple[int, int, int]:
        return (self.num_heads, self.head_size, self.block_size)

    def _allocate_gpu_slab(self, num_blocks: int, device: torch.device) -> List[KVCache]:
        """Every layer's K and V tensor is a view of ONE device allocation (plus a small flag area
        at its end): a single IPC handle then exports the whole cache to the paired worker, and a
        byte offset means the same block on both sides of a prompt/token pair."""
        from .kv_comm import slab_layout
        es = torch.tensor([], dtype=self.dtype).element_size()
        kshape, vshape = self.get_key_block_shape(), self.get_value_block_shape()
        block_elems = self.num_heads * self.head_size * self.block_size
        offsets, _, total = slab_layout(self.num_layers, block_elems * es, num_blocks,
                                        pad_for_ipc=self.exported_to_peer)
        # A device allocation of its own, never a piece of a segment the caching allocator has kept
        # from earlier tensors: the paired worker maps the allocation the slab lives in, and only
        # the size slab_layout() chose is known to be mappable.
        if device.type == "cuda":
            self._slab_pool = torch.cuda.MemPool()
            with torch.cuda.use_mem_pool(self._slab_pool, device=device):
                slab = torch.empty(total, dtype=torch.uint8, device=device)
        else:
            slab = torch.empty(total, dtype=torch.uint8, device=device)
        self.gpu_slab = slab
        n = num_blocks * block_elems * es
        cache: List[KVCache] = []
        for k_off, v_off in offsets:
            k = slab[k_off:k_off + n].view(self.dtype).view(num_blocks, *kshape)
            v = slab[v_off:v_off + n].view(self.dtype).view(num_blocks, *vshape)
            k._slab = slab
            cache.append((k, v))
        return cache

    def _allocate(self, num_blocks: int, device: torch.device, pinned: bool) -> List[KVCache]:
        cache: List[KVCache] = []
        for _ in range(self.num_layers):
            k = torch.empty((num_blocks, *self.get_key_block_shape()), dtype=self.dtype, device=device,
                            pin_memory=pinned)
            v = torch.empty((num_blocks, *self.get_value_block_shape()), dtype=self.dtype, device=device,
                            pin_memory=pinned)
            cache.append((k, v))
        return cache

    def _swap(self, src: List[KVCache], dst: List[KVCache], src_to_dst: Dict[int, int]) -> None:
        # the blocks may still be written by kernels queued on the compute stream
        self.cache_stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self.cache_stream):
            for layer in range(self.num_layers):
                _C.cache_ops.swap_blocks(src[layer][0], dst[layer][0], src_to_dst)
                _C.cache_ops.swap_blocks(src[layer][1], dst[layer][1], src_to_dst)
                self.events[layer].record(stream=self.cache_stream)

    def swap_in(self, src_to_dst: Dict[int, int]) -> None:
        self._swap(self.cpu_cache, self.gpu_cache, src_to_dst)

    def swap_out(self, src_to_dst: Dict[int, int]) -> None:
        self._swap(self.gpu_cache, self.cpu_cache, src_to_dst)

    def copy(self, src_to_dsts: Dict[int, List[int]]) -> None:
        # one launch for all layers; runs on the compute stream, so it is ordered before the step
        _C.cache_ops.copy_blocks([k for k, _ in self.gpu_cache], [v for _, v in self.gpu_cache], src_to_dsts)

    @staticmethod
    def get_cache_block_size(block_size: int, cache_dtype: str, model_config: ModelConfig,
                             parallel_config: ParallelConfig) -> int:
        """Bytes of one block over all layers, K and V (vllm/worker/cache_engine.py:155-174)."""
        head_size = model_config.get_head_size()
        num_heads = model_config.get_num_kv_heads(parallel_config)
        num_layers = model_config.get_num_layers(parallel_config)
        per_tensor = block_size * num_heads * head_size
        dtype = _cache_torch_dtype(cache_dtype, model_config.dtype)
        return torch.tensor([], dtype=dtype).element_size() * num_layers * 2 * per_tensor
