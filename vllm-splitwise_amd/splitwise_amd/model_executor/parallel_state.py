"""Tensor-parallel and stage (prompt pool / token pool) process groups.

Group layout of vllm/model_executor/parallel_utils/parallel_state.py:22-103: consecutive
blocks of ``tensor_model_parallel_size`` ranks form TP groups; with ``sep_prompt_token`` the
first half of the world is the prompt stage and the second half the token stage, and the
stage group is what step metadata is broadcast over.  Collectives are RCCL's through
``torch.distributed`` (backend "nccl" on ROCm) -- or gloo in CPU tests.
"""
from typing import Optional

import torch
import torch.distributed as dist

_TP_GROUP = None
_STAGE_GROUP = None
_TP_WORLD = 1
_TP_RANK = 0
_HOST_STAGED = False      # gloo stand-in for tests: collectives run on host copies of device tensors


def initialize_model_parallel(tensor_model_parallel_size: int = 1, sep_prompt_token: bool = False,
                              backend: Optional[str] = None, token_pools: int = 1) -> None:
    global _TP_GROUP, _STAGE_GROUP, _TP_WORLD, _TP_RANK, _HOST_STAGED
    if tensor_model_parallel_size == 1:
        # pools of one GPU need no collective at all; creating one-member groups would only add
        # rendezvous traffic (and gloo chatter on stdout) to whatever job launched the workers
        _TP_WORLD, _TP_RANK = 1, 0
        return
    assert dist.is_initialized(), "torch.distributed must be initialised for TP > 1"
    world, rank = dist.get_world_size(), dist.get_rank()
    assert world % tensor_model_parallel_size == 0
    # gloo's own handling of device tensors (private streams, pinned staging on a helper thread)
    # is not something this build wants to depend on: stage explicitly on the caller's stream
    _HOST_STAGED = (backend or dist.get_backend()) == "gloo"
    for i in range(world // tensor_model_parallel_size):
        ranks = list(range(i * tensor_model_parallel_size, (i + 1) * tensor_model_parallel_size))
        group = dist.new_group(ranks, backend=backend)
        if rank in ranks:
            _TP_GROUP = group
            _TP_RANK = ranks.index(rank)
    _TP_WORLD = tensor_model_parallel_size
    if sep_prompt_token:
        # the prompt pool, then the token pool(s): the reference has one of each, half the world each
        pool = world // (1 + token_pools)
        for first in range(0, world, pool):
            ranks = list(range(first, first + pool))
            group = dist.new_group(ranks, backend=backend)
            if rank in ranks:
                _STAGE_GROUP = group
    else:
        _STAGE_GROUP = dist.group.WORLD


def destroy_model_parallel() -> None:
    global _TP_GROUP, _STAGE_GROUP, _TP_WORLD, _TP_RANK, _HOST_STAGED
    from .custom_all_reduce import destroy_custom_ar
    destroy_custom_ar()
    _TP_GROUP = _STAGE_GROUP = None
    _TP_WORLD, _TP_RANK = 1, 0
    _HOST_STAGED = False


def get_tensor_model_parallel_world_size() -> int:
    return _TP_WORLD


def get_tensor_model_parallel_rank() -> int:
    return _TP_RANK


def get_tensor_model_parallel_group():
    return _TP_GROUP


def get_stage_parallel_group():
    return _STAGE_GROUP


def tensor_model_parallel_all_reduce(x: torch.Tensor) -> torch.Tensor:
    """In-place SUM over the TP group (communication_op.py:19-43)."""
    if _TP_WORLD == 1:
        return x
    # small messages: direct peer reads (communication_op.py:27-31 tries the same path first)
    from .custom_all_reduce import custom_all_reduce
    out = custom_all_reduce(x)
    if out is not None:
        return out
    if _HOST_STAGED and x.is_cuda:
        host = x.cpu()
        dist.all_reduce(host, group=_TP_GROUP)
        x.copy_(host)
        return x
    dist.all_reduce(x, group=_TP_GROUP)
    return x


def tensor_model_parallel_all_gather(x: torch.Tensor, dim: int = -1) -> torch.Tensor:
    """communication_op.py:46-71."""
    if _TP_WORLD == 1:
        return x
    if dim < 0:
        dim += x.dim()
    src = x.cpu() if _HOST_STAGED and x.is_cuda else x.contiguous()
    parts = [torch.empty_like(src) for _ in range(_TP_WORLD)]
    dist.all_gather(parts, src, group=_TP_GROUP)
    return torch.cat(parts, dim=dim).to(x.device)


def tensor_model_parallel_gather(x: torch.Tensor, dst: int = 0, dim: int = -1):
    """The ranks' tensors concatenated along ``dim`` on rank ``dst`` of the tensor-parallel group, None on
    the others (communication_op.py:74-105)."""
    if _TP_WORLD == 1:
        return x
    if dim < 0:
        dim += x.dim()
    if _HOST_STAGED and x.is_cuda:
        src = x.cpu()
        parts = [torch.empty_like(src) for _ in range(_TP_WORLD)]
        dist.all_gather(parts, src, group=_TP_GROUP)
        return torch.cat(parts, dim=dim).to(x.device) if _TP_RANK == dst else None
    parts = [torch.empty_like(x) for _ in range(_TP_WORLD)] if _TP_RANK == dst else None
    dist.gather(x.contiguous(), parts, dst=dist.get_global_rank(_TP_GROUP, dst), group=_TP_GROUP)
    return torch.cat(parts, dim=dim) if _TP_RANK == dst else None


def tensor_model_parallel_broadcast_from_first(x: torch.Tensor) -> torch.Tensor:
    """``x`` of the tensor-parallel group's first rank, on every rank of the group, in place (the other ranks
    pass a tensor of the same shape and dtype to receive into).  Stream-ordered: the host does not wait."""
    if _TP_WORLD == 1:
        return x
    src = dist.get_global_rank(_TP_GROUP, 0)
    if _HOST_STAGED and x.is_cuda:
        host = x.cpu()
        dist.broadcast(host, src=src, group=_TP_GROUP)
        x.copy_(host)
        return x
    dist.broadcast(x, src=src, group=_TP_GROUP)
    return x
