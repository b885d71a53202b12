"""All-reduce over the tensor-parallel ranks of one node by direct peer reads.

Interface of vllm/model_executor/parallel_utils/custom_all_reduce.py:27-234 (``init_custom_ar``,
``custom_all_reduce``, ``get_handle``, ``CustomAllreduce`` with ``should_custom_ar``,
``all_reduce_unreg``, ``all_reduce_reg``, ``close``).  Every rank allocates one device buffer
(signal area + scratch + staging), the ranks swap IPC handles over the group's host channel
(``all_gather_object``), and the kernel reads the peers' staging areas over xGMI.  The reference
restricts itself to NVLink-connected CUDA devices and is compiled out on ROCm
(csrc/pybind.cpp:113-127); here the only requirement is that the peers' memory can be mapped.

Graph capture: a launch carries no per-call host state (the barrier values come from counters on the
device), so ``all_reduce_unreg`` may be captured into a HIP graph and replayed -- the staging copy
becomes a copy node in front of the kernel.  Where the reference registers the addresses of a
captured graph's buffers with its peers afterwards (custom_all_reduce.py:188-213), nothing is
needed here: peers only ever read the staging area, whose address never changes.
"""
from typing import List, Optional

import torch
import torch.distributed as dist

from .. import _C

_SUPPORTED_WORLD_SIZES = (2, 4, 6, 8)
_MAX_SIZE = 8192 * 1024          # largest message in bytes (custom_all_reduce.py:148-149)
_HANDLE: Optional["CustomAllreduce"] = None


def _align(n: int, a: int) -> int:
    return -(-n // a) * a


class CustomAllreduce:
    def __init__(self, rank: int, world_size: int, group=None, max_size: int = _MAX_SIZE,
                 device: Optional[torch.device] = None) -> None:
        if world_size not in _SUPPORTED_WORLD_SIZES:
            raise ValueError(f"custom all-reduce supports {_SUPPORTED_WORLD_SIZES} ranks, not {world_size}")
        self.rank, self.world_size, self.group, self.max_size = rank, world_size, group, max_size
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.signal_off = 0
        self.scratch_off = _align(_C.custom_ar.signal_bytes(), 256)
        self.staging_off = self.scratch_off + _align(max_size, 256)
        total = self.staging_off + _align(max_size, 256)
        from ..worker.kv_comm import _IPC_OPEN_TIMEOUT_S, _bounded, _ipc_safe_size
        # A device allocation of its own, never a piece of a segment that the caching allocator has kept
        # from earlier tensors: a handle names the whole allocation, and a peer's hipIpcOpenMemHandle
        # never returns for allocations of some sizes (see _ipc_safe_size)
        if self.device.type == "cuda":
            self._pool = torch.cuda.MemPool()
            with torch.cuda.use_mem_pool(self._pool, device=self.device):
                self.buffer = torch.zeros(total, dtype=torch.uint8, device=self.device)    # zeroed signal area
        else:
            self.buffer = torch.zeros(total, dtype=torch.uint8, device=self.device)
        torch.cuda.synchronize(self.device)
        # the buffer's offset inside its allocation travels beside the handle
        alloc_base, alloc_bytes = _C.kv_ops.mem_address_range(self.buffer.data_ptr())
        if _ipc_safe_size(alloc_bytes) != alloc_bytes:
            raise RuntimeError(f"the all-reduce buffer lives in a device allocation of {alloc_bytes} bytes, "
                               "which a peer process cannot map on this runtime")
        mine = (_C.kv_ops.ipc_get_mem_handle(alloc_base), self.buffer.data_ptr() - alloc_base)
        everyone: List = [None] * world_size
        dist.all_gather_object(everyone, mine, group=group)
        self._mappings: List[int] = []
        bases: List[int] = []
        for r, (handle, offset) in enumerate(everyone):
            if r == rank:
                bases.append(self.buffer.data_ptr())
            else:
                mapped = _bounded(lambda h=handle: _C.kv_ops.ipc_open_mem_handle(h), _IPC_OPEN_TIMEOUT_S,
                                  f"mapping rank {r}'s all-reduce buffer (hipIpcOpenMemHandle)")
                self._mappings.append(mapped)
                bases.append(mapped + offset)
        self._ptr = _C.custom_ar.init(bases, rank, self.signal_off, self.scratch_off, self.staging_off, max_size)
        # nobody may announce itself before every signal area has been zeroed and mapped
        dist.barrier(group=group)

    def should_custom_ar(self, inp: torch.Tensor) -> bool:
        """csrc/custom_all_reduce.cu:58-68 for fully connected peers."""
        nbytes = inp.numel() * inp.element_size()
        return (inp.is_cuda and inp.is_contiguous() and nbytes > 0 and nbytes % 16 == 0 and nbytes <= self.max_size
                and inp.dtype in (torch.float16, torch.bfloat16, torch.float32))

    def all_reduce_unreg(self, inp: torch.Tensor, out: Optional[torch.Tensor] = None, algorithm: int = 0):
        """``inp`` is copied to the staging area on the stream, then reduced (custom_all_reduce.py:222-226)."""
        if out is None:
            out = torch.empty_like(inp)
        _C.custom_ar.all_reduce(self._ptr, inp, out, algorithm)
        return out

    all_reduce = all_reduce_unreg

    def staging(self, like: torch.Tensor) -> torch.Tensor:
        """A view of this rank's staging area with ``like``'s shape and dtype: a producer that writes
        its result here saves the copy (the reference's registered-buffer case)."""
        nbytes = like.numel() * like.element_size()
        return self.buffer[self.staging_off:self.staging_off + nbytes].view(like.dtype).view(like.shape)

    def all_reduce_reg(self, inp: torch.Tensor, out: Optional[torch.Tensor] = None, algorithm: int = 0):
        """``inp`` must be ``self.staging(...)`` (custom_all_reduce.py:215-220)."""
        if inp.data_ptr() != self.buffer.data_ptr() + self.staging_off:
            raise RuntimeError("buffer address is not registered!")
        return self.all_reduce_unreg(inp, out, algorithm)

    def _status_word(self) -> torch.Tensor:
        off = self.signal_off + _C.custom_ar.signal_bytes() - 256
        return self.buffer[off:off + 4].view(torch.int32)

    def check_status(self) -> None:
        """Raises if a barrier ran out of its spin bound (host-synchronising: off the hot path)."""
        status = int(self._status_word().item())
        if status != 0:
            raise RuntimeError(f"all-reduce barrier timed out waiting for rank {status - 1}")

    def check_status_after_step(self) -> None:
        """The same check for the end of every step, without making the host wait for the device: the status
        word travels to pinned host memory with a stream-ordered copy, and a call looks at what the copy
        queued by an EARLIER call has brought (a timed-out barrier is reported one step late; a rank that
        has run ahead of its device, as tensor-parallel followers do, keeps its lead)."""
        if getattr(self, "_status_host", None) is None:
            self._status_host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self._status_event = None
        if self._status_event is not None:
            if not self._status_event.query():
                return                               # the earlier copy is still queued behind device work
            status = int(self._status_host[0])
            if status != 0:
                raise RuntimeError(f"all-reduce barrier timed out waiting for rank {status - 1}")
        self._status_host.copy_(self._status_word(), non_blocking=True)
        self._status_event = torch.cuda.Event()
        self._status_event.record(torch.cuda.current_stream(self.device))

    def close(self) -> None:
        if getattr(self, "_ptr", 0):
            torch.cuda.synchronize(self.device)
            _C.custom_ar.dispose(self._ptr)
            self._ptr = 0
            for mapped in self._mappings:
                _C.kv_ops.ipc_close_mem_handle(mapped)
            self._mappings = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def init_custom_ar(rank: int, world_size: int, group=None) -> None:
    """custom_all_reduce.py:27-50: no-op for one rank or an unsupported group size."""
    global _HANDLE
    if _HANDLE is not None or world_size == 1:
        return
    if world_size not in _SUPPORTED_WORLD_SIZES:
        return
    handle = CustomAllreduce(rank, world_size, group)
    if _self_check(handle, rank, world_size, group):
        _HANDLE = handle
    else:
        handle.close()
        print(f"[all-reduce] rank {rank}: the peer-read all-reduce did not reproduce a known sum on this node; "
              "small all-reduces go through the collective library instead", flush=True)


def _self_check(handle: "CustomAllreduce", rank: int, world_size: int, group) -> bool:
    """One all-reduce with a known answer, in both forms, before the path is trusted: peer reads over the links of
    this node have to work, not only map.  The ranks then agree on the verdict (a MIN over their flags), so the
    path is used by all of them or by none."""
    ok = True
    try:
        for numel, algorithm in ((4096, 1), (512 * 1024, 2)):
            base = torch.arange(numel, device="cuda") % 7
            out = handle.all_reduce_unreg((base + rank).to(torch.bfloat16), algorithm=algorithm)
            want = (base * world_size + world_size * (world_size - 1) // 2).to(torch.bfloat16)     # exact in bf16
            ok = ok and bool(torch.equal(out, want))
        torch.cuda.synchronize()
        handle.check_status()
    except Exception as exc:                        # noqa: BLE001 - any failure means "do not use this path"
        print(f"[all-reduce] rank {rank}: self-check raised {exc!r}", flush=True)
        ok = False
    host_group = dist.get_backend(group) == "gloo"
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if host_group else "cuda")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(int(flag.item()))


def get_handle() -> Optional[CustomAllreduce]:
    return _HANDLE


def is_initialized() -> bool:
    return _HANDLE is not None


def destroy_custom_ar() -> None:
    global _HANDLE
    if _HANDLE is not None:
        _HANDLE.close()
        _HANDLE = None


def custom_all_reduce(inp: torch.Tensor) -> Optional[torch.Tensor]:
    """The reduced tensor, or None when this path does not take the message (custom_all_reduce.py:87-108)."""
    handle = _HANDLE
    if handle is None or not handle.should_custom_ar(inp):
        return None
    return handle.all_reduce_unreg(inp)
