"""KV-cache transport between a prompt worker and its paired token worker.

Interface of vllm/worker/comm_utils.py:150-222 (``KVCacheCommManager``): ``setup_comm``,
``put(sem_id, layer_id, block_start, num_blocks)``, ``signal_and_flush(sem_id)``,
``wait(sem_id)``, ``destroy_comm``; peer of rank r is ``(r + num_prompt_workers) % world``
(:165); a put moves ``num_blocks`` blocks of the layer's K tensor and of its V tensor to the
SAME byte offset in the peer's tensors (:132-147).  All three data operations are
stream-ordered and never block the host.

Instead of MSCCL++ proxy channels (a CPU thread issuing the copies), the token worker's cache
slab is mapped into the prompt worker's address space with HIP IPC and the GPU stores into it
directly over xGMI (``transport="kernel"``: sw_kv_push / sw_kv_signal / sw_kv_wait).  A second
transport, ``"runtime"``, lets the HIP runtime move the same bytes and the flag word with
stream-ordered device-to-device copies into the mapped peer slab (the wait is the same
kernel); it exists as a fallback should direct peer stores from a kernel misbehave on a
system.  (Inter-process HIP events are not usable for the hand-off: hipIpcGetEventHandle
fails on this stack.)
"""
import os
import threading
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from .. import _C
from ..utils import MAX_SLOT_IDS

HEAD_TYPES = [0, 1]               # 0 = keys, 1 = values (comm_utils.py:17)
FLAG_BYTES = 4096                 # MAX_SLOT_IDS int32 flags + a status word, padded
_MAX_SPINS = 5_000_000            # bounded wait, about 10 s


_IPC_OPEN_TIMEOUT_S = 60.0
_SETUP_WORD = MAX_SLOT_IDS + 44   # a word of the flag area that no hand-off slot uses: the set-up check's flag
_SETUP_CHECK_TIMEOUT_S = 90.0     # the token side waits this long for the prompt side's known blocks


class TransportSetupTimeout(RuntimeError):
    """A HIP runtime call of the transport set-up did not return.  The stuck call keeps holding
    runtime locks, so the process should leave with os._exit after reporting this."""


def _bounded(fn: Callable, seconds: float, what: str):
    """Runs ``fn`` on a helper thread and gives up after ``seconds``: a set-up call that never
    returns must fail the worker loudly instead of hanging the whole job."""
    box: Dict[str, object] = {}
    device = torch.cuda.current_device()

    def run() -> None:
        try:
            torch.cuda.set_device(device)
            box["value"] = fn()
        except BaseException as e:              # noqa: BLE001 - re-raised on the caller's thread
            box["error"] = e

    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(seconds)
    if t.is_alive():
        raise TransportSetupTimeout(f"{what} did not return within {seconds:.0f} s")
    if "error" in box:
        raise box["error"]
    return box["value"]


class _DevicePtrView:
    """Exposes raw device memory to torch through __cuda_array_interface__."""

    def __init__(self, ptr: int, nbytes: int) -> None:
        self.__cuda_array_interface__ = {"data": (ptr, False), "shape": (nbytes,), "typestr": "|u1",
                                         "strides": None, "version": 2}


_FOUR_GIB = 1 << 32
_IPC_MARGIN = 64 << 20            # covers the caching allocator's rounding of large requests


def _ipc_safe_size(nbytes: int) -> int:
    """hipIpcOpenMemHandle on ROCm 7.2 never returns for an allocation whose size modulo 4 GiB
    is 2 GiB or more (measured: 1.99, 5.9, 17, 41.9 GiB open in a millisecond; 2.01, 3.9, 6, 19,
    23, 42.1 GiB hang).  Sizes in that band are padded up to the next multiple of 4 GiB."""
    if nbytes % _FOUR_GIB > (_FOUR_GIB // 2) - _IPC_MARGIN:
        nbytes = -(-nbytes // _FOUR_GIB) * _FOUR_GIB
    return nbytes


def slab_layout(num_layers: int, key_block_bytes: int, num_blocks: int,
                pad_for_ipc: bool = True) -> Tuple[List[Tuple[int, int]], int, int]:
    """Byte offsets of every layer's (K, V) tensor inside one cache slab, the offset of the
    flag area and the slab size.  Both workers of a pair compute the same layout.  ``pad_for_ipc``
    (workers with a peer) moves the size out of the band a peer process cannot map; a co-located
    engine has nobody to export the slab to and takes exactly what the block budget allows."""
    offsets = []
    cursor = 0
    per_tensor = key_block_bytes * num_blocks
    per_tensor_aligned = -(-per_tensor // 256) * 256
    for _ in range(num_layers):
        k_off = cursor
        v_off = cursor + per_tensor_aligned
        offsets.append((k_off, v_off))
        cursor += 2 * per_tensor_aligned
    total = cursor + FLAG_BYTES
    if pad_for_ipc:
        total = _ipc_safe_size(total)
    return offsets, total - FLAG_BYTES, total


class PeerCache:
    """The paired worker's cache slab as this process sees it."""

    def __init__(self, base: int, layer_offsets: List[Tuple[int, int]], flags_offset: int, num_blocks: int,
                 remote: bool) -> None:
        self.base = base                          # the IPC mapping (0 on the token side, which never pushes)
        self.layer_offsets = layer_offsets        # byte offsets of every layer's (K, V) tensor in the PEER's slab
        self.flags_offset = flags_offset
        self.num_blocks = num_blocks
        self.remote = remote                      # on another device (a link in between) or on this one
        self.bytes: Optional[torch.Tensor] = None  # runtime transport: the mapping as a tensor
        self.signal_epoch = [0] * MAX_SLOT_IDS


class KVCacheCommunicator:
    """The data path once both sides are connected (comm_utils.py:90-147).  A prompt worker may feed
    several token workers (``peers``; the reference has exactly one, :165)."""

    def __init__(self, block_size: int, layer_offsets: List[Tuple[int, int]], local_base: int,
                 peers: List[PeerCache], flags_offset: int, transport: str, device: torch.device) -> None:
        self.block_size = block_size                  # bytes of one block of one tensor
        self.layer_offsets = layer_offsets
        self.local_base = local_base
        self.peers = peers
        self.flags_offset = flags_offset
        self.transport = transport
        self.device = device
        self.wait_epoch = [0] * MAX_SLOT_IDS
        self.bytes_pushed = 0
        # where a wait that ran out of its spin bound leaves 1 + slot: a word of pinned host memory that the
        # wait kernel stores to directly, so the worker can look at it after every step's own
        # synchronisation (the copy of the sampled tokens) without another device round trip
        self.status_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        # polls before a wait gives up (read once, at set-up; tests shorten it)
        self.max_spins = int(os.environ.get("SPLITWISE_KV_WAIT_SPINS", _MAX_SPINS))
        # Pushes and signals run on a stream of their own: a layer's push waits (on the device) for that
        # layer's cache write and then moves bytes while the compute stream goes on with the layer's
        # attention and the next layers; the signal is enqueued behind the step's last push.  Over one
        # xGMI link a 772-token prompt is milliseconds of link time: on the compute stream it would
        # stall every layer behind link-bound stores.
        self.push_stream = torch.cuda.Stream(device) if device.type == "cuda" else None
        # workgroups of one push launch: a link is filled by a few dozen, a device-local copy (both
        # workers on one GPU: tests, rehearsals) wants one per 64 KiB chunk
        env = os.environ.get("SPLITWISE_KV_PUSH_WORKGROUPS")
        self.push_workgroups = [int(env) if env is not None else (32 if p.remote else 0) for p in peers]
        self.push_calibration: Dict[int, Dict[int, float]] = {}      # peer -> {workgroups: GB/s}, if measured
        # runtime transport state
        self.local_bytes: Optional[torch.Tensor] = None

    # the single-peer names of earlier callers
    @property
    def peer_base(self) -> int:
        return self.peers[0].base

    @property
    def signal_epoch(self) -> List[int]:
        return self.peers[0].signal_epoch

    def peer_num_blocks(self, peer: int = 0) -> int:
        return self.peers[peer].num_blocks

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _after_compute(self) -> "torch.cuda.Stream":
        """The push stream, made to wait for everything enqueued on the compute stream so far (the
        cache write of the layer whose blocks are about to be pushed)."""
        event = torch.cuda.Event()
        event.record(torch.cuda.current_stream(self.device))
        self.push_stream.wait_event(event)
        return self.push_stream

    def synchronize(self) -> None:
        """Host wait for every push and signal issued so far."""
        if self.push_stream is not None:
            self.push_stream.synchronize()

    def pushes_done_event(self) -> Optional["torch.cuda.Event"]:
        """An event behind everything queued on the push stream so far.  A prompt step has not finished
        before it has completed: until then stores into the token worker's blocks are still in flight,
        and a block that went back to the pool (a sequence that ended at its first token, an abort)
        could be handed to a decoding sequence and then overwritten by the late push."""
        if self.push_stream is None:
            return None
        event = torch.cuda.Event()
        event.record(self.push_stream)
        return event

    def put_many(self, layer_id: int, runs: Sequence[Tuple[int, ...]], peer: int = 0) -> None:
        """All runs of one layer, K and V, in one launch.  A run is (block_start, num_blocks) -- the same
        block numbers on both sides, as in the reference -- or (src_block_start, dst_block_start, num_blocks)
        when the prompt side numbers its blocks by itself."""
        if not runs:
            return
        target = self.peers[peer]
        k_src, v_src = self.layer_offsets[layer_id]
        k_dst, v_dst = target.layer_offsets[layer_id]
        src_offsets, dst_offsets, sizes = [], [], []
        for run in runs:
            src_start, dst_start, num_blocks = (run[0], run[0], run[1]) if len(run) == 2 else run
            if (min(src_start, dst_start, num_blocks) < 0 or dst_start + num_blocks > target.num_blocks):
                raise RuntimeError(f"KV push of blocks {src_start}..+{num_blocks} -> {dst_start} is outside the peer's "
                                   f"cache of {target.num_blocks} blocks")
            for src_base, dst_base in ((k_src, k_dst), (v_src, v_dst)):
                src_offsets.append(src_base + src_start * self.block_size)
                dst_offsets.append(dst_base + dst_start * self.block_size)
                sizes.append(num_blocks * self.block_size)
        self.bytes_pushed += sum(sizes)
        stream = self._after_compute()
        if self.transport == "kernel":
            _C.kv_ops.kv_push_mapped(target.base, self.local_base, dst_offsets, src_offsets, sizes,
                                     stream.cuda_stream, self.push_workgroups[peer])
        else:
            with torch.cuda.stream(stream):
                for dst, src, n in zip(dst_offsets, src_offsets, sizes):
                    target.bytes[dst:dst + n].copy_(self.local_bytes[src:src + n], non_blocking=True)

    def calibrate_push_workgroups(self, peer: int = 0, candidates: Sequence[int] = (16, 32, 64, 128, 256),
                                  max_blocks: int = 256, repeats: int = 3) -> int:
        """Picks the workgroup cap of a push launch towards ``peer`` by measurement: the first blocks of layer 0
        (K and V) are pushed with every candidate cap, and the smallest cap within 8 % of the best time wins -- the
        workgroups a push does not need stay with the prompt step's own kernels.  How many a link needs is a
        property of the node (link width, store latency), so it is measured where the worker runs, at set-up, when
        neither cache holds anything yet.  Returns the cap; ``push_calibration[peer]`` keeps the GB/s per cap."""
        target = self.peers[peer]
        k_off, v_off = self.layer_offsets[0]
        local_blocks = (v_off - k_off) // self.block_size
        n = min(max_blocks, local_blocks, target.num_blocks)
        if n <= 0 or self.transport != "kernel":
            return self.push_workgroups[peer]
        pushed_before = self.bytes_pushed
        nbytes = 2 * n * self.block_size
        table: Dict[int, float] = {}
        try:
            for cap in candidates:
                self.push_workgroups[peer] = cap
                self.put_many(0, [(0, n)], peer)                  # untimed: first touch of the mapping
                start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                start.record(self.push_stream)
                for _ in range(repeats):
                    self.put_many(0, [(0, n)], peer)
                end.record(self.push_stream)
                end.synchronize()
                table[cap] = nbytes * repeats / (start.elapsed_time(end) * 1e-3) / 1e9
        finally:
            self.bytes_pushed = pushed_before
        best = max(table.values())
        chosen = min(cap for cap, gbs in table.items() if gbs >= best / 1.08)
        self.push_workgroups[peer] = chosen
        self.push_calibration[peer] = {cap: round(gbs, 1) for cap, gbs in table.items()}
        return chosen

    def put(self, sem_id: int, layer_id: int, block_start: int, num_blocks: int) -> None:
        del sem_id                                    # ordering is by stream, not by channel
        self.put_many(layer_id, [(block_start, num_blocks)])

    def signal_and_flush(self, sem_id: int, epoch: Optional[int] = None, peer: int = 0) -> None:
        target = self.peers[peer]
        if epoch is None:
            epoch = target.signal_epoch[sem_id] + 1
        target.signal_epoch[sem_id] = epoch
        # on the push stream, behind every push of the step (each of which waited for its layer's cache
        # write): the hand-off is announced when the last layer's blocks have left, not when the
        # prompt step's remaining layers, logits and sampling are over
        if self.transport == "kernel":
            _C.kv_ops.kv_signal(target.base + target.flags_offset, sem_id, epoch, self.push_stream.cuda_stream)
        else:
            word = torch.tensor([epoch], dtype=torch.int32, device=self.device).view(torch.uint8)
            off = target.flags_offset + 4 * sem_id
            with torch.cuda.stream(self._after_compute()):               # `word` was made on the compute stream
                target.bytes[off:off + 4].copy_(word, non_blocking=True)      # after the data copies, in stream order
            word.record_stream(self.push_stream)

    def store_flag_word(self, peer: int, word: int, value: int) -> None:
        """``value`` into word ``word`` of the peer's flag area, behind everything queued on the push stream (the
        mechanism of signal_and_flush without its epoch bookkeeping)."""
        target = self.peers[peer]
        if self.transport == "kernel":
            _C.kv_ops.kv_signal(target.base + target.flags_offset, word, value, self.push_stream.cuda_stream)
        else:
            src = torch.tensor([value], dtype=torch.int32, device=self.device).view(torch.uint8)
            off = target.flags_offset + 4 * word
            with torch.cuda.stream(self._after_compute()):
                target.bytes[off:off + 4].copy_(src, non_blocking=True)
            src.record_stream(self.push_stream)

    def wait_flag_word(self, word: int, value: int, seconds: float) -> bool:
        """Host-synchronising: True once word ``word`` of the local flag area has reached ``value``, False after
        ``seconds`` (bounded waits of the usual length, one after the other)."""
        import time
        flags = self.local_base + self.flags_offset
        deadline = time.monotonic() + seconds
        while True:
            _C.kv_ops.kv_wait(flags, word, value, self.max_spins, self.status_host.data_ptr(), self._stream())
            torch.cuda.synchronize(self.device)
            if int(self.status_host[0]) == 0:
                return True
            self.status_host[0] = 0
            if time.monotonic() > deadline:
                return False

    def wait(self, sem_id: int, epoch: Optional[int] = None) -> None:
        if epoch is None:
            epoch = self.wait_epoch[sem_id] + 1
        self.wait_epoch[sem_id] = epoch
        flags = self.local_base + self.flags_offset
        _C.kv_ops.kv_wait(flags, sem_id, epoch, self.max_spins, self.status_host.data_ptr(), self._stream())

    def check_status(self) -> None:
        """Raises if a wait of a step that has FINISHED on the device ran out of its spin bound (reads host
        memory only: call it after the step's own synchronisation)."""
        status = int(self.status_host[0])
        if status != 0:
            self.status_host[0] = 0
            raise RuntimeError(f"KV hand-off wait timed out on slot {status - 1}: the prompt worker's signal "
                               "never arrived, the decode step ran on KV blocks that were not delivered")


class KVCacheCommManager:
    """``init`` carries what the reference's ``mscclpp_init_method`` stands for: how to reach the
    peer.  Keys: ``exchange`` -- callable(obj) -> the paired worker's obj (a symmetric swap over
    any control channel; a prompt worker with several token pools is called as exchange(obj, peer_index));
    ``transport`` -- "kernel" (default) or "runtime".

    World layout: ranks 0..n-1 are the prompt pool (n = num_prompt_workers), followed by one or more
    token pools of n ranks each.  Prompt rank r is paired with rank r of EVERY token pool; with one token
    pool that is the reference's (rank + n) % world (comm_utils.py:165)."""

    def __init__(self, rank: int, world_size: int, num_prompt_workers: int, init: Dict) -> None:
        self.rank = rank
        self.world_size = world_size
        self.num_prompt_workers = num_prompt_workers
        self.is_prompt_side = rank < num_prompt_workers
        n = num_prompt_workers
        if self.is_prompt_side:
            self.peer_ranks = list(range(rank + n, world_size, n))
        else:
            self.peer_ranks = [rank % n]
        self.corr_worker_rank = self.peer_ranks[0]
        self.exchange: Callable = init["exchange"]
        self.transport: str = init.get("transport", "kernel")
        if self.transport not in ("kernel", "runtime"):
            raise ValueError(f"unknown KV transport {self.transport!r}")
        self.kvcache_comm: Optional[KVCacheCommunicator] = None
        self._peer_mappings: List[int] = []
        self._slab: Optional[torch.Tensor] = None

    def _swap(self, obj, peer_index: int):
        if len(self.peer_ranks) == 1:
            return self.exchange(obj)
        return self.exchange(obj, peer_index)

    def setup_comm(self, num_layers: int, kv_cache: List[Tuple[torch.Tensor, torch.Tensor]]) -> None:
        """Registers every layer's K and V tensor (they must be views of one slab, see
        ``CacheEngine``) and connects to the peer(s)."""
        k0 = kv_cache[0][0]
        slab = getattr(k0, "_slab", None)
        if slab is None:
            raise RuntimeError("KV transport needs the cache allocated as one slab (CacheEngine(slab=True))")
        self._slab = slab
        # an IPC handle names a whole device allocation; the slab may sit inside a larger one (a
        # segment of torch's caching allocator), so the allocation is exported and the slab's offset
        # in it travels beside the handle
        alloc_base, alloc_bytes = _C.kv_ops.mem_address_range(slab.data_ptr())
        if _ipc_safe_size(alloc_bytes) != alloc_bytes:
            raise RuntimeError(f"the cache slab lives in a device allocation of {alloc_bytes} bytes, which a peer "
                               "process cannot map on this runtime; allocate it through slab_layout()")
        device = slab.device
        block_bytes = k0[0].numel() * k0.element_size()
        base = slab.data_ptr()
        layer_offsets = []
        for layer_id in range(num_layers):
            k, v = kv_cache[layer_id]
            layer_offsets.append((k.data_ptr() - base, v.data_ptr() - base))
        flags_offset = slab.numel() - FLAG_BYTES
        slab[flags_offset:].zero_()
        torch.cuda.synchronize(device)
        info = {"rank": self.rank, "nbytes": slab.numel(), "layer_offsets": layer_offsets,
                "block_bytes": block_bytes, "num_blocks": int(k0.shape[0]), "flags_offset": flags_offset,
                "handle": _C.kv_ops.ipc_get_mem_handle(alloc_base), "offset": base - alloc_base,
                "pid_device": (torch.cuda.current_device(),)}
        peers: List[PeerCache] = []
        for index, peer_rank in enumerate(self.peer_ranks):
            peer = self._swap(info, index)
            if peer["rank"] != peer_rank:
                raise RuntimeError(f"KV transport paired rank {self.rank} with {peer['rank']}, expected {peer_rank}")
            if peer["block_bytes"] != block_bytes or len(peer["layer_offsets"]) != num_layers:
                raise RuntimeError("prompt and token workers disagree on the cache layout (bytes per block, layers)")
            peer_base = 0
            if self.is_prompt_side:
                mapped = _bounded(lambda h=peer["handle"]: _C.kv_ops.ipc_open_mem_handle(h), _IPC_OPEN_TIMEOUT_S,
                                  "mapping the token worker's KV cache (hipIpcOpenMemHandle)")
                self._peer_mappings.append(mapped)
                peer_base = mapped + peer["offset"]
            remote = tuple(peer.get("pid_device", ())) != (torch.cuda.current_device(),)
            view = PeerCache(peer_base, [tuple(o) for o in peer["layer_offsets"]], peer["flags_offset"],
                             peer["num_blocks"], remote)
            if self.transport == "runtime" and self.is_prompt_side:
                view.bytes = torch.as_tensor(_DevicePtrView(peer_base, peer["nbytes"]), device=device)
            peers.append(view)
        comm = KVCacheCommunicator(block_bytes, layer_offsets, base, peers, flags_offset, self.transport, device)
        if self.transport == "runtime":
            comm.local_bytes = slab
        self.kvcache_comm = comm
        if self.is_prompt_side and self.transport == "kernel" and "SPLITWISE_KV_PUSH_WORKGROUPS" not in os.environ:
            # over a link the cap is measured (a device-local peer takes one workgroup per chunk, uncapped);
            # SPLITWISE_KV_PUSH_CALIBRATE=1 measures it for local peers too, =0 keeps the default of 32
            mode = os.environ.get("SPLITWISE_KV_PUSH_CALIBRATE", "")
            for index, view in enumerate(peers):
                if mode != "0" and (view.remote or mode == "1"):
                    try:
                        comm.calibrate_push_workgroups(index)
                    except RuntimeError as exc:                          # keep the default, say so
                        print(f"[kv transport] push calibration towards peer {index} failed ({exc}); "
                              f"keeping {comm.push_workgroups[index]} workgroups", flush=True)
        verified = self._known_answer_check(comm, slab, layer_offsets[0][0], block_bytes, int(k0.shape[0]), peers)
        for index in range(len(self.peer_ranks)):
            # both sides mapped before any put; the token side's answer says whether the check's blocks arrived
            answer = self._swap({"rank": self.rank, "ready": True, "verified": verified}, index)
            if self.is_prompt_side and isinstance(answer, dict) and answer.get("verified") is False:
                raise RuntimeError(f"KV transport ({self.transport}): the blocks of the set-up check did not arrive "
                                   f"intact in the cache of rank {answer.get('rank')}")

    def _known_answer_check(self, comm: "KVCacheCommunicator", slab: torch.Tensor, k_offset: int, block_bytes: int,
                            num_blocks: int, peers: List[PeerCache]) -> Optional[bool]:
        """Before the first real hand-off, a few blocks with known bytes travel the real path -- push (or copy),
        flag word, wait -- and the token side compares what its cache then holds: a transport that maps but does
        not deliver (or delivers something else) fails at set-up, loudly, not as wrong tokens later.  The prompt
        side returns None; the token side its verdict, which travels back with the "ready" message."""
        n = min(4, num_blocks, min(p.num_blocks for p in peers))
        nbytes = n * block_bytes
        pattern = ((torch.arange(nbytes, device=slab.device, dtype=torch.int64) * 131 + 7) % 251).to(torch.uint8)
        if self.is_prompt_side:
            pushed = comm.bytes_pushed
            slab[k_offset:k_offset + nbytes].copy_(pattern)
            for index in range(len(peers)):
                comm.put_many(0, [(0, 0, n)], index)
                comm.store_flag_word(index, _SETUP_WORD, 1)
            comm.synchronize()
            slab[k_offset:k_offset + nbytes].zero_()
            torch.cuda.synchronize(slab.device)
            comm.bytes_pushed = pushed
            return None
        arrived = comm.wait_flag_word(_SETUP_WORD, 1, _SETUP_CHECK_TIMEOUT_S)
        intact = arrived and bool(torch.equal(slab[k_offset:k_offset + nbytes], pattern))
        if not intact:
            print(f"[kv transport] rank {self.rank}: set-up check failed ({'bytes differ' if arrived else 'no flag'})",
                  flush=True)
        return intact

    def destroy_comm(self) -> None:
        if self.kvcache_comm is not None:
            torch.cuda.synchronize(self.kvcache_comm.device)
        for mapped in self._peer_mappings:
            _C.kv_ops.ipc_close_mem_handle(mapped)
        self._peer_mappings = []
        self.kvcache_comm = None

    def check_status(self) -> None:
        """Raises if a wait ran out of its spin bound (host-synchronising: call off the hot path)."""
        comm = self.kvcache_comm
        if comm is None or self._slab is None:
            return
        torch.cuda.synchronize(comm.device)            # every stream of the device, the push stream included
        comm.check_status()

    def check_status_after_step(self) -> None:
        """The same check without a synchronisation of its own, for the end of a step whose samples
        have already reached the host."""
        if self.kvcache_comm is not None:
            self.kvcache_comm.check_status()

    def wait(self, sem_id: int, epoch: Optional[int] = None) -> None:
        self.kvcache_comm.wait(sem_id, epoch)

    def signal_and_flush(self, sem_id: int, epoch: Optional[int] = None, peer: int = 0) -> None:
        self.kvcache_comm.signal_and_flush(sem_id, epoch, peer)

    def put(self, sem_id: int, layer_id: int, block_start: int, num_blocks: int) -> None:
        self.kvcache_comm.put(sem_id, layer_id, block_start, num_blocks)

    def put_many(self, layer_id: int, runs: Sequence[Tuple[int, ...]], peer: int = 0) -> None:
        self.kvcache_comm.put_many(layer_id, runs, peer)

    def pushes_done_event(self):
        return self.kvcache_comm.pushes_done_event() if self.kvcache_comm is not None else None
