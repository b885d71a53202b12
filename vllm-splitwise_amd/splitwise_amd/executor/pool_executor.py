"""Executor for pools of tensor-parallel workers, co-located or split.

World layout of vllm/config.py:433-443 and vllm/executor/ray_gpu_executor.py:87-232: without
``sep_prompt_token`` one MIXED pool of ``tensor_parallel_size`` ranks; with it a prompt pool
(ranks 0..tp-1) followed by a token pool (ranks tp..2tp-1), rank i pushing its shard of the KV
cache to rank i+tp.  One process per GPU: rank 0 lives in the engine process, every other rank is
a child process running the worker command loop; TP collectives are RCCL's (torch.distributed
"nccl") inside each pool.

Differences from the reference, all deliberate:
  * no Ray and no per-step tensor broadcast: every rank of a pool receives the same small step
    command (sequence metadata, block operations) over its control channel and prepares its own
    input tensors, so there is no driver-to-follower tensor traffic at all; only the first rank
    of a pool reports the sampled tokens;
  * with split pools, ``execute_split`` runs a prompt step and a decode step at the same time;
  * pipelined steps (``launch_prompt`` / ``decode_send`` / ``decode_collect``, and ``execute_model(defer=True)``
    for a co-located pool): a pool always has its next step queued while the engine works on the last one's
    tokens -- the reference waits for every rank's reply before it schedules again
    (ray_gpu_executor.py:284-324, its TODO at :292-293).  The sampled ids reach the pool's other ranks device
    to device (ModelRunner._share_sampled_ids);
  * several token pools behind one prompt pool (``parallel_config.num_token_pools``), and a prompt pool
    whose cache holds only the prompts in flight (``staged_prompt_cache``), as in SplitwiseExecutor.
"""
import copy
import os
import time
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch

from ..config import CacheConfig, DeviceConfig, ModelConfig, ParallelConfig, SchedulerConfig
from ..sequence import SamplerOutput, SequenceGroupMetadata, slim_decode_metadata, unpack_sampler_output
from ..worker.worker import Worker
from .channel import PipeChannel, pickled
from .executor_base import ExecutorBase
from .gpu_executor import check_block_budget


def _control_timeout() -> float:
    """Longest wait for one reply of a worker process (seconds).  Generous: the first reply comes after
    the model has been built and the memory profiled."""
    return float(os.environ.get("SPLITWISE_CONTROL_TIMEOUT_S", "900"))


def _listening_store(world_size: int):
    """The rendezvous store of the workers' process group, served from THIS process on a port that
    the kernel picks at bind time and that stays bound from then on.  (Looking for a free port first
    and binding it later leaves a window in which any outgoing connection of any process can take the
    number: rank 0 then fails to bind while the other ranks wait for a store that never comes up.)"""
    import datetime

    import torch.distributed as dist
    store = dist.TCPStore("127.0.0.1", 0, world_size, is_master=True, wait_for_workers=False,
                          timeout=datetime.timedelta(seconds=_control_timeout()))
    return store, store.port


class _PoolPending:
    """A pipelined prompt step of a tensor-parallel pool: the first rank's samples, finished when the other
    ranks have acknowledged the step (which they do once their KV pushes have left) as well."""

    def __init__(self, pending, executor: "PoolExecutor", followers: List[int]) -> None:
        self.pending = pending
        self.executor = executor
        self.followers = followers
        # a follower's acknowledgements arrive in the order of its commands: this step's is the one that brings
        # its count of read ones up to the number sent so far (a later step may be queued behind this one)
        self.need = {r: executor._acks_sent.get(r, 0) for r in followers}

    def done(self) -> bool:
        if not self.pending.done():
            return False
        for r in self.followers:
            self.executor._drain_acks(r)
        return all(self.executor._acks_read.get(r, 0) >= self.need[r] for r in self.followers)

    def wait(self) -> None:
        self.pending.wait()
        for r in self.followers:
            self.executor._await_acks(r, self.need[r])

    def resolve(self):
        self.wait()
        return self.pending.resolve()


class PoolExecutor(ExecutorBase):
    def __init__(self, model_config: ModelConfig, cache_config: CacheConfig, parallel_config: ParallelConfig,
                 scheduler_config: SchedulerConfig, device_config: DeviceConfig,
                 num_gpu_blocks_override: Optional[int] = None, devices: Optional[Sequence[int]] = None,
                 transport: str = "kernel", worker_cls=Worker, channels: Optional[Dict[int, Any]] = None) -> None:
        self.model_config = model_config
        self.cache_config = cache_config
        self.parallel_config = parallel_config
        self.scheduler_config = scheduler_config
        self.transport = transport
        self.worker_cls = worker_cls
        self.tp = parallel_config.tensor_parallel_size
        self.split = parallel_config.sep_prompt_token
        self.world = parallel_config.world_size
        if self.tp > 1 and not model_config.enforce_eager and not self._collectives_capturable():
            model_config.enforce_eager = True
        if devices is None:
            n = max(torch.cuda.device_count(), 1)
            devices = [r % n for r in range(self.world)]
        self.devices = list(devices)
        init_method, store = None, None
        if self.tp > 1:
            store, port = _listening_store(self.world)
            init_method = f"tcp://127.0.0.1:{port}"
        self.processes = []
        self.channels: Dict[int, Any] = dict(channels or {})
        if not self.channels and self.world > 1:
            import torch.multiprocessing as mp
            from .remote_worker import spawn_entry
            ctx = mp.get_context("spawn")
            for rank in range(1, self.world):
                parent, child = ctx.Pipe(duplex=True)
                proc = ctx.Process(target=spawn_entry, args=(child, self._remote_cfg(rank, init_method)), daemon=True)
                proc.start()
                child.close()
                self.processes.append(proc)
                self.channels[rank] = PipeChannel(parent)
        transport_init = {"exchange": self._exchange, "transport": transport} if self.split else None
        extra = {"distributed_init_method": init_method, "distributed_store": store} if init_method else {}
        self.driver_worker = worker_cls(model_config, parallel_config, scheduler_config, device_config,
                                        local_rank=self.devices[0], rank=0, kv_cache_dtype=cache_config.cache_dtype,
                                        is_driver_worker=True, kv_transport_init=transport_init, **extra)
        self.all_ranks = list(range(self.world))
        self.prompt_pool = list(range(self.tp))
        self.num_lanes = max(getattr(parallel_config, "num_token_pools", 1), 1) if self.split else 1
        self.staged = bool(getattr(parallel_config, "staged_prompt_cache", False))
        self.token_pools = ([list(range(self.tp * (1 + j), self.tp * (2 + j))) for j in range(self.num_lanes)]
                            if self.split else [self.prompt_pool])
        self.token_pool = self.token_pools[0]
        self.supports_pipelined_split = self.split
        self.supports_deferred_samples = not self.split
        self._acks_sent: Dict[int, int] = {}       # per follower: commands sent whose acknowledgement nobody waits for
        self._acks_read: Dict[int, int] = {}       # ... and how many of those acknowledgements have been read
        try:
            self._call_all(self.all_ranks, "init_model")
            self._call_all(self.all_ranks, "load_model")
            self._init_cache(num_gpu_blocks_override)
        except BaseException:
            # the other ranks may be inside a collective or a rendezvous that will never complete now:
            # do not talk to them, end them
            self.shutdown(graceful=False)
            raise

    def _collectives_capturable(self) -> bool:
        """Decode graphs of tensor-parallel ranks hold the layer's all-reduces: that works when every one
        of them goes through the peer-read kernel (its launches carry no per-call host state), i.e. the
        group size is one it supports, it is not switched off, and the largest captured batch's
        [rows, hidden] message fits its buffer.  The reference captures its collectives through cupy's
        NCCL binding or the registered-buffer form of its custom kernel (model_runner.py:701-760,
        custom_all_reduce.py:188-234)."""
        from ..model_executor.custom_all_reduce import _MAX_SIZE, _SUPPORTED_WORLD_SIZES
        if self.tp not in _SUPPORTED_WORLD_SIZES or getattr(self.parallel_config, "disable_custom_all_reduce", False):
            return False
        hidden = self.model_config.hf_config["hidden_size"]
        elt = torch.tensor([], dtype=self.model_config.dtype).element_size()
        rows = min(256, max(self.scheduler_config.max_num_seqs, 1))
        rows = 1 << (rows - 1).bit_length() if rows <= 8 else -(-rows // 8) * 8      # padded graph batch
        return rows * hidden * elt <= _MAX_SIZE

    def _remote_cfg(self, rank: int, init_method: Optional[str]) -> Dict[str, Any]:
        return {"model_config": self.model_config, "parallel_config": self.parallel_config,
                "scheduler_config": self.scheduler_config, "cache_config": self.cache_config, "rank": rank,
                "local_rank": self.devices[rank], "transport": self.transport,
                "distributed_init_method": init_method, "reports_output": rank % self.tp == 0,
                "worker_cls": None if self.worker_cls is Worker else self.worker_cls}

    # ------------------------------------------------------------ control plane
    @staticmethod
    def _message(name: str, kwargs: dict) -> tuple:
        if name in ("execute_model", "execute_deferred"):
            # decode commands travel without the sequences' token ids
            slim = slim_decode_metadata(kwargs["seq_group_metadata_list"])
            tag = "execute" if name == "execute_model" else "execute_deferred"
            return (tag, dict(kwargs, seq_group_metadata_list=slim))
        return ("call", name, kwargs)

    def _send(self, rank: int, name: str, kwargs: dict) -> None:
        self.channels[rank].send(self._message(name, kwargs))

    def _send_many(self, ranks: Sequence[int], name: str, kwargs: dict, before=None, after=None) -> None:
        """The same command to several ranks, built and serialised once: with 256 sequences in flight a decode
        command is ~100 KB of block tables, and seven copies of the slimming and pickling were a third of the
        host's time per step of a TP4 + TP4 engine.  ``before(rank)`` / ``after(rank)`` run around each send."""
        if not ranks:
            return
        message = self._message(name, kwargs)
        payload = None
        for r in ranks:
            if before is not None:
                before(r)
            ch = self.channels[r]
            if hasattr(ch, "send_pickled"):
                if payload is None:
                    payload = pickled(message)
                ch.send_pickled(payload)
            else:
                ch.send(message)
            if after is not None:
                after(r)

    def _recv(self, rank: int, deadline: float):
        """One message from ``rank``, looking once a second whether its process is still alive (a worker
        that died inside a HIP call does not always get to close its end of the pipe at once)."""
        ch = self.channels[rank]
        proc = self.processes[rank - 1] if 0 < rank <= len(self.processes) else None
        while True:
            try:
                return ch.recv(timeout=1.0)
            except TimeoutError:
                pass
            except (EOFError, OSError) as exc:
                raise RuntimeError(f"worker {rank} went away (exit code "
                                   f"{proc.exitcode if proc is not None else 'unknown'})") from exc
            if proc is not None and not proc.is_alive() and not ch.poll(0.0):
                raise RuntimeError(f"worker {rank} died (exit code {proc.exitcode})")
            if time.monotonic() > deadline:
                raise RuntimeError(f"worker {rank} did not answer within {_control_timeout():g} s "
                                   "(SPLITWISE_CONTROL_TIMEOUT_S)")

    def _wait(self, rank: int):
        status, result = self._recv(rank, time.monotonic() + _control_timeout())
        if status == "samples":
            return unpack_sampler_output(result)
        if status != "ok":
            raise RuntimeError(f"worker {rank} failed: {result}")
        return result

    def _call_all(self, ranks: List[int], name: str, **kwargs) -> Dict[int, Any]:
        """Starts ``name`` on every remote rank, runs it on the local one, waits for all."""
        remote = [r for r in ranks if r != 0]
        self._send_many(remote, name, kwargs)
        results: Dict[int, Any] = {}
        local_error = None
        if 0 in ranks:
            try:
                results[0] = getattr(self.driver_worker, name)(**kwargs)
            except Exception as exc:                # noqa: BLE001
                local_error = exc
        if local_error is not None:
            # collect what the others have to say, briefly: their error is usually the cause of ours
            remote_errors = []
            for r in remote:
                try:
                    status, result = self._recv(r, time.monotonic() + 5.0)
                    if status == "error":
                        remote_errors.append(f"worker {r}: {result}")
                except RuntimeError as exc:
                    remote_errors.append(str(exc))
            if remote_errors:
                raise RuntimeError(f"{name} failed on rank 0 ({local_error}); " + "; ".join(remote_errors)) from local_error
            raise local_error
        for r in remote:
            results[r] = self._wait(r)
        return results

    def _exchange(self, obj, lane: int = 0):
        """One round of the KV-transport handshake with token pool ``lane``: every prompt rank and every rank of
        that pool hands in one object and gets its pair's (prompt rank i <-> rank i of the pool).  Runs inside
        the local prompt worker's set-up, which is when the others ask too; the prompt ranks go through the
        pools in the same order as this one."""
        deadline = time.monotonic() + _control_timeout()
        token = self.token_pools[lane]
        waiting = set(self.prompt_pool[1:]) | set(token)
        handed_in: Dict[int, Any] = {}
        while waiting:                                      # in whatever order they arrive
            progressed = False
            for r in sorted(waiting):
                if not self.channels[r].poll(0.0):
                    continue
                tag, payload = self._recv(r, deadline)
                if tag == "error":                          # the worker's set-up failed before it got here
                    raise RuntimeError(f"worker {r} failed during the KV-transport handshake: {payload}")
                if tag != "kv_exchange":
                    raise RuntimeError(f"worker {r} sent {tag!r} during the KV-transport handshake")
                handed_in[r] = payload
                waiting.discard(r)
                progressed = True
            if not progressed and waiting:
                r = min(waiting)                            # block (bounded, liveness-checked) on one of them
                tag, payload = self._recv(r, deadline)
                if tag != "kv_exchange":
                    raise RuntimeError(f"worker {r} failed during the KV-transport handshake: {tag!r} {payload}")
                handed_in[r] = payload
                waiting.discard(r)
        for i in range(1, self.tp):                         # pairs that live entirely in other processes
            self.channels[i].send(("kv_exchange", handed_in[token[i]]))
            self.channels[token[i]].send(("kv_exchange", handed_in[i]))
        self.channels[token[0]].send(("kv_exchange", obj))  # the local worker's own pair
        return handed_in[token[0]]

    # -------------------------------------------------------------------- cache
    def _call_pools(self, name: str, prompt_kwargs: dict, token_kwargs: dict) -> None:
        """``name`` on every rank, the prompt pool and the token pools with arguments of their own."""
        if not self.split or prompt_kwargs is token_kwargs:
            self._call_all(self.all_ranks, name, **token_kwargs)
            return
        token_ranks = [r for pool in self.token_pools for r in pool]
        self._send_many(token_ranks, name, token_kwargs)
        self._call_all(self.prompt_pool, name, **prompt_kwargs)
        for r in token_ranks:
            self._wait(r)

    def _init_cache(self, override: Optional[int]) -> None:
        from .splitwise_executor import staged_prompt_blocks
        cc = self.cache_config
        staging = staged_prompt_blocks(self.scheduler_config, cc.block_size) if self.staged else None
        if override is None:
            sizes = self._call_all(self.all_ranks, "profile_num_available_blocks", block_size=cc.block_size,
                                   gpu_memory_utilization=cc.gpu_memory_utilization,
                                   cpu_swap_space=cc.swap_space_bytes, cache_dtype=cc.cache_dtype)
            # a block id means the same block on every rank that shares a block space: all of them get the
            # smallest size (a staged prompt pool has a block space of its own)
            counted = [s for r, s in sizes.items() if not (self.staged and r in self.prompt_pool)]
            num_gpu_blocks = min(s[0] for s in counted)
            num_cpu_blocks = min(s[1] for s in counted)
            if self.staged and min(sizes[r][0] for r in self.prompt_pool) < staging:
                raise ValueError(f"a prompt GPU has room for {min(sizes[r][0] for r in self.prompt_pool)} cache "
                                 f"blocks, a prompt step needs {staging}")
        else:
            per_block = self.driver_worker.get_cache_block_size_bytes(cc.block_size, cc.cache_dtype)
            num_gpu_blocks, num_cpu_blocks = override, int(cc.swap_space_bytes // per_block)
        check_block_budget(self.model_config.max_model_len, cc.block_size, num_gpu_blocks)
        cc.num_gpu_blocks, cc.num_cpu_blocks = num_gpu_blocks, num_cpu_blocks
        prompt_cc = cc
        if self.staged:
            prompt_cc = copy.copy(cc)
            prompt_cc.num_gpu_blocks, prompt_cc.num_cpu_blocks = staging, 0
        self.prompt_cache_blocks = prompt_cc.num_gpu_blocks
        self._call_pools("init_cache_engine", dict(cache_config=prompt_cc), dict(cache_config=cc))
        self._call_all(self.all_ranks, "warm_up_model")
        if self.split:
            self._call_all(self.all_ranks, "setup_kvcache_comm")

    # --------------------------------------------------------------------- steps
    @staticmethod
    def _kwargs(metas, swap_in, swap_out, copy, nw, epochs) -> dict:
        return dict(seq_group_metadata_list=metas, blocks_to_swap_in=swap_in, blocks_to_swap_out=swap_out,
                    blocks_to_copy=copy, blocks_to_nw=nw or {}, nw_epochs=epochs)

    def execute_model(self, seq_group_metadata_list: List[SequenceGroupMetadata],
                      blocks_to_swap_in: Dict[int, int], blocks_to_swap_out: Dict[int, int],
                      blocks_to_copy: Dict[int, List[int]],
                      blocks_to_nw: Optional[Dict[int, List[Tuple[int, int]]]] = None,
                      nw_epochs: Optional[Dict[int, int]] = None, defer: bool = False) -> SamplerOutput:
        if not seq_group_metadata_list:
            return SamplerOutput([])
        kwargs = self._kwargs(seq_group_metadata_list, blocks_to_swap_in, blocks_to_swap_out, blocks_to_copy,
                              blocks_to_nw, nw_epochs)
        if defer:
            return self.execute_deferred(kwargs)
        for r in self.prompt_pool[1:]:
            self._await_acks(r)
        pool = self.prompt_pool if seq_group_metadata_list[0].is_prompt else self.token_pool
        return self._call_all(pool, "execute_model", **kwargs)[pool[0]]

    def execute_split(self, prompt_kwargs: Optional[dict], decode_kwargs: Optional[dict]):
        """A prompt step on the prompt pool and a decode step on the token pool, concurrently."""
        assert self.split
        if decode_kwargs is not None:
            self._send_many(self.token_pool, "execute_model", decode_kwargs)
        prompt_out = None
        if prompt_kwargs is not None:
            prompt_out = self._call_all(self.prompt_pool, "execute_model", **prompt_kwargs)[0]
        decode_out = None
        if decode_kwargs is not None:
            outs = {r: self._wait(r) for r in self.token_pool}
            decode_out = outs[self.token_pool[0]]
        return prompt_out, decode_out

    # ------------------------------------------------------------ pipelined steps
    def _drain_acks(self, rank: int) -> None:
        """Reads the acknowledgements a rank has sent for commands whose result nobody waits for (a follower's
        deferred steps).  An error among them is raised here, one command late."""
        ch = self.channels[rank]
        while self._acks_read.get(rank, 0) < self._acks_sent.get(rank, 0) and ch.poll(0.0):
            self._read_ack(rank)

    def _read_ack(self, rank: int) -> None:
        status, result = self._recv(rank, time.monotonic() + _control_timeout())
        self._acks_read[rank] = self._acks_read.get(rank, 0) + 1
        if status == "error":
            raise RuntimeError(f"worker {rank} failed: {result}")

    def _owes_ack(self, rank: int) -> None:
        self._acks_sent[rank] = self._acks_sent.get(rank, 0) + 1

    def _await_acks(self, rank: int, upto: Optional[int] = None) -> None:
        """Blocks until ``upto`` acknowledgements of the rank have been read (all that are owed by default)."""
        target = self._acks_sent.get(rank, 0) if upto is None else upto
        while self._acks_read.get(rank, 0) < target:
            self._read_ack(rank)

    def execute_deferred(self, kwargs: dict):
        """A co-located pool's step with the samples left on the device: -> the first rank's PendingSamples.
        The other ranks acknowledge when they have queued the step; nobody waits for that."""
        self._send_many(self.prompt_pool[1:], "execute_deferred", kwargs, before=self._drain_acks, after=self._owes_ack)
        return self.driver_worker.execute_model(**kwargs, defer=True)

    def launch_prompt(self, prompt_kwargs: dict, lane: int = 0):
        """Starts a prompt step on the prompt pool whose sequences will decode on token pool ``lane``;
        -> an object with done() / wait() / resolve() like the first rank's PendingSamples.  The step counts as
        finished when every rank of the pool has pushed its shard of the KV cache."""
        followers = self.prompt_pool[1:]
        kwargs = dict(prompt_kwargs, nw_peer=lane) if lane else prompt_kwargs
        # (a follower replies once its pushes have left: Worker.execute_model)
        self._send_many(followers, "execute_model", kwargs, before=self._drain_acks, after=self._owes_ack)
        pending = self.driver_worker.execute_model(**kwargs, defer=True)
        return _PoolPending(pending, self, followers)

    def decode_send(self, decode_kwargs: Optional[dict], lane: int = 0) -> None:
        if decode_kwargs is not None:
            self._send_many(self.token_pools[lane], "execute_deferred", decode_kwargs)
            return
        for r in self.token_pools[lane]:
            self.channels[r].send(("flush",))

    def decode_collect(self, lane: int = 0) -> Optional[SamplerOutput]:
        outs = [self._wait(r) for r in self.token_pools[lane]]
        return outs[0]

    def profile_token_pool(self, name: str, lane: int = 0, **kwargs):
        """Runs a measurement method of ``Worker`` on the first rank of the token pool (rank 0 when the pool is
        co-located)."""
        rank = self.token_pool[0]
        if rank == 0:
            return getattr(self.driver_worker, name)(**kwargs)
        self._send(rank, name, kwargs)
        return self._wait(rank)

    def kv_bytes_pushed(self) -> int:
        comm = self.driver_worker.kvcache_comm_manager
        return comm.kvcache_comm.bytes_pushed if comm and comm.kvcache_comm else 0

    def check_health(self) -> None:
        for proc in self.processes:
            if not proc.is_alive():
                raise RuntimeError("a worker process died")

    def shutdown(self, graceful: bool = True) -> None:
        """Ends the worker processes.  Never waits on a worker for long: one that is stuck (inside a
        rendezvous, a collective or a HIP call) is killed."""
        try:
            if self.split:
                self.driver_worker.destroy_kvcache_comm()
        except Exception:
            pass
        if graceful:
            for ch in self.channels.values():
                try:
                    ch.send(("shutdown",))
                except Exception:
                    pass
            deadline = time.monotonic() + 10.0
            for ch in self.channels.values():
                try:
                    ch.recv(timeout=max(deadline - time.monotonic(), 0.1))
                except Exception:
                    pass
        for proc in self.processes:
            proc.join(timeout=10 if graceful else 0.1)
            if proc.is_alive():
                proc.kill()
                proc.join(timeout=5)
        self.processes = []
        self.channels = {}
        if self.tp > 1:
            import torch.distributed as dist
            from ..model_executor import parallel_state
            parallel_state.destroy_model_parallel()
            if dist.is_initialized():
                dist.destroy_process_group()
            self.tp = 1

    def __del__(self):
        try:
            self.shutdown()
        except Exception:
            pass
