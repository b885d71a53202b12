"""Executor for split pools of one GPU each: a prompt worker in this process, token workers in others.

Role of vllm/executor/ray_gpu_executor.py with ``sep_prompt_token`` (:87-232 worker set-up,
:234-282 cache sizing over ALL workers, :284-324 / :396-452 stage dispatch, :491-493 KV
transport set-up), for pools of one GPU each.  Differences, all deliberate:
  * no Ray: a token worker is a child process (or a torchrun rank) reached over a small
    control channel; KV bytes go GPU to GPU;
  * ``execute_split`` runs a prompt step and a decode step AT THE SAME TIME on the two GPUs --
    the reference dispatches one stage per engine step and leaves the other GPU idle
    (ray_gpu_executor.py:292-293);
  * one prompt worker may feed SEVERAL token workers (``parallel_config.num_token_pools``; the
    reference's world is as many token workers as prompt workers, vllm/config.py:434-438): worker
    1 + j is token pool j, with a cache -- a block space -- of its own;
  * with ``parallel_config.staged_prompt_cache`` the prompt worker's cache holds only the prompts in
    flight (a few hundred blocks) under block numbers of its own; by default a block number names the
    same block on both sides, as in the reference (vllm/worker/comm_utils.py:132-147).
"""
import copy
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch

from ..config import CacheConfig, DeviceConfig, ModelConfig, ParallelConfig, SchedulerConfig
from ..sequence import SamplerOutput, SequenceGroupMetadata
from ..sequence import slim_decode_metadata, unpack_sampler_output
from ..worker.worker import Worker
from .channel import PipeChannel
from .executor_base import ExecutorBase
from .gpu_executor import check_block_budget


def staged_prompt_blocks(scheduler_config: SchedulerConfig, block_size: int) -> int:
    """Blocks a prompt pool needs when it keeps only the prompts in flight: one prompt step's worth -- at
    most ``max_num_batched_tokens`` tokens in at most ``max_num_seqs`` prompts, each of which may end in a
    partly filled block -- twice over, so that the next step can be laid out while one is in flight."""
    per_step = -(-scheduler_config.max_num_batched_tokens // block_size) + scheduler_config.max_num_seqs
    return 2 * per_step


class SplitwiseExecutor(ExecutorBase):
    def __init__(self, model_config: ModelConfig, cache_config: CacheConfig, parallel_config: ParallelConfig,
                 scheduler_config: SchedulerConfig, device_config: DeviceConfig,
                 num_gpu_blocks_override: Optional[int] = None, channel=None, prompt_device: int = 0,
                 token_device: Optional[int] = None, transport: str = "kernel", worker_cls=Worker,
                 token_devices: Optional[Sequence[int]] = None, channels: Optional[Sequence[Any]] = None) -> None:
        if not parallel_config.sep_prompt_token:
            raise ValueError("SplitwiseExecutor needs sep_prompt_token=True")
        if parallel_config.tensor_parallel_size != 1:
            raise NotImplementedError("split pools of more than one GPU (TP > 1) are PoolExecutor's")
        self.model_config = model_config
        self.cache_config = cache_config
        self.parallel_config = parallel_config
        self.scheduler_config = scheduler_config
        self.transport = transport
        self.worker_cls = worker_cls
        self.num_lanes = parallel_config.num_token_pools
        self.staged = parallel_config.staged_prompt_cache
        self.processes: List[Any] = []
        if channel is not None:
            channels = [channel]
        if channels is None:
            if token_devices is None:
                if token_device is not None:
                    token_devices = [token_device]
                else:
                    n = max(torch.cuda.device_count(), 1)
                    token_devices = [(prompt_device + 1 + j) % n for j in range(self.num_lanes)]
            if len(token_devices) != self.num_lanes:
                raise ValueError(f"{self.num_lanes} token pool(s) need as many devices, got {list(token_devices)}")
            import torch.multiprocessing as mp
            from .remote_worker import spawn_entry
            ctx = mp.get_context("spawn")
            channels = []
            for lane, device in enumerate(token_devices):
                parent, child = ctx.Pipe(duplex=True)
                proc = ctx.Process(target=spawn_entry, args=(child, self._remote_cfg(1 + lane, device)), daemon=True)
                proc.start()
                child.close()
                self.processes.append(proc)
                channels.append(PipeChannel(parent))
        self.channels = list(channels)
        if len(self.channels) != self.num_lanes:
            raise ValueError(f"{self.num_lanes} token pool(s) need as many control channels")
        self.driver_worker = worker_cls(model_config, parallel_config, scheduler_config, device_config,
                                        local_rank=prompt_device, rank=0, kv_cache_dtype=cache_config.cache_dtype,
                                        is_driver_worker=True,
                                        kv_transport_init={"exchange": self._exchange, "transport": transport})
        try:
            # bring all workers up in parallel: a remote call returns when it is done
            self._remote_async_all("init_model")
            self.driver_worker.init_model()
            self._remote_wait_all()
            self._remote_async_all("load_model")
            self.driver_worker.load_model()
            self._remote_wait_all()
            self._init_cache(num_gpu_blocks_override)
        except BaseException:
            self._kill()
            raise

    # the single-pair names of earlier callers
    @property
    def channel(self):
        return self.channels[0]

    @property
    def process(self):
        return self.processes[0] if self.processes else None

    def _remote_cfg(self, rank: int, device: int) -> Dict[str, Any]:
        return {"model_config": self.model_config, "parallel_config": self.parallel_config,
                "scheduler_config": self.scheduler_config, "cache_config": self.cache_config, "rank": rank,
                "local_rank": device, "transport": self.transport,
                "worker_cls": None if self.worker_cls is Worker else self.worker_cls}

    # ------------------------------------------------------------ control plane
    def _exchange(self, obj, lane: int = 0):
        """KV-transport handshake of the local prompt worker with the token worker of ``lane``."""
        tag, peer = self._recv(lane)
        if tag != "kv_exchange":          # e.g. ("error", traceback): the worker's set-up failed before it got here
            raise RuntimeError(f"token worker {lane} failed during the KV-transport handshake: {tag!r} {peer}")
        self.channels[lane].send(("kv_exchange", obj))
        return peer

    def _remote_async(self, name: str, lane: int = 0, **kwargs) -> None:
        self.channels[lane].send(("call", name, kwargs))

    def _remote_async_all(self, name: str, **kwargs) -> None:
        for lane in range(self.num_lanes):
            self._remote_async(name, lane, **kwargs)

    def _recv(self, lane: int = 0):
        """The token worker's next message; a worker that has died ends the wait with an error."""
        channel = self.channels[lane]
        process = self.processes[lane] if lane < len(self.processes) else None
        if process is None or not hasattr(channel, "poll"):
            return channel.recv()
        while True:
            try:
                return channel.recv(timeout=1.0)
            except TimeoutError:
                if not process.is_alive() and not channel.poll(0.0):
                    raise RuntimeError(f"token worker {lane} died (exit code {process.exitcode})") from None
            except (EOFError, OSError) as exc:
                raise RuntimeError(f"token worker {lane} went away (exit code {process.exitcode})") from exc

    def _remote_wait(self, lane: int = 0):
        status, result = self._recv(lane)
        if status == "samples":
            return unpack_sampler_output(result)
        if status != "ok":
            raise RuntimeError(f"token worker {lane} failed: {result}")
        return result

    def _remote_wait_all(self) -> List[Any]:
        return [self._remote_wait(lane) for lane in range(self.num_lanes)]

    def _remote_call(self, name: str, lane: int = 0, **kwargs):
        self._remote_async(name, lane, **kwargs)
        return self._remote_wait(lane)

    def profile_token_pool(self, name: str, lane: int = 0, **kwargs):
        """Runs a measurement method of ``Worker`` on the token worker of ``lane``."""
        return self._remote_call(name, lane, **kwargs)

    # -------------------------------------------------------------------- cache
    def _init_cache(self, override: Optional[int]) -> None:
        cc = self.cache_config
        staging = staged_prompt_blocks(self.scheduler_config, cc.block_size) if self.staged else None
        if override is None:
            self._remote_async_all("profile_num_available_blocks", block_size=cc.block_size,
                                   gpu_memory_utilization=cc.gpu_memory_utilization,
                                   cpu_swap_space=cc.swap_space_bytes, cache_dtype=cc.cache_dtype)
            local = self.driver_worker.profile_num_available_blocks(cc.block_size, cc.gpu_memory_utilization,
                                                                    cc.swap_space_bytes, cc.cache_dtype)
            remote = self._remote_wait_all()
            # every token pool gets the same number of blocks; without staging a block number means the same
            # block on the prompt GPU too, so its (smaller or larger) budget counts as well
            sizes = remote if self.staged else remote + [local]
            num_gpu_blocks, num_cpu_blocks = min(s[0] for s in sizes), min(s[1] for s in sizes)
            if self.staged and local[0] < staging:
                raise ValueError(f"the prompt GPU has room for {local[0]} cache blocks, a prompt step needs {staging}")
        else:
            per_block = self.driver_worker.get_cache_block_size_bytes(cc.block_size, cc.cache_dtype)
            num_gpu_blocks, num_cpu_blocks = override, int(cc.swap_space_bytes // per_block)
        check_block_budget(self.model_config.max_model_len, cc.block_size, num_gpu_blocks)
        cc.num_gpu_blocks, cc.num_cpu_blocks = num_gpu_blocks, num_cpu_blocks
        self._remote_async_all("init_cache_engine", cache_config=cc)
        prompt_cc = cc
        if self.staged:
            prompt_cc = copy.copy(cc)
            prompt_cc.num_gpu_blocks, prompt_cc.num_cpu_blocks = staging, 0
        self.prompt_cache_blocks = prompt_cc.num_gpu_blocks
        self.driver_worker.init_cache_engine(prompt_cc)
        self._remote_wait_all()
        # the prompt pool never decodes, so only the token pools capture decode graphs
        self._remote_async_all("warm_up_model")
        self.driver_worker.model_runner.enforce_eager = True
        self._remote_wait_all()
        # transport: the remote sides start their handshake, the local side answers them in turn
        self._remote_async_all("setup_kvcache_comm")
        self.driver_worker.setup_kvcache_comm()
        self._remote_wait_all()

    # --------------------------------------------------------------------- steps
    def execute_model(self, seq_group_metadata_list: List[SequenceGroupMetadata],
                      blocks_to_swap_in: Dict[int, int], blocks_to_swap_out: Dict[int, int],
                      blocks_to_copy: Dict[int, List[int]],
                      blocks_to_nw: Optional[Dict[int, List[Tuple[int, int]]]] = None,
                      nw_epochs: Optional[Dict[int, int]] = None) -> SamplerOutput:
        """The reference's serial dispatch: the step runs on the pool that matches its kind."""
        kwargs = dict(seq_group_metadata_list=seq_group_metadata_list, blocks_to_swap_in=blocks_to_swap_in,
                      blocks_to_swap_out=blocks_to_swap_out, blocks_to_copy=blocks_to_copy,
                      blocks_to_nw=blocks_to_nw or {}, nw_epochs=nw_epochs)
        if not seq_group_metadata_list:
            return SamplerOutput([])
        if seq_group_metadata_list[0].is_prompt:
            return self.driver_worker.execute_model(**kwargs)
        self.channels[0].send(("execute", dict(kwargs, seq_group_metadata_list=slim_decode_metadata(seq_group_metadata_list))))
        return self._remote_wait(0)

    def execute_split(self, prompt_kwargs: Optional[dict], decode_kwargs: Optional[dict]):
        """Runs a prompt step here and a decode step on the token GPU concurrently.
        -> (prompt SamplerOutput or None, decode SamplerOutput or None)"""
        if decode_kwargs is not None:
            slim = slim_decode_metadata(decode_kwargs["seq_group_metadata_list"])
            self.channels[0].send(("execute", dict(decode_kwargs, seq_group_metadata_list=slim)))
        prompt_out = self.driver_worker.execute_model(**prompt_kwargs) if prompt_kwargs is not None else None
        decode_out = self._remote_wait(0) if decode_kwargs is not None else None
        return prompt_out, decode_out

    # pipelined pools: neither call waits for the step it starts
    supports_pipelined_split = True

    def launch_prompt(self, prompt_kwargs: dict, lane: int = 0):
        """Starts a prompt step on the prompt GPU whose sequences will decode on token pool ``lane``;
        -> its PendingSamples (done() / resolve())."""
        if lane:
            return self.driver_worker.execute_model(**prompt_kwargs, defer=True, nw_peer=lane)
        return self.driver_worker.execute_model(**prompt_kwargs, defer=True)

    def decode_send(self, decode_kwargs: Optional[dict], lane: int = 0) -> None:
        """Starts a decode step on token pool ``lane`` (None: starts nothing, only asks for the previous
        step's samples).  Every ``decode_send`` is answered by one ``decode_collect``."""
        if decode_kwargs is not None:
            slim = slim_decode_metadata(decode_kwargs["seq_group_metadata_list"])
            self.channels[lane].send(("execute_deferred", dict(decode_kwargs, seq_group_metadata_list=slim)))
        else:
            self.channels[lane].send(("flush",))

    def decode_collect(self, lane: int = 0) -> Optional[SamplerOutput]:
        """The samples of the decode step started by the call BEFORE the last ``decode_send`` of this lane,
        or None if there was none."""
        return self._remote_wait(lane)

    def decode_deferred(self, decode_kwargs: Optional[dict], lane: int = 0) -> Optional[SamplerOutput]:
        self.decode_send(decode_kwargs, lane)
        return self.decode_collect(lane)

    def kv_bytes_pushed(self) -> int:
        comm = self.driver_worker.kvcache_comm_manager
        return comm.kvcache_comm.bytes_pushed if comm and comm.kvcache_comm else 0

    def check_health(self) -> None:
        for lane, process in enumerate(self.processes):
            if not process.is_alive():
                raise RuntimeError(f"token worker {lane} died")

    def _kill(self) -> None:
        for process in self.processes:
            if process.is_alive():
                process.kill()
        for process in self.processes:
            process.join(timeout=5)

    def shutdown(self) -> None:
        try:
            self.driver_worker.destroy_kvcache_comm()
        except Exception:
            pass
        for channel in self.channels:
            try:
                channel.send(("shutdown",))
            except Exception:
                pass
        for channel in self.channels:
            try:
                channel.recv(timeout=10.0)           # a stuck worker is killed below, not waited for
            except Exception:
                pass
        for process in self.processes:
            process.join(timeout=10)
        self._kill()
        self.processes = []

    def __del__(self):
        try:
            self.shutdown()
        except Exception:
            pass
