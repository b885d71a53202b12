"""Executor for one split pair: a prompt worker in this process, a token worker in another.

Role of vllm/executor/ray_gpu_executor.py with ``sep_prompt_token`` (:87-232 worker set-up,
:234-282 cache sizing over ALL workers, :284-324 / :396-452 stage dispatch, :491-493 KV
transport set-up), for pools of one GPU each.  Two differences, both deliberate:
  * no Ray: the token worker is a child process (or a torchrun rank) reached over a small
    control channel; KV bytes go GPU to GPU;
  * ``execute_split`` runs a prompt step and a decode step AT THE SAME TIME on the two GPUs --
    the reference dispatches one stage per engine step and leaves the other GPU idle
    (ray_gpu_executor.py:292-293).
"""
from typing import Any, Dict, List, Optional, Tuple

import torch

from ..config import CacheConfig, DeviceConfig, ModelConfig, ParallelConfig, SchedulerConfig
from ..sequence import SamplerOutput, SequenceGroupMetadata
from ..sequence import slim_decode_metadata, unpack_sampler_output
from ..worker.worker import Worker
from .channel import PipeChannel
from .executor_base import ExecutorBase
from .gpu_executor import check_block_budget


class SplitwiseExecutor(ExecutorBase):
    def __init__(self, model_config: ModelConfig, cache_config: CacheConfig, parallel_config: ParallelConfig,
                 scheduler_config: SchedulerConfig, device_config: DeviceConfig,
                 num_gpu_blocks_override: Optional[int] = None, channel=None, prompt_device: int = 0,
                 token_device: Optional[int] = None, transport: str = "kernel", worker_cls=Worker) -> None:
        if not parallel_config.sep_prompt_token:
            raise ValueError("SplitwiseExecutor needs sep_prompt_token=True")
        if parallel_config.tensor_parallel_size != 1:
            raise NotImplementedError("split pools of more than one GPU (TP > 1) are not built yet")
        self.model_config = model_config
        self.cache_config = cache_config
        self.parallel_config = parallel_config
        self.scheduler_config = scheduler_config
        self.transport = transport
        self.process = None
        if channel is None:
            if token_device is None:
                token_device = 1 if torch.cuda.device_count() > 1 else prompt_device
            import torch.multiprocessing as mp
            ctx = mp.get_context("spawn")
            parent, child = ctx.Pipe(duplex=True)
            from .remote_worker import spawn_entry
            cfg = self._remote_cfg(token_device)
            self.process = ctx.Process(target=spawn_entry, args=(child, cfg), daemon=True)
            self.process.start()
            child.close()
            channel = PipeChannel(parent)
        self.channel = channel
        self._pending_kv: List[Any] = []
        self.driver_worker = worker_cls(model_config, parallel_config, scheduler_config, device_config,
                                    local_rank=prompt_device, rank=0, kv_cache_dtype=cache_config.cache_dtype,
                                    is_driver_worker=True,
                                    kv_transport_init={"exchange": self._exchange, "transport": transport})
        # bring both workers up in parallel: the remote call returns when it is done
        self._remote_async("init_model")
        self.driver_worker.init_model()
        self._remote_wait()
        self._remote_async("load_model")
        self.driver_worker.load_model()
        self._remote_wait()
        self._init_cache(num_gpu_blocks_override)

    def _remote_cfg(self, token_device: int) -> Dict[str, Any]:
        return {"model_config": self.model_config, "parallel_config": self.parallel_config,
                "scheduler_config": self.scheduler_config, "cache_config": self.cache_config, "rank": 1,
                "local_rank": token_device, "transport": self.transport}

    # ------------------------------------------------------------ control plane
    def _exchange(self, obj):
        """KV-transport handshake of the local prompt worker with the remote token worker."""
        tag, peer = self._recv()
        if tag != "kv_exchange":          # e.g. ("error", traceback): the worker's set-up failed before it got here
            raise RuntimeError(f"token worker failed during the KV-transport handshake: {tag!r} {peer}")
        self.channel.send(("kv_exchange", obj))
        return peer

    def _remote_async(self, name: str, **kwargs) -> None:
        self.channel.send(("call", name, kwargs))

    def _recv(self):
        """The token worker's next message; a worker that has died ends the wait with an error."""
        if self.process is None or not hasattr(self.channel, "poll"):
            return self.channel.recv()
        while True:
            try:
                return self.channel.recv(timeout=1.0)
            except TimeoutError:
                if not self.process.is_alive() and not self.channel.poll(0.0):
                    raise RuntimeError(f"token worker process died (exit code {self.process.exitcode})") from None
            except (EOFError, OSError) as exc:
                raise RuntimeError(f"token worker process went away (exit code {self.process.exitcode})") from exc

    def _remote_wait(self):
        status, result = self._recv()
        if status == "samples":
            return unpack_sampler_output(result)
        if status != "ok":
            raise RuntimeError(f"token worker failed: {result}")
        return result

    def _remote_call(self, name: str, **kwargs):
        self._remote_async(name, **kwargs)
        return self._remote_wait()

    # -------------------------------------------------------------------- cache
    def _init_cache(self, override: Optional[int]) -> None:
        cc = self.cache_config
        if override is None:
            self._remote_async("profile_num_available_blocks", block_size=cc.block_size,
                               gpu_memory_utilization=cc.gpu_memory_utilization,
                               cpu_swap_space=cc.swap_space_bytes, cache_dtype=cc.cache_dtype)
            local = self.driver_worker.profile_num_available_blocks(cc.block_size, cc.gpu_memory_utilization,
                                                                    cc.swap_space_bytes, cc.cache_dtype)
            remote = self._remote_wait()
            # block ids mean the same on both GPUs, so both caches get the smaller size
            num_gpu_blocks, num_cpu_blocks = min(local[0], remote[0]), min(local[1], remote[1])
        else:
            per_block = self.driver_worker.get_cache_block_size_bytes(cc.block_size, cc.cache_dtype)
            num_gpu_blocks, num_cpu_blocks = override, int(cc.swap_space_bytes // per_block)
        check_block_budget(self.model_config.max_model_len, cc.block_size, num_gpu_blocks)
        cc.num_gpu_blocks, cc.num_cpu_blocks = num_gpu_blocks, num_cpu_blocks
        self._remote_async("init_cache_engine", cache_config=cc)
        self.driver_worker.init_cache_engine(cc)
        self._remote_wait()
        # the prompt pool never decodes, so only the token pool captures decode graphs
        self._remote_async("warm_up_model")
        self.driver_worker.model_runner.enforce_eager = True
        self._remote_wait()
        # transport: the remote side starts its handshake, the local side answers it
        self._remote_async("setup_kvcache_comm")
        self.driver_worker.setup_kvcache_comm()
        self._remote_wait()

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
        self.channel.send(("execute", dict(kwargs, seq_group_metadata_list=slim_decode_metadata(seq_group_metadata_list))))
        return self._remote_wait()

    def execute_split(self, prompt_kwargs: Optional[dict], decode_kwargs: Optional[dict]):
        """Runs a prompt step here and a decode step on the token GPU concurrently.
        -> (prompt SamplerOutput or None, decode SamplerOutput or None)"""
        if decode_kwargs is not None:
            slim = slim_decode_metadata(decode_kwargs["seq_group_metadata_list"])
            self.channel.send(("execute", dict(decode_kwargs, seq_group_metadata_list=slim)))
        prompt_out = self.driver_worker.execute_model(**prompt_kwargs) if prompt_kwargs is not None else None
        decode_out = self._remote_wait() if decode_kwargs is not None else None
        return prompt_out, decode_out

    # pipelined pools: neither call waits for the step it starts
    supports_pipelined_split = True

    def launch_prompt(self, prompt_kwargs: dict):
        """Starts a prompt step on the prompt GPU; -> its PendingSamples (done() / resolve())."""
        return self.driver_worker.execute_model(**prompt_kwargs, defer=True)

    def decode_deferred(self, decode_kwargs: Optional[dict]) -> Optional[SamplerOutput]:
        """Starts a decode step on the token GPU (None: starts nothing) and returns the samples of
        the decode step started by the previous call, or None if there was none."""
        if decode_kwargs is not None:
            slim = slim_decode_metadata(decode_kwargs["seq_group_metadata_list"])
            self.channel.send(("execute_deferred", dict(decode_kwargs, seq_group_metadata_list=slim)))
        else:
            self.channel.send(("flush",))
        return self._remote_wait()

    def kv_bytes_pushed(self) -> int:
        comm = self.driver_worker.kvcache_comm_manager
        return comm.kvcache_comm.bytes_pushed if comm and comm.kvcache_comm else 0

    def check_health(self) -> None:
        if self.process is not None and not self.process.is_alive():
            raise RuntimeError("token worker process died")

    def shutdown(self) -> None:
        try:
            self.driver_worker.destroy_kvcache_comm()
            self.channel.send(("shutdown",))
            self.channel.recv(timeout=10.0)          # a stuck worker is killed below, not waited for
        except Exception:
            pass
        if self.process is not None:
            self.process.join(timeout=10)
            if self.process.is_alive():
                self.process.kill()

    def __del__(self):
        try:
            self.shutdown()
        except Exception:
            pass
