"""One worker per GPU: owns the model replica (or TP shard), the KV cache and the KV transport.

Method names and the step contract follow vllm/worker/worker.py:35-403:
  * worker kinds: MIXED (co-located), PROMPT (first half of the world), TOKEN (second half);
    rank r of the prompt pool is paired with rank r + num_prompt_workers (:126-144);
  * execute_model (:271-334): cache swap / copy first; a TOKEN worker then waits on the hand-off
    slot of every sequence in ``blocks_to_nw`` before its first decode step; the model runs (a
    PROMPT worker pushes each layer's blocks from inside the attention layers); a PROMPT worker
    finally signals every slot, ordered after all layers' pushes.
"""
import enum
import gc
import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from ..config import CacheConfig, DeviceConfig, ModelConfig, ParallelConfig, SchedulerConfig
from ..model_executor import parallel_state
from ..sequence import SamplerOutput, SequenceGroupMetadata
from .cache_engine import CacheEngine
from .model_runner import ModelRunner


class WorkerType(enum.Enum):
    PROMPT = enum.auto()
    TOKEN = enum.auto()
    MIXED = enum.auto()


class Worker:
    def __init__(self, model_config: ModelConfig, parallel_config: ParallelConfig,
                 scheduler_config: SchedulerConfig, device_config: DeviceConfig, local_rank: int = 0,
                 rank: int = 0, distributed_init_method: Optional[str] = None, kv_cache_dtype: str = "auto",
                 is_driver_worker: bool = True, kv_transport_init: Optional[dict] = None,
                 distributed_store=None) -> None:
        self.distributed_store = distributed_store    # rank 0 of a pool executor: the store it already serves
        self.model_config = model_config
        self.parallel_config = parallel_config
        self.scheduler_config = scheduler_config
        self.device_config = device_config
        self.local_rank = local_rank
        self.rank = rank
        self.distributed_init_method = distributed_init_method
        self.is_driver_worker = is_driver_worker
        self.kv_transport_init = kv_transport_init
        self.model_runner = ModelRunner(model_config, parallel_config, scheduler_config, device_config,
                                        kv_cache_dtype=kv_cache_dtype, is_driver_worker=is_driver_worker)
        self.cache_config: Optional[CacheConfig] = None
        self.cache_engine: Optional[CacheEngine] = None
        self.gpu_cache = None
        self.kvcache_comm_manager = None
        self.worker_type = WorkerType.MIXED
        self.device: Optional[torch.device] = None
        self.init_gpu_memory = 0

    def is_prompt_worker(self) -> bool:
        return self.worker_type == WorkerType.PROMPT

    def is_token_worker(self) -> bool:
        return self.worker_type == WorkerType.TOKEN

    def is_mixed_worker(self) -> bool:
        return self.worker_type == WorkerType.MIXED

    # ------------------------------------------------------------------ set-up
    def init_model(self) -> None:
        if self.device_config.device.type != "cuda":
            raise RuntimeError(f"Not support device type: {self.device_config.device}")
        self.device = torch.device(f"cuda:{self.local_rank}")
        torch.cuda.set_device(self.device)
        self.model_runner.device = self.device
        torch.cuda.empty_cache()
        self.init_gpu_memory = torch.cuda.mem_get_info()[0]
        # only tensor parallelism needs a process group: a prompt/token pair talks through the KV
        # transport and the engine's control channel
        if self.parallel_config.tensor_parallel_size > 1 and not dist.is_initialized():
            if not self.distributed_init_method:
                raise ValueError("distributed_init_method must be set if torch.distributed is not already "
                                 "initialized")
            # "nccl" is RCCL on ROCm; SPLITWISE_TP_BACKEND=gloo lets several ranks share one GPU (tests)
            import datetime
            # bounded: a rank that died while starting must not leave the others waiting for half an hour
            timeout = datetime.timedelta(seconds=float(os.environ.get("SPLITWISE_DIST_TIMEOUT_S", "600")))
            backend = os.environ.get("SPLITWISE_TP_BACKEND", "nccl")
            if self.distributed_store is not None:
                # the store is up already (its port was bound when it was picked); the key prefix is the
                # one init_process_group puts in front of a store it builds from an init_method itself
                dist.init_process_group(backend=backend, world_size=self.parallel_config.world_size, rank=self.rank,
                                        store=dist.PrefixStore("default_pg", self.distributed_store), timeout=timeout)
            else:
                dist.init_process_group(backend=backend, world_size=self.parallel_config.world_size, rank=self.rank,
                                        init_method=self.distributed_init_method, timeout=timeout)
        if dist.is_initialized() and self.parallel_config.tensor_parallel_size > 1:
            warm = torch.zeros(1, device="cpu" if dist.get_backend() == "gloo" else self.device)
            dist.all_reduce(warm)                                      # warm-up (worker.py:452-455)
        parallel_state.initialize_model_parallel(self.parallel_config.tensor_parallel_size,
                                                 self.parallel_config.sep_prompt_token,
                                                 token_pools=max(self.parallel_config.num_token_pools, 1))
        if (self.parallel_config.tensor_parallel_size > 1
                and not getattr(self.parallel_config, "disable_custom_all_reduce", False)):
            # worker.py:100-103 of the reference: small all-reduces go through direct peer reads
            from ..model_executor.custom_all_reduce import init_custom_ar
            init_custom_ar(parallel_state.get_tensor_model_parallel_rank(),
                           self.parallel_config.tensor_parallel_size,
                           parallel_state.get_tensor_model_parallel_group())
        self.init_kvcache_comm(self.kv_transport_init)
        torch.manual_seed(self.model_config.seed)

    def init_kvcache_comm(self, kv_transport_init: Optional[dict] = None) -> None:
        if kv_transport_init is None:
            return
        from .kv_comm import KVCacheCommManager
        n = self.parallel_config.num_prompt_workers
        self.kvcache_comm_manager = KVCacheCommManager(self.rank, self.parallel_config.world_size, n,
                                                       kv_transport_init)
        self.worker_type = WorkerType.PROMPT if self.rank < n else WorkerType.TOKEN
        # the first rank of each pool drives it (worker.py:138-144)
        self.model_runner.driver_rank = (self.rank // n) * n
        if self.rank == self.model_runner.driver_rank:
            self.is_driver_worker = True
            self.model_runner.is_driver_worker = True

    def load_model(self) -> None:
        self.model_runner.load_model()

    @torch.inference_mode()
    def profile_num_available_blocks(self, block_size: int, gpu_memory_utilization: float, cpu_swap_space: int,
                                     cache_dtype: str) -> Tuple[int, int]:
        """Blocks that fit beside the weights and the peak activations (worker.py:180-227)."""
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()
        self.model_runner.profile_run()
        torch.cuda.synchronize()
        free_gpu_memory, total_gpu_memory = torch.cuda.mem_get_info()
        peak_memory = self.init_gpu_memory - free_gpu_memory
        cache_block_size = self.get_cache_block_size_bytes(block_size, cache_dtype)
        num_gpu_blocks = int((total_gpu_memory * gpu_memory_utilization - peak_memory) // cache_block_size)
        num_cpu_blocks = int(cpu_swap_space // cache_block_size)
        gc.collect()
        torch.cuda.empty_cache()
        return max(num_gpu_blocks, 0), max(num_cpu_blocks, 0)

    def init_cache_engine(self, cache_config: CacheConfig) -> None:
        self.cache_config = cache_config
        self.cache_engine = CacheEngine(cache_config, self.model_config, self.parallel_config, self.device)
        self.gpu_cache = self.cache_engine.gpu_cache
        self.model_runner.set_block_size(self.cache_engine.block_size)

    def warm_up_model(self) -> None:
        # decode graphs; a prompt-pool worker never decodes
        if not self.model_config.enforce_eager and not self.is_prompt_worker():
            self.model_runner.capture_model(self.gpu_cache)
        torch.manual_seed(self.model_config.seed)

    def num_captured_graphs(self) -> int:
        return len(self.model_runner.graph_runners)

    def setup_kvcache_comm(self) -> None:
        if self.kvcache_comm_manager is None:
            return
        num_layers = self.model_config.get_num_layers(self.parallel_config)
        self.kvcache_comm_manager.setup_comm(num_layers, self.gpu_cache)
        self.set_comm_for_attention_modules()

    def destroy_kvcache_comm(self) -> None:
        if self.kvcache_comm_manager is not None:
            self.kvcache_comm_manager.destroy_comm()
            for module in self._attention_modules():
                module.kvcache_comm_manager = None

    def _attention_modules(self):
        return [m for m in self.model_runner.model.modules() if m.__class__.__name__ == "Attention"]

    def set_comm_for_attention_modules(self) -> None:
        for i, module in enumerate(self._attention_modules()):
            module.set_kvcache_comm_manager(self.kvcache_comm_manager)
            module.layer_id = i

    # -------------------------------------------------------------------- step
    def cache_swap(self, blocks_to_swap_in: Dict[int, int], blocks_to_swap_out: Dict[int, int],
                   blocks_to_copy: Dict[int, List[int]]) -> None:
        issued = False
        if blocks_to_swap_in:
            self.cache_engine.swap_in(blocks_to_swap_in)
            issued = True
        if blocks_to_swap_out:
            self.cache_engine.swap_out(blocks_to_swap_out)
            issued = True
        if issued:
            for event in self.cache_engine.events:
                event.wait()                      # the compute stream waits, the host does not
        if blocks_to_copy:
            self.cache_engine.copy(blocks_to_copy)

    def should_execute(self, is_prompt: bool) -> bool:
        return (self.is_mixed_worker() or (self.is_prompt_worker() and is_prompt)
                or (self.is_token_worker() and not is_prompt))

    @torch.inference_mode()
    def execute_model(self, seq_group_metadata_list: Optional[List[SequenceGroupMetadata]] = None,
                      blocks_to_swap_in: Optional[Dict[int, int]] = None,
                      blocks_to_swap_out: Optional[Dict[int, int]] = None,
                      blocks_to_copy: Optional[Dict[int, List[int]]] = None,
                      blocks_to_nw: Optional[Dict[int, List[Tuple[int, int]]]] = None,
                      nw_epochs: Optional[Dict[int, int]] = None, defer: bool = False, nw_peer: int = 0,
                      track_pending: bool = True):
        """``defer`` leaves the sampled tokens on the device (see ModelRunner.execute_model).
        ``nw_epochs`` (optional) gives the hand-off count of every slot in ``blocks_to_nw`` as the
        engine counts it; without it each side counts its own signals / waits.  ``nw_peer`` (prompt
        workers that feed several token pools): the pool this step's sequences will decode on."""
        blocks_to_nw = blocks_to_nw or {}
        nw_epochs = nw_epochs or {}
        self.cache_swap(blocks_to_swap_in or {}, blocks_to_swap_out or {}, blocks_to_copy or {})
        if not seq_group_metadata_list:
            return SamplerOutput([])
        is_prompt = seq_group_metadata_list[0].is_prompt
        comm = self.kvcache_comm_manager
        if blocks_to_nw and comm is not None and self.is_token_worker() and not is_prompt:
            for sem_id in blocks_to_nw:
                comm.wait(sem_id, nw_epochs.get(sem_id))
        output = self.model_runner.execute_model(seq_group_metadata_list, self.gpu_cache, blocks_to_nw, defer=defer,
                                                 nw_peer=nw_peer,
                                                 share_samples=defer and not self.is_prompt_worker(),
                                                 track_pending=track_pending)
        if blocks_to_nw and comm is not None and self.is_prompt_worker() and is_prompt:
            for sem_id in blocks_to_nw:
                if nw_peer:
                    comm.signal_and_flush(sem_id, nw_epochs.get(sem_id), nw_peer)
                else:
                    comm.signal_and_flush(sem_id, nw_epochs.get(sem_id))
            # the pushes and signals run on the transport's own stream: the step is over when they have left
            # too (its blocks may be handed out again as soon as it is reported finished)
            pushed = comm.pushes_done_event() if hasattr(comm, "pushes_done_event") else None
            if pushed is not None:
                if defer and hasattr(output, "extra_events"):
                    output.extra_events.append(pushed)
                else:
                    pushed.synchronize()
        # a bounded wait that gave up (hand-off signal, all-reduce barrier) lets the stream run on over
        # data that never arrived: look at the status words once the step has finished and fail loudly
        # (the reference's proxy_channel.wait(1e8) traps, its all-reduce barrier hangs)
        if defer and hasattr(output, "after_sync") and hasattr(output, "event"):
            output.after_sync.append(self.check_step_status)        # looked at when the step's samples are collected
        else:
            self.check_step_status()
        return output

    def check_step_status(self) -> None:
        """Raises RuntimeError when a device-side wait of a FINISHED step timed out."""
        comm = self.kvcache_comm_manager
        if comm is not None and self.is_token_worker():
            comm.check_status_after_step()
        if self.parallel_config.tensor_parallel_size > 1:
            from ..model_executor.custom_all_reduce import get_handle
            handle = get_handle()
            if handle is not None:
                handle.check_status_after_step()

    def check_kv_status(self) -> None:
        """Host-synchronising form for callers outside a step (tests, the end of a benchmark)."""
        if self.kvcache_comm_manager is not None:
            self.kvcache_comm_manager.check_status()

    # ------------------------------------------- transport self-test (worker.py:350-386)
    def set_gpucache(self) -> None:
        num_layers = self.model_config.get_num_layers(self.parallel_config)
        for layer in range(num_layers):
            for t in (0, 1):
                self.gpu_cache[layer][t].fill_(self.rank * (num_layers * 2) + layer * 2 + t)
        torch.cuda.synchronize()

    def send_recv_kvcache_all(self) -> None:
        if self.kvcache_comm_manager is None:
            return
        num_layers = self.model_config.get_num_layers(self.parallel_config)
        if self.rank < self.parallel_config.num_prompt_workers:
            for layer in range(num_layers):
                self.kvcache_comm_manager.put(0, layer, 0, self.cache_config.num_gpu_blocks)
            self.kvcache_comm_manager.signal_and_flush(0)
        else:
            self.kvcache_comm_manager.wait(0)
        torch.cuda.synchronize()

    def check_gpucache(self) -> None:
        if self.kvcache_comm_manager is None:
            return
        n = self.parallel_config.num_prompt_workers
        num_layers = self.model_config.get_num_layers(self.parallel_config)
        src = self.rank if self.rank < n else self.rank - n
        for layer in range(num_layers):
            for t in (0, 1):
                expected = src * (num_layers * 2) + layer * 2 + t
                cache = self.gpu_cache[layer][t]
                assert torch.all(cache.float() == float(expected)), (layer, t, expected)

    @property
    def max_model_len(self) -> int:
        return self.model_config.max_model_len

    @property
    def vocab_size(self) -> int:
        return self.model_runner.vocab_size

    @torch.inference_mode()
    def profile_decode_attention(self, samples: List[List[int]]) -> Dict[str, float]:
        """Measurement aid: times the decode paged-attention launch of every layer over this worker's
        own cache for batches with the given context lengths (block tables drawn at random from the
        cache), with HIP events on the launch stream.  -> total ms, launches, algorithmic bytes."""
        from ..model_executor.input_metadata import InputMetadata
        from ..model_executor.layers.attention import PagedAttentionImpl
        cfg, par = self.model_config, self.parallel_config
        h = cfg.hf_config["num_attention_heads"] // par.tensor_parallel_size
        kvh, d = cfg.get_num_kv_heads(par), cfg.get_head_size()
        kc0 = self.gpu_cache[0][0]
        es, bs, nb = kc0.element_size(), self.cache_config.block_size, kc0.shape[0]
        scale = d ** -0.5
        stream = torch.cuda.current_stream(self.device)
        gen = torch.Generator(device="cpu").manual_seed(0)
        total_ms, launches, alg_bytes = 0.0, 0, 0.0
        for lens in samples:
            n = len(lens)
            if n == 0:
                continue
            max_blocks = max(-(-c // bs) for c in lens)
            tables = torch.randint(0, nb, (n, max_blocks), generator=gen, dtype=torch.int32).to(self.device)
            ctx = torch.tensor(lens, dtype=torch.int32, device=self.device)
            md = InputMetadata(is_prompt=False, slot_mapping=torch.empty(0, dtype=torch.long, device=self.device),
                               prompt_lens=None, max_seq_len=None, start_loc=None, max_context_len=max(lens),
                               context_lens=ctx, block_tables=tables, use_cuda_graph=False,
                               kv_cache_dtype=self.cache_config.cache_dtype)
            q = torch.empty(n, h, d, dtype=cfg.dtype, device=self.device).uniform_(-scale, scale)
            for kc, vc in self.gpu_cache[:2]:
                PagedAttentionImpl.forward_decode(q, kc, vc, md, kvh, scale, None)
            start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record(stream)
            for kc, vc in self.gpu_cache:
                PagedAttentionImpl.forward_decode(q, kc, vc, md, kvh, scale, None)
            end.record(stream)
            end.synchronize()
            total_ms += start.elapsed_time(end)
            launches += len(self.gpu_cache)
            alg_bytes += (2.0 * sum(lens) * kvh * d * es + 2.0 * n * h * d * q.element_size() + tables.numel() * 4) * len(
                self.gpu_cache)
        return {"total_ms": total_ms, "launches": launches, "alg_bytes": alg_bytes}

    def get_cache_block_size_bytes(self, block_size: int, cache_dtype: str) -> int:
        return CacheEngine.get_cache_block_size(block_size, cache_dtype, self.model_config, self.parallel_config)
