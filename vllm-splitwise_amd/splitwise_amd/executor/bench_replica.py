"""One replica of a multi-GPU benchmark leg: an engine process with the workers it starts itself.

``bench.py --gpus N`` is a launcher that never touches a GPU (executor/bench_launch.py).  For every leg
of its plan it starts R of these replica processes, fresh, each with its own list of devices:

  kind "colocated"  one MIXED pool: a GPUExecutor on devices[0], or (tensor_parallel_size > 1) a
                    PoolExecutor over ``devices``;
  kind "split"      a prompt pool followed by one or more token pools (vllm/config.py:433-443):
                    SplitwiseExecutor (pools of one GPU) or PoolExecutor (tensor-parallel pools).

Replicas share nothing: requests are independent units, the only GPU-to-GPU traffic is each replica's own
KV push (and its pools' all-reduces).  They meet twice, through the launcher's TCP store: once before the
timed steps and once after them, so that every replica times exactly K engine steps while all the others
are running too; the launcher takes the slowest replica's time and the sum of the tokens.
"""
import datetime
import importlib
import json
import os
import sys
import time
from typing import Any, Dict, List

import torch

XGMI_LINK_PEAK_GBS = 153.0          # one xGMI link, the path of one prompt -> token pair
HBM_PEAK_GBS = 8000.0


class StoreBarrier:
    """Meeting points of a leg's replicas on the launcher's TCP store (no process group: an engine process
    may already own torch.distributed's default group for its tensor-parallel ranks)."""

    def __init__(self, port: int, leg: str, index: int, replicas: int, timeout_s: float) -> None:
        import torch.distributed as dist
        self.leg, self.index, self.replicas = leg, index, replicas
        self.timeout = datetime.timedelta(seconds=timeout_s)
        self.store = dist.TCPStore("127.0.0.1", port, is_master=False, timeout=self.timeout) if replicas > 1 else None

    def meet(self, name: str) -> None:
        if self.store is None:
            return
        self.store.set(f"{self.leg}/{name}/{self.index}", "1")
        self.store.wait([f"{self.leg}/{name}/{i}" for i in range(self.replicas)], self.timeout)


def _worker_cls(spec: Dict[str, Any]):
    """The worker implementation: the product's, or (dry runs of the launcher on CPU) a stand-in named as
    ``module:Class``."""
    name = spec.get("worker_cls")
    if not name:
        return None
    module, cls = name.split(":")
    return getattr(importlib.import_module(module), cls)


def build_engine(spec: Dict[str, Any]):
    from .. import EngineArgs, LLMEngine
    conc = spec["concurrency"]
    seq_blocks = -(-(spec["prompt_len"] + spec["decode_len"] + 1) // 16)
    split = spec["kind"] == "split"
    tp = spec.get("tp", 1)
    args = EngineArgs(
        model=spec["model"], dtype=spec["dtype"], quantization=spec.get("quantization"),
        kv_cache_dtype=spec.get("kv_cache_dtype", "auto"), sep_prompt_token=split, tensor_parallel_size=tp,
        max_num_seqs=conc, max_model_len=2048, max_num_batched_tokens=spec["max_num_batched_tokens"],
        num_gpu_blocks_override=max(int(conc * seq_blocks * 1.25) + 64, 2048 // 16 + 8), swap_space=0.25,
        enforce_eager=spec.get("enforce_eager", False), max_context_len_to_capture=1024,
        prefill_batch_tokens=spec["prefill_batch_tokens"], prefill_max_wait=spec["prefill_max_wait"],
        pipeline_steps=spec.get("pipeline_steps", True), token_pools=spec.get("token_pools", 1),
        overlap_prompt_decode=bool(spec.get("overlap")) and not split and tp == 1,
        # a prompt pool keeps only the prompts in flight (the reference's same-numbering layout would hold a second
        # full cache on the prompt GPUs for nothing); SPLITWISE_BENCH_REFERENCE_LAYOUT=1 measures that layout
        staged_prompt_cache=split and spec.get("pipeline_steps", True)
        and os.environ.get("SPLITWISE_BENCH_REFERENCE_LAYOUT") != "1")
    mc, cc, pc, sc, dc = args.create_engine_configs()
    devices = list(spec["devices"])
    if pc.one_gpu_overlap:
        from .overlap_executor import OverlapExecutor
        worker_cls = _worker_cls(spec)
        extra = {"worker_cls": worker_cls} if worker_cls is not None else {}
        executor = OverlapExecutor(mc, cc, pc, sc, dc, num_gpu_blocks_override=args.num_gpu_blocks_override,
                                   device_index=devices[0], **extra)
        return LLMEngine(mc, cc, pc, sc, dc, executor=executor), executor
    if len(devices) != pc.world_size:
        raise ValueError(f"replica of kind {spec['kind']} with tp={tp} needs {pc.world_size} devices, got {devices}")
    worker_cls = _worker_cls(spec)
    extra = {"worker_cls": worker_cls} if worker_cls is not None else {}
    transport = os.environ.get("SPLITWISE_KV_TRANSPORT", "kernel")
    if tp > 1:
        from .pool_executor import PoolExecutor
        executor = PoolExecutor(mc, cc, pc, sc, dc, num_gpu_blocks_override=args.num_gpu_blocks_override,
                                devices=devices, transport=transport, **extra)
    elif split:
        from .splitwise_executor import SplitwiseExecutor
        executor = SplitwiseExecutor(mc, cc, pc, sc, dc, num_gpu_blocks_override=args.num_gpu_blocks_override,
                                     prompt_device=devices[0], token_devices=devices[1:], transport=transport,
                                     **extra)
    else:
        from .gpu_executor import GPUExecutor
        executor = GPUExecutor(mc, cc, pc, sc, dc, num_gpu_blocks_override=args.num_gpu_blocks_override,
                               device_index=devices[0], **extra)
    return LLMEngine(mc, cc, pc, sc, dc, executor=executor), executor


def _kv_push_bandwidth(executor, same_device: bool, iters: int = 6):
    """Times the push kernel alone: every layer's K and V of a run of blocks, prompt -> token."""
    worker = executor.driver_worker
    manager = getattr(worker, "kvcache_comm_manager", None)
    comm = getattr(manager, "kvcache_comm", None)
    if comm is None or not hasattr(comm, "push_stream") or comm.push_stream is None:
        return None
    num_layers = len(worker.gpu_cache)
    num_blocks = min(128, worker.cache_config.num_gpu_blocks, comm.peer_num_blocks())
    per_iter = 2 * num_layers * num_blocks * comm.block_size
    stream = comm.push_stream

    def one():
        for layer in range(num_layers):
            comm.put_many(layer, [(0, num_blocks)])

    one()
    torch.cuda.synchronize()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record(stream)
    for _ in range(iters):                            # a fixed, small amount of work: bounded even on a slow path
        one()
    end.record(stream)
    end.synchronize()
    secs = start.elapsed_time(end) * 1e-3
    gbs = per_iter * iters / secs / 1e9
    peak = HBM_PEAK_GBS / 2 if same_device else XGMI_LINK_PEAK_GBS       # a local copy reads and writes HBM
    return {"achieved": round(gbs, 1), "peak": peak, "unit": "GB/s", "path": "local HBM" if same_device else "xGMI",
            "frac": round(gbs / peak, 4), "bytes_per_iter": per_iter, "iters": iters,
            "push_workgroups": comm.push_workgroups[0],               # 0: one per 64 KiB chunk, uncapped
            "push_calibration_gbs": comm.push_calibration.get(0)}     # {cap: GB/s} when set-up measured it


def _roofline_from_profile(prof: Dict[str, float], where: str):
    if not prof or not prof.get("launches"):
        return None
    avg_s = prof["total_ms"] * 1e-3 / prof["launches"]
    achieved = prof["alg_bytes"] / prof["launches"] / avg_s / 1e9
    return {"bound": "hbm", "kernel": f"paged_attention_kernel (decode, {where})", "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
            "avg_launch_us": round(avg_s * 1e6, 2), "launches_timed": prof["launches"],
            "alg_bytes_per_launch": int(prof["alg_bytes"] / prof["launches"])}


def _running_context_lens(engine) -> List[int]:
    lens = []
    for sched in getattr(engine, "schedulers", [engine.scheduler]):
        lens += [g.get_seqs()[0].get_len() for g in sched.running if g.get_seqs()[0].get_output_len() > 0]
    return lens


def run(spec: Dict[str, Any]) -> None:
    """Never hangs on a failure: any exception ends this process at once with a non-zero status (a
    transport set-up call that is stuck inside the HIP runtime would block a normal interpreter exit);
    the launcher then stops the other replicas of the leg."""
    try:
        _run(spec)
    except BaseException:                      # noqa: BLE001
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(0)            # worker processes are daemons of this one; nothing is left to clean up


def _run(spec: Dict[str, Any]) -> None:
    import bench as bench_main                       # the closed-loop driver lives in bench.py
    verbose = os.environ.get("SPLITWISE_BENCH_VERBOSE") == "1"
    index, replicas, leg = spec["index"], spec["replicas"], spec["leg"]
    fake = bool(spec.get("worker_cls"))
    on_gpu = torch.cuda.is_available() and not fake
    if fake:
        # a dry run of the launcher with stand-in workers: they keep their "caches" as files, every replica in
        # a directory of its own
        import tempfile
        scratch = tempfile.mkdtemp(prefix=f"dryrun_{leg}_{index}_", dir=os.path.dirname(spec["result"]))
        os.environ["SPLITWISE_DRYRUN_DIR"] = scratch
    if on_gpu:
        torch.cuda.set_device(spec["devices"][0])
    barrier = StoreBarrier(spec["store_port"], leg, index, replicas, spec.get("barrier_timeout_s", 900.0))
    engine, executor = build_engine(spec)
    conc = spec["concurrency"] * spec.get("token_pools", 1)
    loop = bench_main.ClosedLoop(engine, conc, spec["prompt_len"], spec["decode_len"], seed=index)
    steps, warmup = spec["steps"], spec["warmup"]

    def progress(step, live):
        if verbose:
            print(f"[{leg} #{index}] fill step {step} live={live}", file=sys.stderr, flush=True)

    def sync():
        if on_gpu:
            torch.cuda.synchronize()

    steady = True if spec.get("no_fill") else loop.fill(progress)
    for _ in range(warmup):
        loop.step(False)
    sync()
    barrier.meet("start")
    loop.open_window()                   # tokens of steps launched before this point are not the region's
    t0 = time.perf_counter()
    ctx_samples: List[List[int]] = []
    stride = max(1, steps // 12)
    step_ms: List[float] = []
    for i in range(steps):
        t_step = time.perf_counter()
        loop.step(True)
        step_ms.append(round((time.perf_counter() - t_step) * 1e3, 3))
        if i % stride == stride // 2 and len(ctx_samples) < 12:      # decode batches of the timed region
            lens = _running_context_lens(engine)
            if lens:
                ctx_samples.append(lens)
    sync()
    elapsed = time.perf_counter() - t0
    barrier.meet("end")                  # nobody tears its pools down while another replica is still timing
    loop.close_window()                  # ... and the steps still in flight ran inside the region

    good, total, met, seen, ttfts, tpots = loop.goodput_tokens()
    win = loop.window_stats()
    steady = bool(steady and not spec.get("no_fill") and win["live_min"] >= 0.95 * conc)
    same_device = len(set(spec["devices"])) == 1
    result: Dict[str, Any] = {
        "leg": leg, "index": index, "kind": spec["kind"], "devices": spec["devices"], "elapsed_s": elapsed,
        "steps": steps, "good_tokens": good, "total_tokens": total, "slo_met": met, "requests_seen": seen,
        "live_min": win["live_min"], "live_mean": win["live_mean"], "steady_state": steady,
        "fill_steps": loop.fill_steps, "concurrency": conc,
        "prompt_steps": loop.step_kinds[True][0], "decode_steps": loop.step_kinds[False][0],
        "ttft_p50_ms": round(sorted(ttfts)[len(ttfts) // 2] * 1e3, 2) if ttfts else None,
        "ttft_p99_ms": round(sorted(ttfts)[int(len(ttfts) * 0.99)] * 1e3, 2) if ttfts else None,
        "tpot_p50_ms": round(sorted(tpots)[len(tpots) // 2] * 1e3, 2) if tpots else None,
        "tpot_p99_ms": round(sorted(tpots)[int(len(tpots) * 0.99)] * 1e3, 2) if tpots else None,
        "pipelined": bool(getattr(engine, "pipeline_split", False) or getattr(engine, "pipeline_steps", False)),
        "kv_bytes_pushed": int(executor.kv_bytes_pushed()) if hasattr(executor, "kv_bytes_pushed") else 0,
        "kv_migration": None, "roofline": None,
        "step_ms_max": max(step_ms) if step_ms else None,      # host time of the slowest engine call
    }
    if verbose:
        result["step_ms"] = step_ms
    # the probes come after the measurement and must never cost it: a probe that fails is reported as such
    try:
        if on_gpu and spec["kind"] == "split":
            result["kv_migration"] = _kv_push_bandwidth(executor, same_device)
        if on_gpu and index == 0 and ctx_samples and not spec.get("no_roofline"):
            if spec["kind"] == "split":
                prof = executor.profile_token_pool("profile_decode_attention", samples=ctx_samples)
                result["roofline"] = _roofline_from_profile(prof, "first token GPU of replica 0")
            else:
                prof = executor.driver_worker.profile_decode_attention(ctx_samples)
                result["roofline"] = _roofline_from_profile(prof, "replica 0")
        if on_gpu and spec["kind"] == "split":
            executor.driver_worker.check_kv_status()
    except Exception as exc:             # noqa: BLE001
        import traceback
        traceback.print_exc()
        result["probe_error"] = f"{type(exc).__name__}: {exc}"[:400]
    try:
        executor.shutdown()
    except Exception:                    # noqa: BLE001  (the numbers are in hand; the processes die with this one)
        pass
    tmp = spec["result"] + ".tmp"
    with open(tmp, "w") as f:
        json.dump(result, f)
    os.replace(tmp, spec["result"])
