"""Multi-GPU leg of bench.py: N/2 independent (prompt GPU, token GPU) pairs.

Launched by torchrun with one rank per GPU.  Even ranks host an engine plus the pair's prompt
worker, odd ranks the pair's token worker (a command loop).  Pairs share nothing: requests are
independent units, the only GPU-to-GPU traffic is each pair's KV push over its xGMI link, and
the control plane is torch.distributed (gloo) point-to-point objects.  Timing follows the
benchmark contract: barrier + device sync on both sides of exactly K engine steps, maximum over
the engine ranks, aggregate goodput of all pairs.
"""
import json
import os
import time

import torch
import torch.distributed as dist

XGMI_LINK_PEAK_GBS = 153.0          # one xGMI link, the path of one prompt -> token pair
HBM_PEAK_GBS = 8000.0


def _kv_push_bandwidth(executor, iters: int = 6):
    """Times the push kernel alone: every layer's K and V of a run of blocks, prompt -> token."""
    worker = executor.driver_worker
    comm = worker.kvcache_comm_manager.kvcache_comm
    num_layers = len(worker.gpu_cache)
    num_blocks = min(128, worker.cache_config.num_gpu_blocks)
    per_iter = 2 * num_layers * num_blocks * comm.block_size
    stream = comm.push_stream                  # where the pushes run

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
    same_device = os.environ.get("SPLITWISE_BENCH_ONE_GPU") == "1"
    peak = HBM_PEAK_GBS / 2 if same_device else XGMI_LINK_PEAK_GBS       # a local copy reads and writes HBM
    return {"achieved": round(gbs, 1), "peak": peak, "unit": "GB/s", "path": "local HBM" if same_device else "xGMI",
            "frac": round(gbs / peak, 4), "bytes_per_iter": per_iter, "iters": iters}


def run(args, rank: int, local_rank: int, world: int) -> None:
    """Never hangs on a failure: any exception ends this rank at once with a non-zero status (a
    transport set-up call that is stuck inside the HIP runtime would block a normal interpreter
    exit), and torchrun then stops the other ranks."""
    try:
        _run(args, rank, local_rank, world)
    except BaseException:                      # noqa: BLE001
        import traceback
        traceback.print_exc()
        os._exit(1)


def _run(args, rank: int, local_rank: int, world: int) -> None:
    from .. import EngineArgs, LLMEngine
    from .channel import DistChannel
    from .remote_worker import worker_loop
    from .splitwise_executor import SplitwiseExecutor

    if os.environ.get("SPLITWISE_BENCH_VERBOSE") == "1":
        import faulthandler
        import sys
        faulthandler.dump_traceback_later(45, repeat=False, file=sys.stderr)
    if world < 2 or world % 2:
        raise SystemExit("the split benchmark needs an even number of ranks (launch with torchrun)")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # a one-member group makes gloo print to stdout, which must carry the JSON line alone
    engines = dist.new_group(list(range(0, world, 2)), backend="gloo") if world > 2 else None
    if os.environ.get("SPLITWISE_BENCH_ONE_GPU") == "1":
        local_rank = 0              # rehearsal on a one-GPU box: both workers of the pair share device 0
    torch.cuda.set_device(local_rank)
    conc = args.concurrency
    seq_blocks = -(-(args.prompt_len + args.decode_len + 1) // 16)
    engine_args = EngineArgs(model=args.model, sep_prompt_token=True, max_num_seqs=conc, max_model_len=2048,
                             max_num_batched_tokens=args.max_num_batched_tokens,
                             num_gpu_blocks_override=int(conc * seq_blocks * 1.25) + 64, swap_space=0.25,
                             enforce_eager=args.enforce_eager, max_context_len_to_capture=1024,
                             prefill_batch_tokens=args.prefill_batch_tokens, prefill_max_wait=args.prefill_max_wait,
                             pipeline_steps=not getattr(args, "no_pipeline_steps", False))
    mc, cc, pc, sc, dc = engine_args.create_engine_configs()
    transport = os.environ.get("SPLITWISE_KV_TRANSPORT", "kernel")

    if rank % 2 == 1:
        cfg = {"model_config": mc, "parallel_config": pc, "scheduler_config": sc, "cache_config": cc, "rank": 1,
               "local_rank": local_rank, "transport": transport}
        worker_loop(DistChannel(rank - 1), cfg)
        dist.barrier()
        return

    import bench as bench_main                       # the closed-loop driver lives in bench.py
    executor = SplitwiseExecutor(mc, cc, pc, sc, dc, num_gpu_blocks_override=engine_args.num_gpu_blocks_override,
                                 channel=DistChannel(rank + 1), prompt_device=local_rank, transport=transport)
    engine = LLMEngine(mc, cc, pc, sc, dc, executor=executor)
    loop = bench_main.ClosedLoop(engine, conc, args.prompt_len, args.decode_len, seed=rank)
    warmup = args.warmup
    import sys
    verbose = os.environ.get("SPLITWISE_BENCH_VERBOSE") == "1"

    def progress(step, live):
        if verbose:
            print(f"[rank {rank}] fill step {step} live={live}", file=sys.stderr, flush=True)

    # untimed fill phase, independent of --warmup: until every slot holds a full-length request
    steady = True if getattr(args, "no_fill", False) else loop.fill(progress)
    for i in range(warmup):
        loop.step(False)
    torch.cuda.synchronize()
    if engines is not None:
        dist.barrier(group=engines)
    loop.open_window()                   # tokens of steps launched before this point are not the region's
    t0 = time.perf_counter()
    ctx_samples = []
    stride = max(1, args.steps // 12)
    for i in range(args.steps):
        loop.step(True)
        if i % stride == stride // 2 and len(ctx_samples) < 12:           # decode batches of the timed region, for the roofline
            lens = [g.get_seqs()[0].get_len() for g in engine.scheduler.running if g.get_seqs()[0].get_output_len() > 0]
            if lens:
                ctx_samples.append(lens)
        if verbose and i % 20 == 0:
            print(f"[rank {rank}] step {i}", file=sys.stderr, flush=True)
    torch.cuda.synchronize()
    if verbose:
        print(f"[rank {rank}] timed region done", file=sys.stderr, flush=True)
    if engines is not None:
        dist.barrier(group=engines)
    elapsed = time.perf_counter() - t0
    loop.close_window()                  # ... and the steps still in flight ran inside it

    good, total, met, seen, ttfts, tpots = loop.goodput_tokens()
    win = loop.window_stats()
    steady = bool(steady and not getattr(args, "no_fill", False) and win["live_min"] >= 0.95 * conc)
    stats = torch.tensor([good, total, met, seen, executor.kv_bytes_pushed(), win["live_mean"]], dtype=torch.float64)
    tmax = torch.tensor([elapsed], dtype=torch.float64)
    tmin = torch.tensor([float(win["live_min"]), float(steady)], dtype=torch.float64)
    if engines is not None:
        dist.all_reduce(stats, group=engines)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX, group=engines)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN, group=engines)
    kv = _kv_push_bandwidth(executor)
    roofline = None
    if rank == 0 and ctx_samples:
        prof = executor._remote_call("profile_decode_attention", samples=ctx_samples)
        if prof["launches"]:
            avg_s = prof["total_ms"] * 1e-3 / prof["launches"]
            achieved = prof["alg_bytes"] / prof["launches"] / avg_s / 1e9
            roofline = {"bound": "hbm", "kernel": "paged_attention_kernel (decode, token GPU of pair 0)",
                        "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                        "avg_launch_us": round(avg_s * 1e6, 2), "launches_timed": prof["launches"],
                        "alg_bytes_per_launch": int(prof["alg_bytes"] / prof["launches"])}
    executor.driver_worker.kvcache_comm_manager.check_status()
    executor.shutdown()
    if rank == 0:
        elapsed = float(tmax.item())
        pairs = world // 2
        live_min, steady_all = int(tmin[0].item()), bool(tmin[1].item())
        live_mean = round(float(stats[5]) / pairs, 1)
        in_flight = live_min if live_min >= 0.95 * conc else live_mean
        state_tag = "steady state" if steady_all else "RAMP (not steady state)"
        result = {
            "metric": "goodput (tok/s under TTFT+TBT SLO), Llama-2-7B split 1P+1D per pair",
            "value": round(float(stats[0]) / elapsed, 2), "unit": "tok/s", "n_gpus": world, "steps": args.steps,
            "warmup": warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.model} bf16, {pairs} independent pair(s) of 1 prompt GPU + 1 token GPU, "
                                   f"KV blocks pushed prompt->token over xGMI ({transport} transport), closed loop, "
                                   f"{state_tag}: {in_flight:g} requests in flight per pair over the timed steps "
                                   f"({conc} asked for), prefill {args.prompt_len} / decode "
                                   f"{args.decode_len} tokens, random-init weights, SLO TTFT<={bench_main.SLO_TTFT}s "
                                   f"TPOT<={bench_main.SLO_TPOT}s",
                       "parallelism": f"{pairs}x(1P+1D)", "concurrency_per_pair": in_flight,
                       "concurrency_target": conc, "live_min": live_min, "live_mean": live_mean,
                       "fill_steps": loop.fill_steps, "steady_state": steady_all,
                       "pipeline_steps": bool(engine.pipeline_split)},
            "throughput_tok_s": round(float(stats[1]) / elapsed, 2),
            "slo_attainment": round(float(stats[2]) / max(float(stats[3]), 1.0), 4),
            "kv_bytes_pushed": int(stats[4]),
            "kv_migration": kv,
            "roofline": roofline,
            "cpu_baseline": None,             # timed at N=1 only (one bounded CPU sample per round is enough)
            "ttft_p50_ms": round(sorted(ttfts)[len(ttfts) // 2] * 1e3, 2) if ttfts else None,
            "tpot_p50_ms": round(sorted(tpots)[len(tpots) // 2] * 1e3, 2) if tpots else None,
        }
        print(json.dumps(result), flush=True)
    dist.barrier()
