#!/usr/bin/env python3
"""Host cost of an engine step over tensor-parallel pools, measured on CPU with stand-in workers whose steps
cost nothing (tools/host_step_workers.py): the closed loop of bench.py (256 requests in flight, 772-token
prompts, 170 generated tokens) over PoolExecutor with 8 worker ranks (a TP4 prompt pool and a TP4 token pool)
or 4 (a co-located TP4 pool), pipelined against lock-step.  What is left of a step is scheduling, the command
pipes to every rank, output processing and the bench's own bookkeeping.

  python tools/host_cost_tp_pools.py [--tp 4] [--concurrency 256] [--steps 300]
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vllm-splitwise_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]


def run(split: bool, pipeline: bool, tp: int, conc: int, steps: int, profile: bool, token_pools: int = 1):
    import bench
    from host_step_workers import NullWorker
    from splitwise_amd import EngineArgs, LLMEngine
    from splitwise_amd.executor.pool_executor import PoolExecutor
    with tempfile.TemporaryDirectory() as tmp:
        os.environ["SPLIT_FAKE_DIR"] = tmp
        seq_blocks = -(-(772 + 170 + 1) // 16)
        blocks = int(conc * seq_blocks * 1.25) + 64
        os.environ["SPLIT_FAKE_BLOCKS"] = str(blocks)
        args = EngineArgs(model="llama-2-70b", dtype="bfloat16", sep_prompt_token=split, tensor_parallel_size=tp,
                          max_num_seqs=conc, max_model_len=2048, max_num_batched_tokens=4096, swap_space=0.001,
                          prefill_batch_tokens=2300, prefill_max_wait=0.08, pipeline_steps=pipeline,
                          staged_prompt_cache=split and pipeline, token_pools=token_pools)
        mc, cc, pc, sc, dc = args.create_engine_configs()
        if tp == 1:
            from splitwise_amd.executor.splitwise_executor import SplitwiseExecutor
            executor = SplitwiseExecutor(mc, cc, pc, sc, dc, worker_cls=NullWorker, prompt_device=0,
                                         token_devices=[0] * token_pools)
        else:
            executor = PoolExecutor(mc, cc, pc, sc, dc, worker_cls=NullWorker, devices=[0] * pc.world_size)
        try:
            engine = LLMEngine(mc, cc, pc, sc, dc, executor=executor)
            loop = bench.ClosedLoop(engine, conc * token_pools, 772, 170, seed=0)
            loop.fill()
            for _ in range(20):
                loop.step(False)
            prof = None
            if profile:
                import cProfile
                prof = cProfile.Profile()
                prof.enable()
            kinds = {True: [0, 0.0], False: [0, 0.0]}
            t0 = time.perf_counter()
            for _ in range(steps):
                t = time.perf_counter()
                loop.step(True)
                was_prompt = bool(getattr(engine, "last_scheduled_prompt_run", False))
                kinds[was_prompt][0] += 1
                kinds[was_prompt][1] += time.perf_counter() - t
            dt = time.perf_counter() - t0
            if prof is not None:
                prof.disable()
                import pstats
                pstats.Stats(prof).sort_stats("tottime").print_stats(28)
            layout = (f"TP{tp}+{token_pools}xTP{tp}" if split else f"TP{tp} co-located") if tp > 1 else f"1P+{token_pools}D"
            return {"layout": layout, "ranks": pc.world_size,
                    "pipelined": pipeline, "in_flight": conc * token_pools, "ms_per_engine_call": round(dt / steps * 1e3, 3),
                    "calls_with_a_prompt_step": kinds[True][0],
                    "ms_per_call_prompt": round(kinds[True][1] / max(kinds[True][0], 1) * 1e3, 3),
                    "ms_per_call_decode": round(kinds[False][1] / max(kinds[False][0], 1) * 1e3, 3)}
        finally:
            executor.shutdown()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tp", type=int, default=4)
    ap.add_argument("--concurrency", type=int, default=256)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--token-pools", type=int, default=1, help="token pools fed by the prompt pool (with --tp 1: 1P+kD)")
    ap.add_argument("--only", default=None, help="split|colocated, lockstep|pipelined (comma separated)")
    args = ap.parse_args()
    print(f"# host cores here: {os.cpu_count()}", flush=True)
    only = set((args.only or "").split(",")) - {""}
    for split in (True, False):
        for pipeline in (False, True):
            if only and not {"split" if split else "colocated", "pipelined" if pipeline else "lockstep"} >= only:
                continue
            if args.tp == 1 and not (split and pipeline):
                continue                                   # pools of one GPU: the pipelined split engine only
            print(json.dumps(run(split, pipeline, args.tp, args.concurrency, args.steps, args.profile,
                                 args.token_pools if split else 1)), flush=True)


if __name__ == "__main__":
    main()
