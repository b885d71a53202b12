"""Launcher of the multi-GPU benchmark legs (``bench.py --gpus N``, N > 1).

This process never touches a GPU.  It lays out the N devices of the node into legs -- each leg a set of
independent replicas (executor/bench_replica.py), each replica an engine process that starts its own
workers -- runs the legs one after the other, every replica a FRESH child process, and prints one JSON
line: the headline leg (N/2 pairs of one prompt GPU and one token GPU, the world of
vllm/config.py:433-443 with pools of one) with the other legs beside it:

  colocated   the same N GPUs as N co-located replicas, timed in the same run: ``vs_colocated``;
  asymmetric  one prompt GPU feeding two token GPUs (which the reference cannot express,
              vllm/config.py:434-438), as many such trios as fit;
  config legs the quantised and tensor-parallel configurations that exist at this N (13B GPTQ int4 + fp8
              cache as TP2 + TP2 pools and as pairs at N = 4; 70B as TP4 + TP4 pools at N = 8), each
              with its own co-located counterpart.

It works the same under ``python bench.py --gpus N`` and under ``python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N``: there rank 0 is the launcher and the other ranks leave at once
(their GPUs are driven by the replicas' own processes).
"""
import json
import os
import signal
import subprocess
import sys
import tempfile
import time
from typing import Any, Dict, List, Optional

_TORCHRUN_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE",
                 "ROLE_RANK", "ROLE_NAME", "ROLE_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                 "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_RUN_ID",
                 "TORCHELASTIC_USE_AGENT_STORE", "TORCHELASTIC_ERROR_FILE", "TORCH_NCCL_ASYNC_ERROR_HANDLING")

SLO_TTFT = 0.25
SLO_TPOT = 0.10


def _replica(kind: str, devices: List[int], model: str, dtype: str, **kw) -> Dict[str, Any]:
    spec = {"kind": kind, "devices": devices, "model": model, "dtype": dtype}
    spec.update(kw)
    return spec


def _pairs_layout(n: int, model: str, dtype: str, **kw) -> List[Dict[str, Any]]:
    """N/2 pairs of (prompt GPU, token GPU); an odd GPU left over serves co-located."""
    reps = [_replica("split", [2 * i, 2 * i + 1], model, dtype, **kw) for i in range(n // 2)]
    if n % 2:
        reps.append(_replica("colocated", [n - 1], model, dtype, **kw))
    return reps


def _trios_layout(n: int, model: str, dtype: str, **kw) -> List[Dict[str, Any]]:
    """As many (1 prompt GPU -> 2 token GPUs) trios as fit; two GPUs left over form a pair, one serves
    co-located."""
    reps, dev = [], 0
    while n - dev >= 3:
        reps.append(_replica("split", [dev, dev + 1, dev + 2], model, dtype, token_pools=2, **kw))
        dev += 3
    if n - dev == 2:
        reps.append(_replica("split", [dev, dev + 1], model, dtype, **kw))
    elif n - dev == 1:
        reps.append(_replica("colocated", [dev], model, dtype, **kw))
    return reps


def _describe(replicas: List[Dict[str, Any]]) -> str:
    names: Dict[str, int] = {}
    for r in replicas:
        tp = r.get("tp", 1)
        if r["kind"] == "colocated":
            tag = "co-located" if tp == 1 else f"TP{tp} co-located"
            if r.get("overlap"):
                tag += ", prompt beside decode"
        else:
            pools = r.get("token_pools", 1)
            tag = f"1P+{pools}D" if tp == 1 else (f"TP{tp}+TP{tp}" if pools == 1 else f"TP{tp}+{pools}xTP{tp}")
        names[tag] = names.get(tag, 0) + 1
    return " + ".join(f"{c}x({t})" if c > 1 else t for t, c in names.items())


def build_plan(n: int, model: str = "llama-2-7b", dtype: str = "bfloat16", legs: str = "default",
               quantization: Optional[str] = None, kv_cache_dtype: str = "auto") -> List[Dict[str, Any]]:
    """The legs of an N-GPU run, the headline leg first."""
    base = dict(quantization=quantization, kv_cache_dtype=kv_cache_dtype)
    plan = [
        {"name": "pairs", "role": "primary", "replicas": _pairs_layout(n, model, dtype, **base)},
        {"name": "colocated", "role": "colocated", "of": "pairs",
         "replicas": [_replica("colocated", [i], model, dtype, **base) for i in range(n)]},
    ]
    if n >= 3:
        plan.append({"name": "asymmetric", "role": "extra", "of": "pairs",
                     "replicas": _trios_layout(n, model, dtype, **base)})
    # the co-located replicas once more with prompt steps BESIDE their decode steps (EngineArgs.overlap_prompt_decode):
    # the best one GPU does by itself, for the split legs to be held against
    plan.append({"name": "colocated_overlap", "role": "extra", "of": "pairs",
                 "replicas": [_replica("colocated", [i], model, dtype, overlap=True, **base) for i in range(n)]})
    q13 = dict(quantization="gptq", kv_cache_dtype="fp8_e5m2")
    if n == 4:
        # (the tensor-parallel leg last: should a pool hang on its first run over real links, the legs before it
        # have their numbers)
        plan += [
            {"name": "13b_gptq_fp8_pairs", "role": "config",
             "replicas": _pairs_layout(4, "llama-2-13b", "float16", **q13)},
            {"name": "13b_gptq_fp8_colocated", "role": "colocated", "of": "13b_gptq_fp8_tp2_pools",
             "replicas": [_replica("colocated", [i], "llama-2-13b", "float16", **q13) for i in range(4)]},
            {"name": "13b_gptq_fp8_tp2_pools", "role": "config",
             "replicas": [_replica("split", [0, 1, 2, 3], "llama-2-13b", "float16", tp=2, **q13)]},
        ]
    if n == 8:
        plan += [
            {"name": "70b_tp4_pools", "role": "config",
             "replicas": [_replica("split", list(range(8)), "llama-2-70b", "bfloat16", tp=4)]},
            {"name": "70b_tp4_colocated", "role": "colocated", "of": "70b_tp4_pools",
             "replicas": [_replica("colocated", [0, 1, 2, 3], "llama-2-70b", "bfloat16", tp=4),
                          _replica("colocated", [4, 5, 6, 7], "llama-2-70b", "bfloat16", tp=4)]},
        ]
    if legs == "primary":
        plan = plan[:2]
    elif legs not in ("default", "all"):
        wanted = [name.strip() for name in legs.split(",") if name.strip()]
        unknown = [w for w in wanted if w not in {leg["name"] for leg in plan}]
        if unknown:
            raise SystemExit(f"unknown leg(s) {unknown}; this N has {[leg['name'] for leg in plan]}")
        plan = [leg for leg in plan if leg["name"] in wanted]
    for leg in plan:
        leg["layout"] = _describe(leg["replicas"])
    return plan


class LegFailed(RuntimeError):
    pass


def _kill(procs: List[subprocess.Popen]) -> None:
    for p in procs:
        if p.poll() is None:
            try:
                os.killpg(p.pid, signal.SIGKILL)       # the replica's own session: its workers go with it
            except (ProcessLookupError, PermissionError):
                pass
    for p in procs:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            pass


def run_leg(leg: Dict[str, Any], common: Dict[str, Any], store_port: int, workdir: str, timeout_s: float,
            bench_path: str, active: List[subprocess.Popen]) -> List[Dict[str, Any]]:
    """Starts every replica of the leg as a fresh process, waits for all of them (or for the first
    failure, which ends the others), returns their results."""
    env = {k: v for k, v in os.environ.items() if k not in _TORCHRUN_ENV}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if os.environ.get("SPLITWISE_BENCH_ONE_GPU") == "1":
        # RCCL refuses several ranks on one device: the rehearsal's tensor-parallel groups run over gloo (the
        # all-reduces of up to 8 MiB still go through the peer-read kernel)
        env.setdefault("SPLITWISE_TP_BACKEND", "gloo")
    env.update(leg.get("env", {}))
    procs: List[subprocess.Popen] = []
    logs = []
    n = len(leg["replicas"])
    for i, rep in enumerate(leg["replicas"]):
        spec = dict(common)
        spec.update({k: v for k, v in rep.items() if v is not None})
        spec.update(leg=leg["name"], index=i, replicas=n, store_port=store_port,
                    result=os.path.join(workdir, f"{leg['name']}.{i}.json"), barrier_timeout_s=timeout_s)
        log = open(os.path.join(workdir, f"{leg['name']}.{i}.log"), "w")
        logs.append(log)
        p = subprocess.Popen([sys.executable, "-u", bench_path, "--replica", json.dumps(spec)], stdout=log,
                             stderr=subprocess.STDOUT, env=env, start_new_session=True)
        procs.append(p)
        active.append(p)
    deadline = time.monotonic() + timeout_s
    try:
        while True:
            codes = [p.poll() for p in procs]
            if any(c not in (None, 0) for c in codes):
                bad = next(i for i, c in enumerate(codes) if c not in (None, 0))
                raise LegFailed(f"replica {bad} of leg {leg['name']} exited with status {codes[bad]}: "
                                + _tail(os.path.join(workdir, f"{leg['name']}.{bad}.log")))
            if all(c == 0 for c in codes):
                break
            if time.monotonic() > deadline:
                raise LegFailed(f"leg {leg['name']} did not finish within {timeout_s:.0f} s")
            time.sleep(0.2)
    finally:
        _kill(procs)
        for p in procs:
            if p in active:
                active.remove(p)
        for log in logs:
            log.close()
    results = []
    for i in range(n):
        path = os.path.join(workdir, f"{leg['name']}.{i}.json")
        if not os.path.exists(path):
            raise LegFailed(f"replica {i} of leg {leg['name']} left no result: " + _tail(path[:-5] + ".log"))
        with open(path) as f:
            results.append(json.load(f))
    return results


def _tail(path: str, lines: int = 12) -> str:
    try:
        with open(path, errors="replace") as f:
            return " | ".join(line.rstrip() for line in f.readlines()[-lines:])
    except OSError:
        return "(no log)"


def summarize(leg: Dict[str, Any], results: List[Dict[str, Any]]) -> Dict[str, Any]:
    """Whole-leg figures: tokens of all replicas over the slowest replica's time."""
    elapsed = max(r["elapsed_s"] for r in results)
    good = sum(r["good_tokens"] for r in results)
    total = sum(r["total_tokens"] for r in results)
    seen = sum(r["requests_seen"] for r in results)
    gpus = sum(len(r["devices"]) for r in results)
    kv = [r["kv_migration"] for r in results if r.get("kv_migration")]
    out = {
        "layout": leg["layout"], "n_gpus": gpus, "replicas": len(results),
        "value": round(good / elapsed, 2), "throughput_tok_s": round(total / elapsed, 2),
        "per_gpu_tok_s": round(good / elapsed / max(gpus, 1), 2),
        "elapsed_s": round(elapsed, 4), "ms_per_step": round(elapsed / results[0]["steps"] * 1e3, 3),
        "slo_attainment": round(sum(r["slo_met"] for r in results) / max(seen, 1), 4),
        "live_min": min(r["live_min"] for r in results),
        "steady_state": all(r["steady_state"] for r in results),
        "pipelined": all(r["pipelined"] for r in results),
        "ttft_p50_ms": _median([r["ttft_p50_ms"] for r in results]),
        "tpot_p50_ms": _median([r["tpot_p50_ms"] for r in results]),
        "kv_bytes_pushed": sum(r["kv_bytes_pushed"] for r in results),
        "model": leg["replicas"][0]["model"],
    }
    if kv:
        # every pair has its own link: the slowest one is what a prompt waits for
        worst = min(kv, key=lambda k: k["frac"])
        out["kv_migration"] = dict(worst, pairs_measured=len(kv),
                                   aggregate_gbs=round(sum(k["achieved"] for k in kv), 1))
    roof = next((r["roofline"] for r in results if r.get("roofline")), None)
    if roof:
        out["roofline"] = roof
    probe_errors = [r["probe_error"] for r in results if r.get("probe_error")]
    if probe_errors:
        out["probe_error"] = probe_errors[0]
    return out


def _median(values):
    values = sorted(v for v in values if v is not None)
    return values[len(values) // 2] if values else None


def main(args) -> int:
    rank = int(os.environ.get("RANK", "0"))
    if rank != 0:
        return 0              # under torchrun: rank 0 lays out the whole node, the other ranks have nothing to do
    n = args.gpus
    one_gpu = os.environ.get("SPLITWISE_BENCH_ONE_GPU") == "1"
    dtype = args.dtype or ("float16" if args.quantization else "bfloat16")
    plan = build_plan(n, args.model, dtype, args.legs, args.quantization, args.kv_cache_dtype)
    if one_gpu:               # rehearsal on a one-GPU box: every worker of every replica shares device 0
        stand_in = os.environ.get("SPLITWISE_BENCH_REHEARSAL_70B", "llama-2-13b")
        for leg in plan:
            for rep in leg["replicas"]:
                rep["devices"] = [0] * len(rep["devices"])
                if rep["model"] == "llama-2-70b":
                    # eight ranks of a 70B model are 280 GB of weights: on one GPU a smaller model of the same
                    # architecture walks through the same code
                    rep["model"] = stand_in
    common = {
        "concurrency": args.concurrency, "prompt_len": args.prompt_len, "decode_len": args.decode_len,
        "steps": args.steps, "warmup": args.warmup, "max_num_batched_tokens": args.max_num_batched_tokens,
        "prefill_batch_tokens": args.prefill_batch_tokens, "prefill_max_wait": args.prefill_max_wait,
        "pipeline_steps": not args.no_pipeline_steps, "enforce_eager": args.enforce_eager,
        "no_fill": args.no_fill, "no_roofline": args.no_roofline,
    }
    if getattr(args, "worker_cls", None):
        common["worker_cls"] = args.worker_cls
    import torch.distributed as dist
    import datetime
    store = dist.TCPStore("127.0.0.1", 0, None, is_master=True, wait_for_workers=False,
                          timeout=datetime.timedelta(seconds=args.leg_timeout))
    bench_path = os.path.abspath(sys.argv[0])
    began = time.monotonic()
    summaries: Dict[str, Any] = {}
    active: List[subprocess.Popen] = []
    state = {"printed": False}

    def emit() -> int:
        if state["printed"]:
            return 0
        state["printed"] = True
        line = compose(args, n, plan, summaries, one_gpu, dtype)
        print(json.dumps(line), flush=True)
        return 0 if line.get("value") is not None else 1

    def on_term(signum, frame):          # the caller's patience ran out: report the legs that finished
        _kill(list(active))
        emit()
        os._exit(143)

    signal.signal(signal.SIGTERM, on_term)
    with tempfile.TemporaryDirectory(prefix="splitwise_bench_") as workdir:
        keep = os.environ.get("SPLITWISE_BENCH_LOG_DIR")
        for leg in plan:                  # (the list may grow by a fallback leg while it is walked)
            left = args.budget_s - (time.monotonic() - began)
            # a run whose split legs failed (or hung until their time limit) still owes the line a value: the
            # co-located leg of the same GPUs then runs whatever is left of the budget
            owed = (leg["role"] == "colocated" and leg.get("of") == plan[0]["name"]
                    and not any("value" in s for s in summaries.values()))
            if leg["role"] != "primary" and left < 60 and not owed:
                summaries[leg["name"]] = {"layout": leg["layout"], "skipped": "time budget of the run used up"}
                continue
            timeout = args.leg_timeout if leg["role"] == "primary" else min(args.leg_timeout,
                                                                            max(left, 300 if owed else 60))
            try:
                results = run_leg(leg, common, store.port, workdir, timeout, bench_path, active)
                summaries[leg["name"]] = summarize(leg, results)
            except LegFailed as exc:
                summaries[leg["name"]] = {"layout": leg["layout"], "error": str(exc)[-1500:]}
                print(f"[bench] {exc}", file=sys.stderr, flush=True)
                fallback = runtime_transport_fallback(leg)
                if fallback is not None:
                    plan.insert(plan.index(leg) + 1, fallback)
            if keep:
                import shutil
                os.makedirs(keep, exist_ok=True)
                for name in os.listdir(workdir):
                    if name.endswith(".log") or name.endswith(".json"):
                        shutil.copy(os.path.join(workdir, name), os.path.join(keep, name))
    return emit()


def runtime_transport_fallback(leg: Dict[str, Any]) -> Optional[Dict[str, Any]]:
    """The headline leg once more with the KV blocks moved by the HIP runtime's device-to-device copies instead of
    the push kernel's stores into the peer's mapped cache (worker/kv_comm.py, transport "runtime"), for a node on
    which the kernel transport does not come up.  Only for the headline leg, only once, and not if the run already
    asked for that transport."""
    if leg["role"] != "primary" or not any(r["kind"] == "split" for r in leg["replicas"]):
        return None
    if os.environ.get("SPLITWISE_KV_TRANSPORT", "kernel") != "kernel":
        return None
    return {"name": leg["name"] + "_runtime_transport", "role": "primary_fallback", "of": leg["name"],
            "replicas": leg["replicas"], "layout": leg["layout"], "env": {"SPLITWISE_KV_TRANSPORT": "runtime"}}


def compose(args, n: int, plan, summaries: Dict[str, Any], one_gpu: bool, dtype: str) -> Dict[str, Any]:
    # the headline: the first leg; if it failed, the same layout over the runtime transport; if that failed too,
    # the co-located leg of the same GPUs, named as such in the line
    primary_leg = plan[0]
    note = None
    for leg in plan:
        if leg is plan[0] or leg["role"] == "primary_fallback" or (leg["role"] == "colocated" and
                                                                    leg.get("of") == plan[0]["name"]):
            if "value" in summaries.get(leg["name"], {}):
                if leg is not plan[0]:
                    note = (f"the headline leg ({plan[0]['name']}) failed: value is the leg {leg['name']} "
                            f"({leg['layout']}); see legs")
                primary_leg = leg
                break
    primary = summaries.get(primary_leg["name"]) or {}
    ok = "value" in primary
    dtype_tag = {"bfloat16": "bf16", "float16": "f16", "half": "f16", "float32": "f32"}.get(dtype, dtype)
    for leg in plan:                      # every leg against its co-located counterpart
        if leg["role"] == "colocated" and "value" in summaries.get(leg["name"], {}):
            colo = summaries[leg["name"]]
            for other in plan:
                s = summaries.get(other["name"], {})
                same_model = other["replicas"][0]["model"] == leg["replicas"][0]["model"]
                if other is not leg and other["role"] != "colocated" and same_model and "value" in s:
                    # per GPU, so that legs over fewer GPUs than the co-located one compare too
                    s["vs_colocated"] = round(s["per_gpu_tok_s"] / max(colo["per_gpu_tok_s"], 1e-9), 4)
                    s["colocated_leg"] = leg["name"]
    path = "device-local copies (one-GPU rehearsal: every worker shares device 0)" if one_gpu else "xGMI"
    moved = f"KV blocks pushed prompt->token over {path}"
    if primary_leg["role"] == "colocated":
        moved = "no KV migration (co-located replicas: the split legs failed)"
    elif primary_leg["role"] == "primary_fallback":
        moved += " by the HIP runtime's copies (the push kernel's transport failed)"
    state_tag = "steady state" if primary.get("steady_state") else "RAMP (not steady state)"
    model_tag = "Llama-2-7B" if args.model == "llama-2-7b" else args.model
    line = {
        "metric": f"goodput (tok/s under TTFT+TBT SLO), {model_tag} split 1P+1D vs co-located",
        "value": primary.get("value") if ok else None, "unit": "tok/s", "n_gpus": n, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": primary.get("ms_per_step"), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": dtype_tag, "data": "synthetic",
        "config": {
            "workload": f"{args.model} {dtype_tag}, {primary_leg['layout']}: {moved}, "
                        f"closed loop, {state_tag}: {primary.get('live_min')} requests in flight per replica at least "
                        f"({args.concurrency} per token GPU asked for), prefill {args.prompt_len} / decode "
                        f"{args.decode_len} tokens, random-init weights, SLO TTFT<={SLO_TTFT}s TPOT<={SLO_TPOT}s",
            "parallelism": primary_leg["layout"], "concurrency_target": args.concurrency,
            "live_min": primary.get("live_min"), "steady_state": primary.get("steady_state"),
            "pipeline_steps": primary.get("pipelined"), "one_gpu_rehearsal": one_gpu,
            "headline_leg": primary_leg["name"],
            "kv_transport": primary_leg.get("env", {}).get("SPLITWISE_KV_TRANSPORT",
                                                            os.environ.get("SPLITWISE_KV_TRANSPORT", "kernel")),
            "prefill_batch_tokens": args.prefill_batch_tokens, "prefill_max_wait_s": args.prefill_max_wait},
        "throughput_tok_s": primary.get("throughput_tok_s"), "slo_attainment": primary.get("slo_attainment"),
        "vs_colocated": primary.get("vs_colocated"),
        "kv_migration": primary.get("kv_migration"), "roofline": primary.get("roofline"),
        "cpu_baseline": None,             # timed at N=1 only (one bounded CPU sample per round is enough)
        "ttft_p50_ms": primary.get("ttft_p50_ms"), "tpot_p50_ms": primary.get("tpot_p50_ms"),
        "legs": summaries,
    }
    if not ok:
        line["error"] = f"the headline leg ({primary_leg['name']}) did not produce a result: see legs"
    elif note:
        line["error"] = note
    return line
