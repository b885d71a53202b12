#!/usr/bin/env python3
"""Headline benchmark: goodput (tok/s under a TTFT + TBT SLO) of Llama-2-7B serving.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (same result)

N = 1  co-located serving on one MI355X (the configuration the metric is quoted on).  The engine runs its
       prompt steps BESIDE its decode steps, on a second stream of the one GPU
       (EngineArgs.overlap_prompt_decode; every timed step is then a decode step with prompt work going on
       next to it); --plain-engine lets the two kinds of step take turns as the reference's scheduler emits
       them, and the default run times that engine too, in a child process, and reports it as "plain_engine".
N > 1  this process becomes a launcher that never touches a GPU and starts every engine as a fresh
       child process (splitwise_amd/executor/bench_launch.py; under torchrun rank 0 is the launcher
       and the other ranks leave at once).  Headline leg: N/2 independent (prompt GPU, token GPU)
       pairs, every pair running the split path with the KV blocks of each prompt pushed prefill ->
       decode over xGMI; requests are independent, so pairs share nothing (weak scaling, no data-path
       collective).  Beside it, in the same JSON line under "legs": the same N GPUs as co-located
       replicas ("vs_colocated"), one prompt GPU feeding two token GPUs, and the quantised /
       tensor-parallel configurations that exist at this N (13B GPTQ + fp8 cache at N = 4, 70B
       TP4 + TP4 pools at N = 8).  SPLITWISE_BENCH_ONE_GPU=1 rehearses any of it on a one-GPU box.

A *step* is one engine iteration: one scheduling decision plus one pass of the hot path
(a prompt batch or a decode batch) on the GPU.  The workload is closed-loop: C requests are
kept in flight (a finished request is replaced at once), each with a 772-token synthetic
prompt and 170 generated tokens -- the mean request shape of the reference's own benchmark
logs.  Goodput counts the tokens generated inside the timed region by requests that meet
the reference's SLO (TTFT <= 0.25 s and mean time between tokens <= 0.1 s).  With pipelined steps
a call returns the tokens of the step launched one call earlier; the region is credited with the
tokens of the steps that RAN in it: not those of a step that finished before it began (returned by
its first call), but those of the step still in flight when its closing synchronisation ends.

The run has three phases.  FILL (untimed, independent of --warmup): C / L requests are admitted per
decode step (C in flight, L generated tokens each: the steady state's own admission rate) until C are in
flight, and the loop runs on until every request of that first cohort has finished; completions, and with
them prompt arrivals, then come at an even rate.  (--window-trace prints generated tok/s per run of steps:
earlier fill rules -- one admission per engine step, random lengths for the first cohort -- left a dense
stretch of arrivals that the closed loop replays once per request lifetime.)  WARMUP: W
more untimed steps.  TIMED: exactly K steps between device synchronisations.  What the timed
region actually held (live requests, decode batch sizes, prompt / decode steps) is printed in
`config`, measured, next to what was asked for.

The JSON line also carries
  roofline      decode paged attention (the dominant, HBM-bound kernel): algorithmic KV bytes
                per launch / average launch time, timed with HIP events on the launch stream
                over decode batches sampled from the timed region (launched alone, after the region: inside an
                overlapped step the kernel shares the GPU with a prompt step's kernels);
  cpu_baseline  the CPU oracle (torch restatement of the same model and operators) timed on the
                host cores on a bounded sample, scaled to tok/s.
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "vllm-splitwise_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

PROMPT_LEN = 772
DECODE_LEN = 170
SLO_TTFT = 0.25
SLO_TPOT = 0.10
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--warmup", type=int, default=20, help="untimed steps after the fill phase")
    ap.add_argument("--no-fill", action="store_true",
                    help="skip the fill phase (the timed region then starts on a ramp, and says so)")
    ap.add_argument("--concurrency", type=int, default=256, help="requests kept in flight per replica")
    ap.add_argument("--model", default="llama-2-7b")
    ap.add_argument("--prompt-len", type=int, default=PROMPT_LEN)
    ap.add_argument("--decode-len", type=int, default=DECODE_LEN)
    ap.add_argument("--max-num-batched-tokens", type=int, default=4096)
    ap.add_argument("--prefill-batch-tokens", type=int, default=2300,
                    help="hold waiting prompts until this many tokens have gathered (0 = admit at once)")
    ap.add_argument("--prefill-max-wait", type=float, default=0.08, help="... or the oldest has waited this long (s)")
    ap.add_argument("--no-pipeline-steps", action="store_true",
                    help="wait for a step's tokens before scheduling the next step (the reference's loop)")
    ap.add_argument("--quantization", default=None, choices=[None, "gptq", "awq"],
                    help="int4 weights (fp16 activations); the default workload is unquantised bf16")
    ap.add_argument("--kv-cache-dtype", default="auto", choices=["auto", "fp8_e5m2"])
    ap.add_argument("--dtype", default=None, help="activation dtype (default bf16, fp16 with --quantization)")
    ap.add_argument("--enforce-eager", action="store_true")
    ap.add_argument("--overlap-prompt-decode", action="store_true",
                    help="N = 1 (the default there): prompt steps beside the decode steps, on a second stream of the "
                         "one GPU (EngineArgs.overlap_prompt_decode)")
    ap.add_argument("--plain-engine", action="store_true",
                    help="N = 1: prompt steps and decode steps take turns, as the reference's scheduler emits them "
                         "(implied by --no-pipeline-steps)")
    ap.add_argument("--no-overlap-leg", "--no-other-engine-leg", dest="no_overlap_leg", action="store_true",
                    help="N = 1: do not time the same workload once more on the other engine (a child process)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--window-trace", type=int, default=0,
                    help="print the generated tok/s of every run of this many timed steps (stderr)")
    ap.add_argument("--first-cohort", default="full", choices=["full", "random", "stratified"],
                    help="decode lengths of the first C requests: all L (default), random 1..L, or L*(i+1)/C shuffled")
    # N > 1 only
    ap.add_argument("--legs", default="default",
                    help="N > 1: 'default' = every leg this N has, 'primary' = pairs + co-located only, or a "
                         "comma-separated list of leg names")
    ap.add_argument("--budget-s", type=float, default=840.0,
                    help="N > 1: legs after the headline leg are skipped once the run has taken this long")
    ap.add_argument("--leg-timeout", type=float, default=600.0, help="N > 1: longest any one leg may take (s)")
    ap.add_argument("--replica", default=None, help=argparse.SUPPRESS)          # set by the launcher
    ap.add_argument("--worker-cls", default=None, help=argparse.SUPPRESS)       # CPU dry runs of the launcher
    return ap.parse_args()


class ClosedLoop:
    """Keeps `concurrency` requests in flight; records a timestamp per generated token."""

    def __init__(self, engine, concurrency, prompt_len, decode_len, seed):
        from splitwise_amd import SamplingParams
        self.engine = engine
        self.concurrency = concurrency
        self.prompt_len = prompt_len
        self.rng = random.Random(seed)
        self.vocab = engine.model_config.get_vocab_size()
        self._token_range = range(self.vocab)
        self.params = SamplingParams(temperature=0.0, max_tokens=decode_len, ignore_eos=True)
        self.next_id = 0
        self.live = 0
        self.ramp = 0
        self.req = {}                # request id -> dict(arrival, first, last, n, window_tokens)
        self.step_kinds = {True: [0, 0.0], False: [0, 0.0]}     # prompt_run -> [steps, seconds] in window
        self.first_cohort_live = 0   # requests of the randomised first cohort not finished yet
        self.live_in_window = []     # requests in flight at the start of every timed step
        self.fill_steps = 0
        self.first_cohort = "full"
        self.credit = 1.0            # admissions owed during the ramp (the first request enters at once)
        self.window_debt = {}        # request id -> tokens of steps launched before the timed region
        self.last_was_decode = True  # did the engine's last call schedule a decode step (or nothing)?
        # one length per slot, evenly spaced over 1..L, in a shuffled order
        self.cohort_lengths = [max(1, round(decode_len * (i + 1) / concurrency)) for i in range(concurrency)]
        random.Random(seed + 1).shuffle(self.cohort_lengths)

    def _add(self, now):
        from splitwise_amd import SamplingParams
        rid = str(self.next_id)
        self.next_id += 1
        prompt = self.rng.choices(self._token_range, k=self.prompt_len)      # (one call: the loop's own host time counts)
        params = self.params
        if self.next_id <= self.concurrency:
            # (optional) other lengths for the first cohort; the staggered admission alone spreads the completions
            full = self.params.max_tokens
            if self.first_cohort == "random":
                n = self.rng.randint(1, full)
            elif self.first_cohort == "stratified":
                n = self.cohort_lengths[self.next_id - 1]
            else:
                n = full
            params = SamplingParams(temperature=0.0, max_tokens=n, ignore_eos=True)
        self.engine.add_request(rid, None, params, prompt_token_ids=prompt)
        self.req[rid] = dict(arrival=now, first=None, last=None, n=0, window=0)
        self.live += 1
        if self.next_id <= self.concurrency:
            self.first_cohort_live += 1

    def feed(self):
        """While ramping up: C / L new requests per DECODE step (C requests in flight, L generated tokens
        each) -- the admission rate of the steady state itself, in which every decode step retires C / L
        requests on average.  The first request then finishes just as the C-th is admitted, and from there on
        each finished request is replaced at once: completions, and with them prompt arrivals, keep the even
        spacing of the ramp.  (One admission per engine step, as an earlier version did, packs the early
        admissions more densely than the late ones, and every request lifetime later the loop runs through
        the same dense stretch again.)"""
        now = time.perf_counter()
        if self.ramp < self.concurrency and self.last_was_decode:
            self.credit += self.concurrency / self.params.max_tokens
            while self.credit >= 1.0 and self.ramp < self.concurrency:
                self.credit -= 1.0
                self.ramp += 1
                self._add(now)
        while self.live < min(self.ramp, self.concurrency):
            self._add(now)

    def steady(self):
        """Every slot is taken by a full-length request: the ramp is over and the randomised
        first cohort has drained."""
        return self.ramp >= self.concurrency and self.first_cohort_live == 0

    def fill(self, progress=None):
        """Untimed fill phase: steps until steady() holds (bounded, so a stuck engine cannot
        spin here for ever).  Returns whether steady state was reached."""
        limit = 8 * self.concurrency + 4 * self.params.max_tokens + 200
        while not self.steady() and self.fill_steps < limit:
            self.step(False)
            self.fill_steps += 1
            if progress is not None and self.fill_steps % 50 == 0:
                progress(self.fill_steps, self.live)
        return self.steady()

    def window_stats(self):
        lv = self.live_in_window
        return {"live_min": min(lv) if lv else 0, "live_mean": round(sum(lv) / max(len(lv), 1), 1)}

    def inflight_requests(self):
        """Request ids of the sequences in steps that the engine has launched and not collected yet (pipelined
        engines; empty otherwise).  Each of them gets one token from that step."""
        return self.engine.inflight_request_ids()

    def open_window(self):
        """Call at the start of the timed region, after the device synchronisation.  With pipelined steps the
        first timed call returns the tokens of steps that were launched -- and, because of the synchronisation,
        finished -- BEFORE the region began: those tokens are not the region's work."""
        self.window_debt = {}
        for rid in self.inflight_requests():
            self.window_debt[rid] = self.window_debt.get(rid, 0) + 1

    def close_window(self):
        """Call at the end of the timed region, after the device synchronisation: the steps still in flight ran
        inside the region, so their tokens (one per sequence) are the region's work although no call has
        returned them yet.  Returns their number."""
        n = 0
        for rid in self.inflight_requests():
            r = self.req.get(rid)
            if r is not None:
                r["window"] += 1
                n += 1
        return n

    def step(self, in_window):
        self.feed()
        if in_window:
            self.live_in_window.append(self.live)
        t_begin = time.perf_counter()
        had_waiting = bool(self.engine.scheduler.waiting)
        outs = self.engine.step()
        now = time.perf_counter()
        self.last_was_decode = not getattr(self.engine, "last_scheduled_prompt_run", False)
        if in_window:
            # the kind of step the engine scheduled in this call (split pools schedule prompt and decode steps
            # independently: there a call counts as a prompt step when prompts were waiting)
            kind = self.step_kinds[bool(getattr(self.engine, "last_scheduled_prompt_run", had_waiting))]
            kind[0] += 1
            kind[1] += now - t_begin
        new_tokens = 0
        for out in outs:
            r = self.req[out.request_id]
            n = len(out.outputs[0].token_ids)
            if n > r["n"]:
                fresh = n - r["n"]
                if in_window:
                    owed = min(fresh, self.window_debt.get(out.request_id, 0))      # launched before the window
                    if owed:
                        self.window_debt[out.request_id] -= owed
                    fresh_in_window = fresh - owed
                    r["window"] += fresh_in_window
                    new_tokens += fresh_in_window
                else:
                    new_tokens += fresh
                r["n"] = n
                if r["first"] is None:
                    r["first"] = now
                r["last"] = now
            if out.finished:
                self.live -= 1
                if int(out.request_id) < self.concurrency:
                    self.first_cohort_live -= 1
        return new_tokens

    def goodput_tokens(self):
        good = total = 0
        met = seen = 0
        ttfts, tpots = [], []
        for r in self.req.values():
            if r["window"] == 0 or r["first"] is None:
                continue
            ttft = r["first"] - r["arrival"]
            tpot = (r["last"] - r["first"]) / (r["n"] - 1) if r["n"] > 1 else 0.0
            ok = ttft <= SLO_TTFT and tpot <= SLO_TPOT
            total += r["window"]
            good += r["window"] if ok else 0
            seen += 1
            met += ok
            ttfts.append(ttft)
            tpots.append(tpot)
        return good, total, met, seen, ttfts, tpots


def measure_attention_roofline(engine, samples):
    """Average launch time of the decode paged-attention kernel over the sampled decode batches
    (block tables / context lengths taken from the timed region), timed with HIP events on the
    stream the kernel is launched on."""
    from splitwise_amd.model_executor.layers.attention import PagedAttentionImpl
    worker = engine.model_executor.driver_worker
    cfg = engine.model_config
    par = engine.parallel_config
    h = cfg.hf_config["num_attention_heads"] // par.tensor_parallel_size
    kvh = cfg.get_num_kv_heads(par)
    d = cfg.get_head_size()
    es = torch.tensor([], dtype=worker.gpu_cache[0][0].dtype).element_size()
    scale = d ** -0.5
    dev = worker.device
    stream = torch.cuda.current_stream(dev)
    total_ms, launches, alg_bytes = 0.0, 0, 0.0
    layers = worker.gpu_cache
    for md in samples:
        n = md.context_lens.numel()
        q = torch.empty(n, h, d, dtype=cfg.dtype, device=dev).uniform_(-scale, scale)
        ctx_sum = int(md.context_lens.sum().item())
        per_launch = 2.0 * ctx_sum * kvh * d * es + 2.0 * n * h * d * q.element_size() + md.block_tables.numel() * 4
        if launches == 0:                          # warm the code path once, before the first timed launch
            for kc, vc in layers[:2]:
                PagedAttentionImpl.forward_decode(q, kc, vc, md, kvh, scale, None)
            torch.cuda.synchronize(dev)
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record(stream)
        for kc, vc in layers:                      # one launch per layer, each over its own cache
            PagedAttentionImpl.forward_decode(q, kc, vc, md, kvh, scale, None)
        end.record(stream)
        end.synchronize()
        total_ms += start.elapsed_time(end)
        launches += len(layers)
        alg_bytes += per_launch * len(layers)
    if launches == 0:
        return None
    avg_s = total_ms * 1e-3 / launches
    achieved = alg_bytes / launches / avg_s / 1e9
    out = {"bound": "hbm", "kernel": "paged_attention_kernel (decode)", "achieved": round(achieved, 1),
           "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
           "avg_launch_us": round(avg_s * 1e6, 2), "launches_timed": launches,
           "alg_bytes_per_launch": int(alg_bytes / launches)}
    out.update(pmc_traffic(out["alg_bytes_per_launch"]))
    return out


def pmc_traffic(alg_bytes_per_launch):
    """HBM bytes per launch of the same kernel from hardware counters.  Counters cannot be read from
    inside this process, so `traffic` is null in an ordinary run.  tools/pmc_traffic.sh runs this very
    command under rocprofv3 --pmc (one pass per counter), averages the last `launches_timed` dispatches
    of the kernel -- the timed launches above -- into a summary and re-runs the command with
    SPLITWISE_PMC_SUMMARY pointing at it: only then, and only when the summary was collected on the
    same algorithmic bytes per launch (5 %), is `traffic` filled, with its source beside it."""
    path = os.environ.get("SPLITWISE_PMC_SUMMARY")
    if not path:
        return {}
    try:
        with open(path) as f:
            pmc = json.load(f)
        if abs(int(pmc["alg_bytes_per_launch"]) - alg_bytes_per_launch) > 0.05 * alg_bytes_per_launch:
            return {"traffic_note": "offline counter summary belongs to another workload; ignored"}
        return {"traffic": int(pmc["traffic_bytes_per_launch"]), "traffic_unit": "bytes/launch",
                "traffic_source": "offline rocprofv3 --pmc passes of this command: " + os.path.basename(path),
                "traffic_alg_bytes_per_launch": int(pmc["alg_bytes_per_launch"])}
    except (OSError, KeyError, ValueError):
        return {}


def cpu_baseline(model_name, prompt_len, decode_len):
    """The CPU oracle on a bounded sample of the same workload, scaled to tok/s."""
    from oracle import serving as oserving
    threads = torch.get_num_threads()
    # up to 8 layers (a quarter of the 7B model), four requests: ~10 s of host work, scaled by layers (they are identical)
    geo_layers = oserving._GEOMETRY[model_name]["layers"]
    res = oserving.timed_sample(model_name, prompt_len=prompt_len, decode_len=decode_len, layers=min(8, max(2, geo_layers // 4)),
                                num_requests=4, decode_steps=8, budget_s=25.0)
    return {"value": round(res["tokens_per_s"], 3), "unit": "tok/s", "cores": threads, "kind": "port",
            "sample": res["sample"]}


def main():
    args = parse_args()
    if args.replica is not None:                     # one replica of a multi-GPU leg, started by the launcher
        from splitwise_amd.executor import bench_replica
        return bench_replica.run(json.loads(args.replica))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1:
        args.gpus = max(args.gpus, world)
        from splitwise_amd.executor import bench_launch
        sys.exit(bench_launch.main(args))

    args.overlap_prompt_decode = not (args.plain_engine or args.no_pipeline_steps)

    def run_single() -> dict:
        """The timed run itself; everything it builds (engine, cache, graphs) dies with its frame."""
        from splitwise_amd import EngineArgs, LLMEngine
        torch.cuda.set_device(0)
        conc = args.concurrency
        warmup = args.warmup
        seq_blocks = -(-(args.prompt_len + args.decode_len + 1) // 16)
        dtype = args.dtype or ("float16" if args.quantization else "bfloat16")
        dtype_tag = {"bfloat16": "bf16", "float16": "f16", "half": "f16", "float32": "f32"}.get(dtype, dtype)
        variant = dtype_tag + (f" {args.quantization} int4" if args.quantization else "") + \
            (" fp8 KV cache" if args.kv_cache_dtype != "auto" else "")
        engine = LLMEngine.from_engine_args(EngineArgs(
            model=args.model, dtype=dtype, quantization=args.quantization, kv_cache_dtype=args.kv_cache_dtype,
            max_num_seqs=conc, max_model_len=2048, max_num_batched_tokens=args.max_num_batched_tokens,
            num_gpu_blocks_override=int(conc * seq_blocks * 1.25) + 64, swap_space=0.25,
            enforce_eager=args.enforce_eager, max_context_len_to_capture=1024,
            prefill_batch_tokens=args.prefill_batch_tokens, prefill_max_wait=args.prefill_max_wait,
            pipeline_steps=not args.no_pipeline_steps, overlap_prompt_decode=args.overlap_prompt_decode))
        loop = ClosedLoop(engine, conc, args.prompt_len, args.decode_len, seed=0)
        loop.first_cohort = args.first_cohort

        # decode batches of the timed region: their sizes (all of them) and, for the roofline
        # measurement, the block tables and context lengths of up to 12 spread over the region
        runner = engine.model_executor.driver_worker.model_runner
        samples = []
        batch_sizes = []
        orig_prepare = runner._prepare_decode
        state = {"on": False, "count": 0}
        stride = max(1, args.steps // 24)

        def sampling_prepare(metas):
            out = orig_prepare(metas)
            if state["on"]:
                n = sum(len(m.seq_data) for m in metas)
                batch_sizes.append(n)
                if state["count"] % stride == 0 and len(samples) < 12:
                    import copy
                    md = out[2]
                    keep = copy.copy(md)
                    keep.context_lens = md.context_lens[:n].clone()
                    keep.block_tables = md.block_tables[:n].clone()
                    keep.max_context_len = md.max_context_len
                    samples.append(keep)
                state["count"] += 1
            return out

        runner._prepare_decode = sampling_prepare

        steady = True
        if not args.no_fill:
            steady = loop.fill()
        for _ in range(warmup):
            loop.step(False)
        torch.cuda.synchronize()
        state["on"] = True
        loop.open_window()
        t0 = time.perf_counter()
        tokens = 0
        trace = []                                   # (seconds since t0, tokens so far) after every timed step
        for _ in range(args.steps):
            tokens += loop.step(True)
            trace.append((time.perf_counter() - t0, tokens, bool(getattr(engine, "last_scheduled_prompt_run", False))))
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        tokens += loop.close_window()
        if args.window_trace:
            # generated tok/s of every run of `window_trace` consecutive steps: how much a short window's figure
            # depends on where in the request cycle it falls
            w = args.window_trace
            rates = []
            for i in range(w, len(trace) + 1, w):
                t_a, n_a = trace[i - w - 1][:2] if i > w else (0.0, 0)
                t_b, n_b = trace[i - 1][:2]
                rates.append((n_b - n_a) / (t_b - t_a))
            print(f"tok/s per {w}-step window: " + " ".join(f"{r:.0f}" for r in rates) +
                  f" | min {min(rates):.0f} mean {sum(rates) / len(rates):.0f} max {max(rates):.0f}", file=sys.stderr, flush=True)
            if len(trace) <= 60:                     # every step: kind scheduled in the call, tokens returned, host time
                prev_t, prev_n = 0.0, 0
                parts = []
                for t, n, prompt in trace:
                    parts.append(f"{'P' if prompt else 'D'}{n - prev_n}:{(t - prev_t) * 1e3:.0f}ms")
                    prev_t, prev_n = t, n
                print("steps: " + " ".join(parts), file=sys.stderr, flush=True)
        state["on"] = False

        good, total, met, seen, ttfts, tpots = loop.goodput_tokens()
        win = loop.window_stats()
        in_flight = win["live_min"] if win["live_min"] >= 0.95 * conc else win["live_mean"]
        state_tag = "steady state" if (steady and not args.no_fill and win["live_min"] >= 0.95 * conc) else \
            "RAMP (not steady state)"
        result = {
            "metric": "goodput (tok/s under TTFT+TBT SLO), " + ("Llama-2-7B" if args.model == "llama-2-7b" else args.model) + " co-located", "value": round(good / elapsed, 2),
            "unit": "tok/s", "n_gpus": 1, "steps": args.steps, "warmup": warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype_tag, "data": "synthetic",
            "config": {"workload": f"{args.model} {variant} co-located on 1 GPU, closed loop, {state_tag}: {in_flight:g} requests "
                                   f"in flight over the timed steps ({conc} asked for), "
                                   f"prefill {args.prompt_len} / decode {args.decode_len} tokens, random-init weights, "
                                   f"SLO TTFT<={SLO_TTFT}s TPOT<={SLO_TPOT}s"
                                   + ("; prompt steps run BESIDE the decode steps on a second stream (every timed step "
                                      "is a decode step; a call counts as a prompt step when prompts were waiting)"
                                      if args.overlap_prompt_decode else ""),
                       "concurrency": in_flight, "concurrency_target": conc,
                       "live_min": win["live_min"], "live_mean": win["live_mean"],
                       "decode_batch_min": min(batch_sizes) if batch_sizes else 0,
                       "decode_batch_mean": round(sum(batch_sizes) / max(len(batch_sizes), 1), 1),
                       "fill_steps": loop.fill_steps, "steady_state": state_tag == "steady state",
                       "prompt_len": args.prompt_len, "decode_len": args.decode_len,
                       "prefill_batch_tokens": args.prefill_batch_tokens, "prefill_max_wait_s": args.prefill_max_wait,
                       "hip_graphs": not args.enforce_eager, "pipeline_steps": not args.no_pipeline_steps,
                       "overlap_prompt_decode": args.overlap_prompt_decode},
            "throughput_tok_s": round(total / elapsed, 2),
            "slo_attainment": round(met / max(seen, 1), 4),
            "prompt_steps": loop.step_kinds[True][0],
            "avg_prompt_step_ms": round(loop.step_kinds[True][1] / max(loop.step_kinds[True][0], 1) * 1e3, 2),
            "decode_steps": loop.step_kinds[False][0],
            "avg_decode_step_ms": round(loop.step_kinds[False][1] / max(loop.step_kinds[False][0], 1) * 1e3, 2),
            "ttft_p50_ms": round(sorted(ttfts)[len(ttfts) // 2] * 1e3, 2) if ttfts else None,
            "ttft_p99_ms": round(sorted(ttfts)[int(len(ttfts) * 0.99)] * 1e3, 2) if ttfts else None,
            "tpot_p50_ms": round(sorted(tpots)[len(tpots) // 2] * 1e3, 2) if tpots else None,
            "tpot_p99_ms": round(sorted(tpots)[int(len(tpots) * 0.99)] * 1e3, 2) if tpots else None,
        }
        if not args.no_roofline:
            result["roofline"] = measure_attention_roofline(engine, samples)
            if args.overlap_prompt_decode and result["roofline"]:
                result["roofline"]["note"] = ("launch durations of the kernel running alone; inside this run's steps it "
                                              "shares the GPU with a prompt step's kernels")
        return result

    result = run_single()
    import gc
    gc.unfreeze()                   # (the engine parks its long-lived objects out of the collector's reach)
    gc.collect()
    torch.cuda.empty_cache()        # the cache and the weights go back: the overlap leg's child needs the room
    print(f"after the timed run: {torch.cuda.mem_get_info()[0] / 2**30:.1f} GiB of device memory free, "
          f"{torch.cuda.memory_allocated() / 2**30:.2f} GiB still allocated by this process, "
          f"{torch.cuda.memory_reserved() / 2**30:.2f} GiB reserved", file=sys.stderr, flush=True)
    if not (args.no_overlap_leg or args.no_fill or args.no_pipeline_steps):
        # the same workload on the other engine, beside the headline
        result["plain_engine" if args.overlap_prompt_decode else "overlap_prompt_decode"] = \
            other_engine_leg(args, result["value"])
    if not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.model, args.prompt_len, args.decode_len)
    print(json.dumps(result))


def other_engine_leg(args, this_value: float) -> dict:
    """The same workload once more on the other engine -- prompt and decode steps taking turns when this run had
    them side by side, and the other way round -- in a child process of its own.  Whatever happens to the child,
    the line of this process is printed."""
    import subprocess
    preload = os.environ.get("LD_PRELOAD", "")
    if "rocprof" in preload or "ROCP_TOOL_LIBRARIES" in os.environ:
        return {"skipped": "under a profiler: a second engine's kernels would land in the same trace"}
    other = "--plain-engine" if args.overlap_prompt_decode else "--overlap-prompt-decode"
    cmd = [sys.executable, os.path.abspath(__file__), other, "--no-overlap-leg", "--no-cpu-baseline", "--no-roofline",
           "--steps", str(args.steps), "--warmup", str(args.warmup), "--concurrency", str(args.concurrency),
           "--model", args.model, "--prompt-len", str(args.prompt_len), "--decode-len", str(args.decode_len),
           "--max-num-batched-tokens", str(args.max_num_batched_tokens),
           "--prefill-batch-tokens", str(args.prefill_batch_tokens), "--prefill-max-wait", str(args.prefill_max_wait),
           "--kv-cache-dtype", args.kv_cache_dtype, "--first-cohort", args.first_cohort]
    if args.quantization:
        cmd += ["--quantization", args.quantization]
    if args.dtype:
        cmd += ["--dtype", args.dtype]
    if args.enforce_eager:
        cmd += ["--enforce-eager"]
    try:
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=420)
        lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
        if proc.returncode != 0 or not lines:
            return {"error": f"exit status {proc.returncode}: {proc.stderr[-400:]}"}
        line = json.loads(lines[-1])
        keep = ("value", "ms_per_step", "slo_attainment", "ttft_p50_ms", "ttft_p99_ms", "tpot_p50_ms", "tpot_p99_ms")
        out = {k: line.get(k) for k in keep}
        out["steady_state"] = line["config"].get("steady_state")
        out["vs_headline"] = round(line["value"] / max(this_value, 1e-9), 4)
        out["what"] = (("prompt steps and decode steps taking turns (the reference's scheduling; pipelined steps and "
                        "prefill batching as in the headline run)" if args.overlap_prompt_decode else
                        "prompt steps beside the decode steps on a second stream of the same GPU "
                        "(EngineArgs.overlap_prompt_decode)") + ", same workload and SLO, timed in a child process")
        return out
    except (subprocess.TimeoutExpired, OSError, ValueError, KeyError) as exc:
        return {"error": repr(exc)[:400]}


if __name__ == "__main__":
    main()
