#!/usr/bin/env python3
"""Open-loop serving benchmark: Poisson arrivals at a sweep of request rates, one GPU, co-located.

The reference's client (benchmarks/benchmark_serving_distserve.py:44-55,176-177) draws the gaps
between requests from gamma(shape 1, scale 1/rate) with numpy seed 0 and reports, per request,
ftl = first token - send and tpot = (last - first) / (out_len - 1) (benchmark_utils.py:139-141);
its plots count the share of requests inside TTFT <= 0.25 s and TPOT <= 0.1 s and call a rate
sustainable while that share is >= 90 %.  This drives the engine in-process with the same
arrival law and the same accounting (one monotonic clock for send and token stamps), synthetic
prompts of the mean shape of the reference's logs (772 prompt / 170 generated tokens).

  python tools/bench_rates.py [--rates 2,4,8,...] [--seconds 12]
One JSON line per rate on stdout."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "vllm-splitwise_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

SLO_TTFT, SLO_TPOT = 0.25, 0.10


def run_rate(engine, rate, num_requests, prompt_len, decode_len, temperature, first_id):
    from splitwise_amd import SamplingParams
    rng = np.random.default_rng(first_id)
    vocab = engine.model_config.get_vocab_size()
    gaps = np.random.gamma(shape=1.0, scale=1.0 / rate, size=num_requests)
    arrivals = np.cumsum(gaps) - gaps[0]
    params = SamplingParams(temperature=temperature, max_tokens=decode_len, ignore_eos=True)
    req = {}
    sent = done = 0
    t0 = time.perf_counter()
    while done < num_requests:
        now = time.perf_counter() - t0
        while sent < num_requests and arrivals[sent] <= now:
            rid = str(first_id + sent)
            prompt = rng.integers(0, vocab, prompt_len).tolist()
            engine.add_request(rid, None, params, prompt_token_ids=prompt)
            # a request counts from the moment it was due, not from when this loop got to it
            req[rid] = dict(arrival=t0 + arrivals[sent], first=None, last=None, n=0)
            sent += 1
        if not engine.has_unfinished_requests():
            time.sleep(max(0.0, min(0.002, arrivals[sent] - now)))
            continue
        outs = engine.step()
        stamp = time.perf_counter()
        for out in outs:
            r = req[out.request_id]
            n = len(out.outputs[0].token_ids)
            if n > r["n"]:
                r["n"] = n
                if r["first"] is None:
                    r["first"] = stamp
                r["last"] = stamp
            if out.finished:
                done += 1
    makespan = time.perf_counter() - t0
    ttft = np.array([r["first"] - r["arrival"] for r in req.values()])
    tpot = np.array([(r["last"] - r["first"]) / max(r["n"] - 1, 1) for r in req.values()])
    toks = np.array([r["n"] for r in req.values()])
    ok = (ttft <= SLO_TTFT) & (tpot <= SLO_TPOT)
    return {"request_rate": rate, "requests": num_requests, "achieved_rate": round(num_requests / makespan, 2),
            "slo_attainment": round(float(ok.mean()), 4), "goodput_tok_s": round(float(toks[ok].sum()) / makespan, 1),
            "out_tok_s": round(float(toks.sum()) / makespan, 1),
            "total_tok_s": round(float(toks.sum() + prompt_len * num_requests) / makespan, 1),
            "ttft_mean_ms": round(float(ttft.mean()) * 1e3, 1), "ttft_p90_ms": round(float(np.quantile(ttft, 0.9)) * 1e3, 1),
            "tpot_mean_ms": round(float(tpot.mean()) * 1e3, 1), "tpot_p90_ms": round(float(np.quantile(tpot, 0.9)) * 1e3, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rates", default="2,4,8,12,16,20,24,28,32,36")
    ap.add_argument("--seconds", type=float, default=12.0, help="arrival window per rate (at least 100 requests)")
    ap.add_argument("--model", default="llama-2-7b")
    ap.add_argument("--prompt-len", type=int, default=772)
    ap.add_argument("--decode-len", type=int, default=170)
    ap.add_argument("--temperature", type=float, default=1.0)       # the reference client's setting
    ap.add_argument("--max-num-seqs", type=int, default=256)
    ap.add_argument("--prefill-batch-tokens", type=int, default=2300)
    ap.add_argument("--prefill-max-wait", type=float, default=0.08)
    ap.add_argument("--no-pipeline-steps", action="store_true")
    ap.add_argument("--overlap-prompt-decode", action="store_true", help="prompt steps beside decode steps (two streams)")
    ap.add_argument("--quantization", default=None, choices=[None, "gptq", "awq"])
    ap.add_argument("--kv-cache-dtype", default="auto", choices=["auto", "fp8_e5m2"])
    args = ap.parse_args()

    from splitwise_amd import EngineArgs, LLMEngine
    torch.cuda.set_device(0)
    np.random.seed(0)
    seq_blocks = -(-(args.prompt_len + args.decode_len + 1) // 16)
    dtype = "float16" if args.quantization else "bfloat16"
    engine = LLMEngine.from_engine_args(EngineArgs(
        model=args.model, dtype=dtype, quantization=args.quantization, kv_cache_dtype=args.kv_cache_dtype, max_num_seqs=args.max_num_seqs, max_model_len=2048, max_num_batched_tokens=4096,
        num_gpu_blocks_override=int(args.max_num_seqs * seq_blocks * 1.25) + 64, swap_space=0.25,
        max_context_len_to_capture=1024, prefill_batch_tokens=args.prefill_batch_tokens,
        prefill_max_wait=args.prefill_max_wait, pipeline_steps=not args.no_pipeline_steps,
        overlap_prompt_decode=args.overlap_prompt_decode))
    run_rate(engine, 8.0, 32, args.prompt_len, args.decode_len, args.temperature, 10 ** 6)      # warm-up
    first_id = 0
    for rate in [float(r) for r in args.rates.split(",")]:
        n = int(min(max(100, rate * args.seconds), 600))
        res = run_rate(engine, rate, n, args.prompt_len, args.decode_len, args.temperature, first_id)
        first_id += n
        res["config"] = (f"{args.model} {dtype}" + (f" {args.quantization} int4" if args.quantization else "") +
                         (" fp8 KV cache" if args.kv_cache_dtype != "auto" else "") +
                         f" co-located, 1 GPU, prefill {args.prompt_len} / decode {args.decode_len}" +
                         (", prompt steps beside decode steps" if args.overlap_prompt_decode else ""))
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
