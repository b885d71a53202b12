#!/usr/bin/env python3
"""Two-turn conversations on one GPU, with and without the prefix cache.

Turn 1: a prompt of --prompt-len tokens, --decode-len generated.  Turn 2 of the same conversation: turn 1's prompt,
its answer and --new-tokens more as the prompt, --decode-len-2 generated.  Turn 1 arrives at a fixed rate; a
conversation's turn 2 is sent --think-time seconds after its turn 1 has finished, while other conversations are
still in their first turn.  With the prefix cache the second prompt finds the first prompt's blocks AND the blocks the
decode steps filled with the answer, and computes only its tail.  Reported per mode: mean / p90 time to the first
token of both turns, and the tokens per second of the whole run.

  python tools/bench_next_turns.py [--prefix-cache] [--conversations 96] [--rate 6] [--plain-engine]
One JSON line on stdout."""
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


def run(engine, args, first_id):
    from splitwise_amd import SamplingParams
    rng = np.random.default_rng(first_id)
    vocab = engine.model_config.get_vocab_size()
    n = args.conversations
    arrivals = np.arange(n) / args.rate
    first = SamplingParams(temperature=0.0, max_tokens=args.decode_len, ignore_eos=True)
    second = SamplingParams(temperature=0.0, max_tokens=args.decode_len_2, ignore_eos=True)
    prompts, req, due = {}, {}, []                # due: (time, conversation) of second turns
    sent = done = 0
    t0 = time.perf_counter()
    while done < 2 * n:
        now = time.perf_counter() - t0
        while sent < n and arrivals[sent] <= now:
            rid = f"{first_id + sent}.1"
            prompts[rid] = rng.integers(0, vocab, args.prompt_len).tolist()
            engine.add_request(rid, None, first, prompt_token_ids=prompts[rid])
            req[rid] = dict(arrival=t0 + arrivals[sent], first=None, n=0, turn=1, prompt=args.prompt_len)
            sent += 1
        for item in [d for d in due if d[0] <= now]:
            due.remove(item)
            when, rid, prompt = item
            engine.add_request(rid, None, second, prompt_token_ids=prompt)
            req[rid] = dict(arrival=t0 + when, first=None, n=0, turn=2, prompt=len(prompt))
        if not engine.has_unfinished_requests():
            time.sleep(0.001)
            continue
        outs = engine.step()
        stamp = time.perf_counter()
        for out in outs:
            r = req[out.request_id]
            tokens = out.outputs[0].token_ids
            if len(tokens) > r["n"]:
                r["n"] = len(tokens)
                if r["first"] is None:
                    r["first"] = stamp
            if out.finished:
                done += 1
                if r["turn"] == 1:
                    conv = out.request_id[:-2]
                    turn = prompts.pop(out.request_id) + list(tokens) + rng.integers(0, vocab, args.new_tokens).tolist()
                    due.append((stamp - t0 + args.think_time, conv + ".2", turn))
    makespan = time.perf_counter() - t0
    res = {}
    for turn in (1, 2):
        ttft = np.array([r["first"] - r["arrival"] for r in req.values() if r["turn"] == turn])
        res[f"turn{turn}_ttft_mean_ms"] = round(float(ttft.mean()) * 1e3, 1)
        res[f"turn{turn}_ttft_p90_ms"] = round(float(np.quantile(ttft, 0.9)) * 1e3, 1)
    res["turn2_prompt_tokens"] = int(np.mean([r["prompt"] for r in req.values() if r["turn"] == 2]))
    res["total_tok_s"] = round(sum(r["n"] + r["prompt"] for r in req.values()) / makespan, 1)
    res["out_tok_s"] = round(sum(r["n"] for r in req.values()) / makespan, 1)
    res["makespan_s"] = round(makespan, 2)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--conversations", type=int, default=96)
    ap.add_argument("--rate", type=float, default=6.0, help="first turns per second")
    ap.add_argument("--think-time", type=float, default=1.0)
    ap.add_argument("--model", default="llama-2-7b")
    ap.add_argument("--prompt-len", type=int, default=772)
    ap.add_argument("--decode-len", type=int, default=170)
    ap.add_argument("--new-tokens", type=int, default=60)
    ap.add_argument("--decode-len-2", type=int, default=170)
    ap.add_argument("--max-num-seqs", type=int, default=128)
    ap.add_argument("--prefix-cache", action="store_true")
    ap.add_argument("--plain-engine", action="store_true", help="prompt steps and decode steps taking turns")
    args = ap.parse_args()

    from splitwise_amd import EngineArgs, LLMEngine
    torch.cuda.set_device(0)
    longest = args.prompt_len + args.decode_len + args.new_tokens + args.decode_len_2 + 1
    engine = LLMEngine.from_engine_args(EngineArgs(
        model=args.model, dtype="bfloat16", max_num_seqs=args.max_num_seqs, max_model_len=2048,
        max_num_batched_tokens=4096, num_gpu_blocks_override=int(args.max_num_seqs * -(-longest // 16) * 1.5) + 64,
        swap_space=0.25, max_context_len_to_capture=2048, prefill_batch_tokens=2300, prefill_max_wait=0.08,
        pipeline_steps=True, overlap_prompt_decode=not args.plain_engine, enable_prefix_caching=args.prefix_cache))
    warm = argparse.Namespace(**{**vars(args), "conversations": 8, "rate": 8.0, "think_time": 0.1})
    run(engine, warm, 10 ** 6)
    res = run(engine, args, 0)
    res["prefix_cache"] = args.prefix_cache
    res["config"] = (f"{args.model} bfloat16, 1 GPU, {args.conversations} conversations at {args.rate}/s: "
                     f"{args.prompt_len} + {args.decode_len} tokens, then the whole turn + {args.new_tokens} new "
                     f"tokens + {args.decode_len_2} generated" +
                     (", prompt steps beside decode steps" if not args.plain_engine else ""))
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
