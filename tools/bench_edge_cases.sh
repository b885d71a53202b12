#!/bin/bash
# Unusual ways to call the N=1 bench: very few steps, the reference's loop, the plain engine, a small concurrency.
mkdir -p gpurun_out/edge
(while sleep 30; do echo tick; done) &
T=$!
run() {
  timeout -k 10 400 python bench.py $2 --no-cpu-baseline > gpurun_out/edge/$1.json 2> gpurun_out/edge/$1.err; rc=$?
  python - <<P
import json
try:
    d = json.load(open("gpurun_out/edge/$1.json"))
    other = d.get("plain_engine") or d.get("overlap_prompt_decode") or {}
    print("$1", "rc=$rc", d["value"], d["steps"], d["config"]["overlap_prompt_decode"], d["config"]["steady_state"], "other:", other.get("value"), other.get("error"), other.get("skipped"))
except Exception as e:
    print("$1", "rc=$rc", "no line", e)
P
}
run few "--gpus 1 --steps 5 --warmup 2"
run refloop "--steps 40 --warmup 5 --no-pipeline-steps --prefill-batch-tokens 0"
run plain "--steps 40 --warmup 5 --plain-engine"
run c32 "--steps 40 --warmup 5 --concurrency 32"
kill $T
