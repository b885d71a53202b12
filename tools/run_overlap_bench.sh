#!/bin/bash
# The N=1 bench with prompt and decode steps on two streams, beside the default: tools/run_overlap_bench.sh STEPS
mkdir -p gpurun_out
(while sleep 30; do echo tick; done) &
T=$!
timeout -k 10 500 python bench.py --steps $1 --warmup 20 --no-cpu-baseline --overlap-prompt-decode > gpurun_out/bench_overlap.json 2> gpurun_out/bench_overlap.err
rc=$?
tail -n 5 gpurun_out/bench_overlap.err
timeout -k 10 500 python bench.py --steps $1 --warmup 20 --no-cpu-baseline > gpurun_out/bench_plain.json 2> gpurun_out/bench_plain.err
kill $T
python - <<'P'
import json
for n in ("overlap", "plain"):
    try:
        d = json.load(open(f"gpurun_out/bench_{n}.json"))
        print(n, d["value"], d["ms_per_step"], d.get("slo_attainment"), d.get("ttft_p50_ms"), d.get("tpot_p50_ms"), d["roofline"] and d["roofline"]["avg_launch_us"], d["config"].get("live_min"))
    except Exception as e:
        print(n, "failed", e)
P
exit $rc
