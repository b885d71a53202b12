#!/bin/bash
# More runs of the overlap engine: a 4000-step soak of the 7B bench and the 13B GPTQ + fp8 configuration.
mkdir -p gpurun_out
(while sleep 30; do echo tick; done) &
T=$!
timeout -k 10 600 python bench.py --steps 4000 --warmup 20 --no-cpu-baseline --overlap-prompt-decode > gpurun_out/bench_overlap_4000.json 2> gpurun_out/bench_overlap_4000.err
timeout -k 10 600 python bench.py --model llama-2-13b --quantization gptq --kv-cache-dtype fp8_e5m2 --steps 600 --warmup 20 --no-cpu-baseline --overlap-prompt-decode > gpurun_out/bench_overlap_13b.json 2> gpurun_out/bench_overlap_13b.err
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --overlap-prompt-decode > gpurun_out/bench_overlap_20.json 2> gpurun_out/bench_overlap_20.err
kill $T
python - <<'P'
import json
for n in ("overlap_4000", "overlap_13b", "overlap_20"):
    try:
        d = json.load(open(f"gpurun_out/bench_{n}.json"))
        print(n, d["value"], d["ms_per_step"], d.get("slo_attainment"), d.get("ttft_p50_ms"), d.get("ttft_p99_ms"), d.get("tpot_p50_ms"), d["config"].get("live_min"), d["config"].get("steady_state"))
    except Exception as e:
        print(n, "failed", e, open(f"gpurun_out/bench_{n}.err").read()[-600:])
P
