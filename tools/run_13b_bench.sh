#!/bin/bash
# 13B GPTQ int4 + fp8 cache on one GPU, 600 steps (the run behind profiles/*/bench_13b_gptq_fp8_*.json).
mkdir -p gpurun_out
(while sleep 30; do echo tick; done) &
T=$!
timeout -k 10 900 python bench.py --model llama-2-13b --quantization gptq --kv-cache-dtype fp8_e5m2 --steps 600 --warmup 20 \
  > gpurun_out/bench_13b_gptq_fp8_r3b.json 2> gpurun_out/bench_13b_gptq_fp8_r3b.err
rc=$?
kill $T
tail -c 300 gpurun_out/bench_13b_gptq_fp8_r3b.err
python - <<'P'
import json
d = json.load(open("gpurun_out/bench_13b_gptq_fp8_r3b.json"))
print(d["value"], d["ms_per_step"], d["roofline"])
P
exit $rc
