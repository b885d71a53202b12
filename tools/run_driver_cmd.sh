#!/bin/bash
# The driver's N=1 command as it runs it, with the wall time of the whole process.
mkdir -p gpurun_out
(while sleep 30; do echo tick; done) &
T=$!
t0=$(date +%s)
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_driver_final.json 2> gpurun_out/bench_driver_final.err
rc=$?
t1=$(date +%s)
kill $T
echo "wall $((t1 - t0)) s, exit $rc"
tail -n 3 gpurun_out/bench_driver_final.err
python - <<'P'
import json
d = json.load(open("gpurun_out/bench_driver_final.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
print(d.get("plain_engine") or d.get("overlap_prompt_decode"))
P
exit $rc
