#!/bin/bash
# Longer runs of the N=1 bench for the record: tools/run_long_bench.sh STEPS NAME
mkdir -p gpurun_out
(while sleep 30; do echo tick; done) &
T=$!
timeout -k 10 1000 python bench.py --steps $1 --warmup 20 > gpurun_out/$2.json 2> gpurun_out/$2.err
rc=$?
kill $T
tail -c 300 gpurun_out/$2.err
python - <<P
import json
d = json.load(open("gpurun_out/$2.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"], d.get("slo_attainment"))
P
exit $rc
