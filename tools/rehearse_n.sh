#!/bin/bash
# One-GPU rehearsal of every leg of an N-GPU run: tools/rehearse_n.sh N CONCURRENCY STEPS
N=$1; C=$2; K=$3
mkdir -p gpurun_out/reh
(while sleep 30; do echo tick; done) &
T=$!
SPLITWISE_BENCH_ONE_GPU=1 SPLITWISE_BENCH_LOG_DIR=gpurun_out/reh/n$N \
  timeout -k 10 1100 python bench.py --gpus $N --concurrency $C --steps $K --warmup 5 \
  > gpurun_out/reh/n$N.json 2> gpurun_out/reh/n$N.err
rc=$?
kill $T
tail -c 600 gpurun_out/reh/n$N.err
python - <<P
import json
d = json.load(open("gpurun_out/reh/n$N.json"))
print(d["value"], d.get("error"), d["config"]["headline_leg"])
for k, v in d["legs"].items():
    print(k, v.get("value"), v.get("vs_colocated"), (v.get("error") or v.get("skipped") or "")[:300])
P
exit $rc
