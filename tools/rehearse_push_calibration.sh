#!/bin/bash
# One-GPU rehearsal of the pairs leg with the push workgroup cap measured at set-up (forced for the local peer).
mkdir -p gpurun_out/reh
(while sleep 30; do echo tick; done) &
T=$!
SPLITWISE_KV_PUSH_CALIBRATE=1 SPLITWISE_BENCH_ONE_GPU=1 SPLITWISE_BENCH_LOG_DIR=gpurun_out/reh \
  timeout -k 10 500 python bench.py --gpus 2 --legs pairs --concurrency 64 --steps 40 --warmup 5 \
  > gpurun_out/reh/n2_calibrated.json 2> gpurun_out/reh/n2_calibrated.err
rc=$?
kill $T
tail -c 400 gpurun_out/reh/n2_calibrated.err
python - <<'P'
import json
d = json.load(open("gpurun_out/reh/n2_calibrated.json"))
print(d["value"], d["kv_migration"])
P
exit $rc
