#!/bin/bash
# One-GPU rehearsal of the launcher's fallback: the pairs leg with the KV blocks moved by the HIP runtime's copies.
mkdir -p gpurun_out/reh
(while sleep 30; do echo tick; done) &
T=$!
SPLITWISE_KV_TRANSPORT=runtime SPLITWISE_BENCH_ONE_GPU=1 SPLITWISE_BENCH_LOG_DIR=gpurun_out/reh \
  timeout -k 10 500 python bench.py --gpus 2 --legs pairs --concurrency 64 --steps 40 --warmup 5 \
  > gpurun_out/reh/n2_runtime.json 2> gpurun_out/reh/n2_runtime.err
rc=$?
kill $T
tail -c 600 gpurun_out/reh/n2_runtime.err
head -c 1200 gpurun_out/reh/n2_runtime.json
exit $rc
