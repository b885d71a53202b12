#!/bin/bash
# The whole GPU suite the way the round's end runs it, output kept under gpurun_out/.
mkdir -p gpurun_out
timeout -k 10 1150 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/gpu_tests_latest.log 2>&1
rc=$?
tail -n 5 gpurun_out/gpu_tests_latest.log
exit $rc
