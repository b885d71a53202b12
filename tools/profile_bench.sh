#!/bin/bash
# The driver's bench command under rocprofv3 --kernel-trace --stats: the JSON line, the per-kernel summary and the
# averages over the timed window, all under gpurun_out/prof_latest/ (extra arguments go to bench.py).
set -e
R=$PWD
O=$R/gpurun_out/prof_latest
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(while sleep 30; do echo tick; done) &
T=$!
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 $R/bench.py --steps 20 --warmup 5 "$@" > $O/bench_under_rocprof.log 2> $O/rocprof.err || { kill $T; tail -n 20 $O/rocprof.err; exit 1; }
kill $T
cd $R
trap 'rm -rf $O/trace' EXIT
find $O/trace -type f | head -n 20
python3 tools/window_kernel_avg.py $O/trace 556 > $O/bench_window_kernel_avg.log
S=$(find $O/trace -name "*kernel_stats.csv" | head -n 1)
head -n 40 "$S" > $O/bench_kernel_stats.csv
rm -rf $O/trace
tail -n 1 $O/bench_under_rocprof.log | cut -c1-400
head -n 8 $O/bench_window_kernel_avg.log
