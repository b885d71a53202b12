#!/bin/bash
# Two-turn conversations with and without the prefix cache (tools/bench_next_turns.py), each mode in its own process.
mkdir -p gpurun_out
(while sleep 30; do echo tick; done) &
T=$!
trap "kill $T" EXIT
timeout -k 10 120 python tools/bench_next_turns.py --model llama-tiny --conversations 12 --prompt-len 100 --decode-len 40 \
  --new-tokens 9 --decode-len-2 20 --prefix-cache > gpurun_out/next_turns_tiny.json 2> gpurun_out/next_turns_tiny.err || { tail -n 20 gpurun_out/next_turns_tiny.err; exit 1; }
cat gpurun_out/next_turns_tiny.json
: > gpurun_out/next_turns.json
for mode in "" "--prefix-cache"; do
  timeout -k 10 400 python tools/bench_next_turns.py $mode "$@" >> gpurun_out/next_turns.json 2> gpurun_out/next_turns.err || { tail -n 20 gpurun_out/next_turns.err; exit 1; }
  tail -n 1 gpurun_out/next_turns.json
done
