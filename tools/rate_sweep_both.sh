#!/bin/bash
# Open-loop rate sweep of the plain and the overlap engine on the same box.
mkdir -p gpurun_out
(while sleep 30; do echo tick; done) &
T=$!
timeout -k 10 500 python -u tools/bench_rates.py --rates 8,16,24,28,32,36,40 --seconds 10 > gpurun_out/rate_sweep_plain.log 2> gpurun_out/rate_sweep_plain.err
timeout -k 10 500 python -u tools/bench_rates.py --rates 8,16,24,28,32,36,40 --seconds 10 --overlap-prompt-decode > gpurun_out/rate_sweep_overlap.log 2> gpurun_out/rate_sweep_overlap.err
kill $T
python - <<'P'
import json
for n in ("plain", "overlap"):
    for ln in open(f"gpurun_out/rate_sweep_{n}.log"):
        if ln.startswith("{"):
            d = json.loads(ln)
            print(n, {k: d[k] for k in d if k != "config"})
P
