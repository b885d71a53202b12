#!/usr/bin/env python3
"""Per-kernel launch counts and average durations over the LAST `window_ms` of a rocprofv3 kernel trace of
bench.py -- the timed, steady-state steps -- next to the same over the whole trace (which the `--stats` summary
reports and which mixes in the fill phase's smaller batches).  usage: window_kernel_avg.py <trace dir> [window_ms]"""
import csv
import glob
import sys
from collections import defaultdict

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
window = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 500e6
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the trace ends with the roofline replays and the teardown: anchor the window at the last sampler launch of a
# step instead (the last launch of torch's argmax / multinomial kernels), approximated by the last rms_norm launch
anchor = max(e for s, e, n in rows if "rms_norm" in n)
lo = anchor - window


def family(name):
    if "Cijk" in name:
        return "gemm " + name[name.find("MT"):name.find("MT") + 12]
    return name.split("(")[0].replace("void ", "")[:70]


whole, win = defaultdict(lambda: [0, 0]), defaultdict(lambda: [0, 0])
for s, e, n in rows:
    k = family(n)
    whole[k][0] += 1
    whole[k][1] += e - s
    if lo <= s <= anchor:
        win[k][0] += 1
        win[k][1] += e - s
print(f"window: the last {window / 1e6:.0f} ms before the last model kernel")
print(f"{'kernel':72s} {'window':>18s} {'whole trace':>20s}")
for k, (c, t) in sorted(win.items(), key=lambda kv: -kv[1][1])[:12]:
    wc, wt = whole[k]
    print(f"{k:72s} {c:6d} x {t / c / 1e3:8.1f} us {wc:8d} x {wt / wc / 1e3:8.1f} us")
