#!/usr/bin/env python3
"""GPU idle share at the end of a rocprofv3 kernel trace of bench.py: over the last `window_ms` of the trace
(the timed, steady-state steps) prints busy time, idle time and the largest idle gaps with the kernels on
either side.  usage: gpu_idle.py <trace dir> [window_ms]"""
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
window = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 500e6
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
end = rows[-1][1]
rows = [r for r in rows if r[0] >= end - window]
span = rows[-1][1] - rows[0][0]
busy = 0
cursor = rows[0][0]
gaps = []
for i, (s, e, n) in enumerate(rows):
    if s > cursor:
        gaps.append((s - cursor, rows[i - 1][2] if i else "", n))
    if e > cursor:
        busy += e - max(s, cursor)
        cursor = e
print(f"window {span / 1e6:.1f} ms: busy {busy / 1e6:.1f} ms ({100 * busy / span:.1f} %), idle {(span - busy) / 1e6:.1f} ms in {len(gaps)} gaps")
short = lambda n: n.split("(")[0][-60:]
hist = {}
for g, a, b in gaps:
    key = "<10us" if g < 10e3 else "<100us" if g < 100e3 else "<1ms" if g < 1e6 else ">=1ms"
    hist.setdefault(key, [0, 0])
    hist[key][0] += 1
    hist[key][1] += g
for k, (c, t) in hist.items():
    print(f"  gaps {k:7s}: {c:6d}  total {t / 1e6:8.2f} ms")
for g, a, b in sorted(gaps, reverse=True)[:12]:
    print(f"  {g / 1e3:9.1f} us  after {short(a)}  before {short(b)}")
