#!/usr/bin/env python3
"""Where a decode step's time goes on the GPU: from a rocprofv3 kernel trace of bench.py, takes the
last full-batch decode steps (runs of kernels between two sampler launches whose paged-attention
launches are the longest) and prints busy time, idle gaps and the per-kernel shares."""
import csv
import glob
import sys
from collections import defaultdict

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# a step = the kernels from one paged-attention layer-0 launch to the next: split at the largest idle gaps instead
pa = [i for i, r in enumerate(rows) if "paged_attention" in r[2]]
print(f"{len(rows)} kernels, {len(pa)} decode-attention launches")
# group paged-attention launches into steps of `layers` consecutive launches (argv[2], default 32)
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = [pa[i:i + layers] for i in range(0, len(pa) - layers + 1, layers)]
best = sorted(steps, key=lambda st: -sum(rows[i][1] - rows[i][0] for i in st))[:8]
for st in sorted(best):
    lo, hi = st[0], st[-1]
    # extend to the kernels between this step's first attention launch and the next step's first
    nxt = next((s[0] for s in steps if s[0] > hi), None)
    if nxt is None:
        continue
    # one full period: from layer-0 attention of this step to layer-0 attention of the next
    seg = rows[lo:nxt]
    span = seg[-1][1] - seg[0][0] if False else rows[nxt][0] - rows[lo][0]
    busy = sum(e - s for s, e, _ in seg)
    by = defaultdict(int)
    for s, e, n in seg:
        key = n.split("(")[0].split("<")[0][-40:] if "Cijk" not in n else "gemm " + n[n.find("MT"):n.find("MT") + 12]
        by[key] += e - s
    gaps = [seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1)]
    big = sorted(gaps, reverse=True)[:3]
    print(f"period {span / 1e6:7.2f} ms  kernels {len(seg):4d}  busy {busy / 1e6:7.2f} ms  idle {(span - busy) / 1e6:6.2f} ms  "
          f"largest gaps {[round(g / 1e3) for g in big]} us")
    top = sorted(by.items(), key=lambda kv: -kv[1])[:9]
    print("     " + "  ".join(f"{k.strip()}={v / 1e6:.2f}" for k, v in top))
