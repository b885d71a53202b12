#!/usr/bin/env python3
"""Where a prompt step's GPU time goes: from a rocprofv3 kernel trace of bench.py, takes the last prompt steps
(runs of kernels that contain `layers` prompt-attention launches) and prints, per step, the span from the step's
first kernel to its last and the time per kernel family.  usage: prompt_step_breakdown.py <trace dir> [layers]"""
import csv
import glob
import sys
from collections import defaultdict

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
layers = int(sys.argv[2]) if len(sys.argv) > 2 else 32
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
pf = [i for i, r in enumerate(rows) if "prefill_attention_kernel" in r[2]]
steps = [pf[i:i + layers] for i in range(0, len(pf) - layers + 1, layers)]


def family(name):
    if "Cijk" in name:
        return "gemm " + name[name.find("MT"):name.find("MT") + 12]
    return name.split("(")[0].split("<")[0].replace("void ", "")[-36:]


for st in steps[-4:]:
    first_attn, last_attn = st[0], st[-1]
    # the step runs from the embedding lookup before the first layer to the sampler after the last: extend both ways
    # to the nearest decode-attention launch (the neighbouring decode steps)
    lo = first_attn
    while lo > 0 and "paged_attention" not in rows[lo - 1][2]:
        lo -= 1
    hi = last_attn
    while hi + 1 < len(rows) and "paged_attention" not in rows[hi + 1][2]:
        hi += 1
    # the kernels between the previous decode step's last attention and this step's first belong partly to that
    # decode step (its last layer's GEMMs, its sampler): count from the first kernel after the previous sampler
    seg = rows[lo:hi + 1]
    by = defaultdict(lambda: [0, 0])
    for s, e, n in seg:
        by[family(n)][0] += e - s
        by[family(n)][1] += 1
    busy = sum(e - s for s, e, _ in seg)
    print(f"kernels between two decode steps' attention launches: {len(seg)}  span {(seg[-1][1] - seg[0][0]) / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms")
    for k, (t, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f"    {k:40s} {t / 1e6:7.3f} ms  x{c}")
