#!/usr/bin/env python3
"""Averages HBM traffic per launch of one kernel from two rocprofv3 --pmc passes.

  python tools/pmc_traffic.py <fetch_pass_dir> <write_pass_dir> <bench_json_line_file> <out.json>

Each pass directory holds a *counter_collection.csv of `bench.py` run under
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE`.  The bench's roofline measurement launches the
decode paged-attention kernel `launches_timed` times at the very end of the process, on batches
whose algorithmic bytes it reports, so the last `launches_timed` dispatches of the kernel are
exactly those launches.  FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies the
128-byte requests of wide streaming reads at 64 bytes, so it is doubled.
"""
import csv
import glob
import json
import os
import sys

KERNEL = "paged_attention_kernel"


def last_values(pass_dir, counter, count):
    files = glob.glob(os.path.join(pass_dir, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {pass_dir}")
    per_dispatch = {}
    for path in files:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if KERNEL in row["Kernel_Name"] and row["Counter_Name"] == counter:
                    key = int(row["Dispatch_Id"])
                    per_dispatch[key] = per_dispatch.get(key, 0.0) + float(row["Counter_Value"])
    ordered = [per_dispatch[k] for k in sorted(per_dispatch)]
    if len(ordered) < count:
        raise SystemExit(f"{counter}: {len(ordered)} dispatches of {KERNEL}, need {count}")
    return ordered[-count:]


def main():
    fetch_dir, write_dir, bench_file, out_path = sys.argv[1:5]
    line = [ln for ln in open(bench_file) if ln.startswith("{")][-1]
    roof = json.loads(line)["roofline"]
    n = roof["launches_timed"]
    fetch = last_values(fetch_dir, "FETCH_SIZE", n)
    write = last_values(write_dir, "WRITE_SIZE", n)
    fetch_bytes = sum(fetch) / n * 1024.0 * 2.0
    write_bytes = sum(write) / n * 1024.0
    summary = {
        "kernel": KERNEL, "launches": n, "alg_bytes_per_launch": roof["alg_bytes_per_launch"],
        "fetch_bytes_per_launch": int(fetch_bytes), "write_bytes_per_launch": int(write_bytes),
        "traffic_bytes_per_launch": int(fetch_bytes + write_bytes),
        "traffic_over_algorithmic": round((fetch_bytes + write_bytes) / roof["alg_bytes_per_launch"], 4),
        "corrections": "FETCH_SIZE KiB x 1024 x 2 (gfx950 tallies 128-byte read requests at 64 bytes); WRITE_SIZE KiB x 1024",
    }
    with open(out_path, "w") as f:
        json.dump(summary, f, indent=1)
        f.write("\n")
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
