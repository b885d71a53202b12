#!/usr/bin/env python3
"""Workload and summary for the counter passes over the int4 GEMM kernels (tools/pmc_quant.sh).

  python tools/pmc_quant.py run                      the workload (one launch set per shape)
  python tools/pmc_quant.py sum <sq dir> <fetch dir> <write dir> <out.json>
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vllm-splitwise_amd")]
SHAPES = [("gptq", 1), ("gptq", 32), ("gptq", 64), ("gptq", 128), ("gptq", 256), ("awq", 32)]
K, N, GROUP, REPS = 5120, 13824, 128, 5


def run():
    import torch
    from splitwise_amd import _C
    for fmt, m in SHAPES:
        a = torch.randn(m, K, dtype=torch.float16, device="cuda")
        scales = (torch.rand(K // GROUP, N, device="cuda") * 2e-3).to(torch.float16)
        qz = torch.randint(-2**31, 2**31 - 1, (K // GROUP, N // 8), dtype=torch.int32, device="cuda")
        if fmt == "gptq":
            qw = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int32, device="cuda")
            fn = lambda: _C.ops.gptq_gemm(a, qw, qz, scales, torch.empty((1, 1), device="meta"), True, 4)
        else:
            qw = torch.randint(-2**31, 2**31 - 1, (K, N // 8), dtype=torch.int32, device="cuda")
            fn = lambda: _C.ops.awq_gemm(a, qw, scales, qz, 8)
        for _ in range(REPS):
            fn()
        torch.cuda.synchronize()


def _rows(d):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"] and "Cijk" not in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return rows


def _per_dispatch(rows):
    out = {}
    for r in rows:
        out.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"][:70]})[r["Counter_Name"]] = float(r["Counter_Value"])
    return [out[k] for k in sorted(out)]


def summarise(sq_dir, fetch_dir, write_dir, out_path):
    sq, fe, wr = (_per_dispatch(_rows(d)) for d in (sq_dir, fetch_dir, write_dir))
    assert len(sq) == len(fe) == len(wr) == len(SHAPES) * REPS, (len(sq), len(fe), len(wr))
    result = []
    for i, (fmt, m) in enumerate(SHAPES):
        last = slice(i * REPS + 1, (i + 1) * REPS)           # skip the first (cold) launch of each shape
        n = REPS - 1
        busy = sum(d["SQ_VALU_MFMA_BUSY_CYCLES"] for d in sq[last]) / n
        cycles = sum(d["GRBM_GUI_ACTIVE"] for d in sq[last]) / n            # summed over the 8 XCDs
        fetch = sum(d["FETCH_SIZE"] for d in fe[last]) / n * 1024 * 2      # KiB; gfx950 counts 128-B requests as 64 B
        write = sum(d["WRITE_SIZE"] for d in wr[last]) / n * 1024
        alg = K * N // 2 + (K // GROUP) * N * 2 + (K // GROUP) * N // 2 + m * K * 2 + m * N * 2
        result.append({"kernel": sq[i * REPS + 1]["name"], "format": fmt, "M": m, "K": K, "N": N,
                       "gpu_cycles": round(cycles), "mfma_busy_cycles_all_simds": round(busy),
                       "mfma_utilisation": round(busy / (cycles / 8 * 1024), 4),
                       "hbm_bytes": round(fetch + write), "algorithmic_bytes": alg,
                       "hbm_over_algorithmic": round((fetch + write) / alg, 3)})
    json.dump({"note": "rocprofv3 --pmc, one pass per counter group; MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / "
                       "(GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); HBM bytes = FETCH_SIZE KiB x 1024 x 2 + WRITE_SIZE KiB x 1024",
               "kernels": result}, open(out_path, "w"), indent=1)
    for r in result:
        print(r)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        summarise(*sys.argv[2:6])
