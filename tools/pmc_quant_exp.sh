#!/bin/bash
# Counter passes over one variant of the tuning harness (tools/quant_exp.hip).
# usage (GPU box, repo root): tools/pmc_quant_exp.sh "<variant name substring>" <kernel name substring> [M N K]
root=$(pwd)
variant=$1; kern=$2; M=${3:-256}; N=${4:-13824}; K=${5:-5120}
export TMPDIR=/tmp
cd /tmp
pass() {
  rm -rf /tmp/pmc_qe_$1
  timeout -s KILL 120 rocprofv3 --pmc $2 --output-format csv -d /tmp/pmc_qe_$1 -o pmc -- $root/tools/bin/quant_exp $M $N $K "$variant" > /tmp/pmc_qe_$1.log 2>&1 || { tail -5 /tmp/pmc_qe_$1.log; exit 1; }
}
pass a "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY"
pass b "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
pass c "SQ_WAVES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD"
pass d "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
python3 - "$kern" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_qe_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if sys.argv[1] in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
for name in sorted(agg):
    v = agg[name]
    print(f"{name:32s} {sum(v) / len(v):16.0f}  ({len(v)} launches)")
PY
