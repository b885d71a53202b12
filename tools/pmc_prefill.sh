#!/bin/bash
# Counter passes over the prompt-attention kernel; prints per-launch averages of every counter.
# usage (GPU box, repo root): tools/pmc_prefill.sh <seq_len> <num_seqs>
root=$(pwd)
export TMPDIR=/tmp
cd /tmp
pass() {
  rm -rf /tmp/pmc_pf_$1
  timeout -s KILL 120 rocprofv3 --pmc $2 --output-format csv -d /tmp/pmc_pf_$1 -o pmc -- python3 $root/tools/prefill_once.py $3 $4 > /tmp/pmc_pf_$1.log 2>&1 || { tail -5 /tmp/pmc_pf_$1.log; exit 1; }
}
pass a "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" $1 $2
pass b "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" $1 $2
pass c "SQ_WAVES SQ_INSTS_VALU_TRANS_F32 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES" $1 $2
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/pmc_pf_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "prefill_attention_kernel" in row["Kernel_Name"]:
            agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
for name in sorted(agg):
    v = agg[name]
    print(f"{name:32s} {sum(v) / len(v):16.0f}  ({len(v)} launches)")
PY
