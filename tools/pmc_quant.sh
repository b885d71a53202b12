#!/bin/bash
# Counter passes over the int4 GEMM kernels: matrix-core busy cycles, HBM reads, HBM writes.
# usage (on the GPU box, from the repo root): tools/pmc_quant.sh <out.json>
set -e
out=$1
root=$(pwd)
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pq_sq /tmp/pq_fetch /tmp/pq_write
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pq_sq -o pq -- python3 $root/tools/pmc_quant.py run > /tmp/pq_sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pq_fetch -o pq -- python3 $root/tools/pmc_quant.py run > /tmp/pq_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pq_write -o pq -- python3 $root/tools/pmc_quant.py run > /tmp/pq_write.log 2>&1
cd $root
python3 tools/pmc_quant.py sum /tmp/pq_sq /tmp/pq_fetch /tmp/pq_write $out
