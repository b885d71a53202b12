#!/bin/bash
# Two counter passes over the same short bench command, then the per-launch summary.
# usage (on the GPU box, from the repo root): tools/pmc_traffic.sh <out.json> [bench args...]
set -e
out=$1; shift
root=$(pwd)
export TMPDIR=/tmp SPLITWISE_PMC_SUMMARY=/nonexistent
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout -s KILL 900 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o pmc -- python3 $root/bench.py "$@" > /tmp/pmc_$c.log 2>&1
  grep '^{' /tmp/pmc_$c.log | tail -1 > /tmp/pmc_$c.json
done
cd $root
python3 tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE /tmp/pmc_FETCH_SIZE.json $out
