#!/bin/bash
# Development tool (runs on the GPU box): the two HBM-traffic counter passes only.  tools/pmc_traffic.sh <k> <out-dir>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
K=$1
OUT=$R/gpurun_out/$2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $SET --output-format csv -d "$OUT/pass$i" -- python3 "$R/tools/prof_driver.py" "$K" 2 > "$OUT/pass$i.log" 2>&1
  rc=$?
  echo "pass $i rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 "$OUT/pass$i.log"; exit $rc; fi
done
