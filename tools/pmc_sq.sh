#!/bin/bash
# Development tool (runs on the GPU box): the two SQ counter passes only.  tools/pmc_sq.sh <k> <out-dir>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
K=$1
OUT=$R/gpurun_out/$2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for SET in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
  "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD" ; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $SET --output-format csv -d "$OUT/pass$i" -- python3 "$R/tools/prof_driver.py" "$K" 2 > "$OUT/pass$i.log" 2>&1
  rc=$?
  echo "pass $i rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 "$OUT/pass$i.log"; exit $rc; fi
done
