#!/bin/bash
# Development tool (runs on the GPU box): SQ / LDS counter passes over the unprojection backward kernels
# (tools/backward_lab.py, 8 frames).   tools/pmc_backward.sh <aggregation 0..3> <out-dir-under-gpurun_out>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
A=$1
OUT=$R/gpurun_out/$2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for SET in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
  "SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" ; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $SET --output-format csv -d "$OUT/pass$i" -- python3 "$R/tools/backward_lab.py" 8 "$A" "$R/poseestimation_amd/libmvx.so" > "$OUT/pass$i.log" 2>&1
  rc=$?
  echo "pass $i rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 "$OUT/pass$i.log"; exit $rc; fi
done
python3 "$R/tools/pmc_summary.py" "$OUT" unproject_backward_maps > "$OUT/summary.txt"
cat "$OUT/summary.txt"
