#!/bin/bash
# Development tool (runs on the GPU box): SQ / LDS / TA counter passes over one variant of tools/brick_lab.
#   tools/pmc_lab.sh <variant> <out-dir-under-gpurun_out>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
V=$1
OUT=$R/gpurun_out/$2
mkdir -p "$OUT"
[ -f /tmp/brick_lab.bin ] || python3 "$R/tools/brick_lab.py" 64 /tmp/brick_lab.bin || exit 1
cd /tmp && export TMPDIR=/tmp
i=0
for SET in \
  "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
  "SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS" \
  "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM" \
  "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" ; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $SET --output-format csv -d "$OUT/pass$i" -- "$R/tools/brick_lab" /tmp/brick_lab.bin "$V" 3 > "$OUT/pass$i.log" 2>&1
  rc=$?
  echo "pass $i rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 "$OUT/pass$i.log"; exit $rc; fi
done
python3 "$R/tools/pmc_summary.py" "$OUT" unproject > "$OUT/summary.txt"
cat "$OUT/summary.txt"
