#!/bin/bash
# Development tool (runs on the GPU box): the bench lines and rocprofv3 kernel summaries kept under profiles/.
#   tools/measure_all.sh <out-dir-under-gpurun_out>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
mkdir -p "$OUT"
cd "$R" || exit 1
run() {  # name, args...
  local name=$1; shift
  timeout -k 10 280 python3 -u bench.py "$@" > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err" || { echo "bench $name failed"; tail -5 "$OUT/bench_$name.err"; exit 1; }
  echo "bench $name ok"
}
run default
run algebraic_coords_only --no-writeback --no-cpu-baseline
run volumetric --workload volumetric
run ransac --workload ransac
run training --workload training --steps 20 --warmup 5
MVX_BENCH_REHEARSAL=1 run rehearsal_2ranks_one_gpu --gpus 2 --steps 5
cd /tmp && export TMPDIR=/tmp
# (20 warm-up steps: the clocks ramp up over the first ~15 launches after an idle spell, and the per-kernel average of
# a profile that starts cold sits well above the steady state the bench line reports)
for W in algebraic volumetric ransac training; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$W" -- python3 "$R/bench.py" --workload $W --steps 30 --warmup 20 --no-cpu-baseline --no-extra > "$OUT/prof_$W.log" 2>&1 || { echo "rocprof $W failed"; tail -5 "$OUT/prof_$W.log"; exit 1; }
  echo "rocprof $W ok"
done
