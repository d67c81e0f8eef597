#!/bin/bash
# Development tool (runs on the GPU box): one counter pass.  tools/pmc_set.sh <prof_driver key> <out-dir> <counters...>
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
K=$1
OUT=$R/gpurun_out/$2
shift 2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pass" -- python3 "$R/tools/prof_driver.py" "$K" 2 > "$OUT/pass.log" 2>&1
rc=$?
echo "pass rc=$rc"
if [ $rc -ne 0 ]; then tail -5 "$OUT/pass.log"; exit $rc; fi
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/pass/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        acc[(row["Kernel_Name"][:60], row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("%-62s %-34s n=%d max=%.4g" % (k, c, len(v), max(v)))
PY
