"""Development tool: prints per-kernel counter values (mean over dispatches) from rocprofv3 --pmc CSVs.
    python3 tools/pmc_summary.py gpurun_out/pmc_k3 [kernel-substring]"""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "mvx"
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob(root + "/pass*/*/*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if flt not in name:
            continue
        short = name.replace("(anonymous namespace)::", "").replace("void mvx::", "").replace("mvx::", "").split("(")[0][:60]
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print("==", k)
    for c, vals in cs.items():
        print("   %-36s mean %.6g  (n=%d)" % (c, sum(vals) / len(vals), len(vals)))
