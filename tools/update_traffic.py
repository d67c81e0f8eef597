"""Development tool: turns the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh into a record of
profiles/pmc_traffic.json, stamped with the SHA-256 of the kernel sources (bench.py reports the traffic only while
that hash matches the sources the library is built from).

    python3 tools/update_traffic.py <pass-dir under gpurun_out> <kernel key> <kernel-name substring> '<shape json>' <note>
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_hash, KERNEL_SOURCES)

pass_dir, key, needle, shape, note = sys.argv[1:6]
vals = {"FETCH_SIZE": [], "WRITE_SIZE": []}
for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", pass_dir, "pass*", "*", "*counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        if needle in row["Kernel_Name"] and row["Counter_Name"] in vals:
            vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
# the first dispatch of a run can be a small check launch: keep the dispatches of the dominant (largest) size
fetch = max(vals["FETCH_SIZE"])
write = max(vals["WRITE_SIZE"])
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
rec = json.load(open(path))
rec[key] = {
    "shape": json.loads(shape),
    "FETCH_SIZE_kb": fetch, "WRITE_SIZE_kb": write,
    "correction": "gfx950: FETCH_SIZE tallies the 128-byte requests of a wide coalesced read at 64 bytes, so it is doubled; "
                  "WRITE_SIZE is exact for 16-byte-per-lane streaming stores (MI355X_MICROARCH.md, HBM section)",
    "hbm_bytes_per_launch": int(round((2 * fetch + write) * 1024)),
    "source": note,
    "source_sha256": bench.kernel_source_hash(key),
}
json.dump(rec, open(path, "w"), indent=1)
print(key, rec[key]["hbm_bytes_per_launch"], rec[key]["source_sha256"][:12])
