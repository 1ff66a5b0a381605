#!/usr/bin/env python3
"""tools/pmc_summary.py DIR TAG -- fold the rocprofv3 CSVs tools/profile_round.sh wrote under DIR into
DIR/TAG_kernel_stats.csv (copy of the --stats kernel table), DIR/TAG_pmc_counters.json (per-kernel
averages per launch of every counter) and DIR/traffic.json (HBM bytes per launch, corrected as
MI355X_MICROARCH.md's HBM section prescribes: FETCH_SIZE/WRITE_SIZE are KiB-ish units of 1 KB... see note)."""
import csv
import glob
import json
import os
import shutil
import sys

out, tag = sys.argv[1], sys.argv[2]
counters = {}
for f in sorted(glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    acc = {}
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0]
        if not name.replace("void ", "").startswith(("hvk_k_raster", "hvk_k_filter", "hvk_k_direct", "hvk_k_prep", "hvk_k_secam", "hvk_k_fused")):
            continue
        name = name.replace("void ", "")
        acc.setdefault((name, row["Counter_Name"]), []).append(float(row["Counter_Value"]))
    for (name, c), v in acc.items():
        counters.setdefault(name, {})[c] = round(sum(v) / len(v))
json.dump(counters, open(os.path.join(out, tag + "_pmc_counters.json"), "w"), indent=1)

for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(out, tag + "_kernel_stats.csv"))

bench = {}
try:
    bench = json.loads(open(os.path.join(out, "bench.json")).read().strip().splitlines()[-1])
except Exception as e:  # noqa: BLE001
    print("no bench line:", e)
frames = bench.get("config", {}).get("frames_per_gpu_per_step", 128)
t = {"frames": frames,
     "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), averages per launch of bench.py --steps 3. "
             "Units KiB. FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md (HBM section); WRITE_SIZE as reported."}
for name, c in counters.items():
    key = name.split("<")[0]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        t[key + "_fetch_KiB_raw"] = c["FETCH_SIZE"]
        t[key + "_write_KiB"] = c["WRITE_SIZE"]
        t[key + "_bytes_per_launch"] = int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
if "roofline" in bench:
    t["algorithmic_bytes_per_launch"] = bench["roofline"].get("algorithmic_bytes_per_launch")
json.dump(t, open(os.path.join(out, "traffic.json"), "w"), indent=1)
print(json.dumps(counters)[:1500])
print(json.dumps(t))
