"""Turns a tools/prof_pmc.sh output directory into profiles/<name>.json: HBM bytes per ray of the render
kernels.  FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts half the bytes of wide
reads (128-byte requests tallied as 64), so it is doubled; WRITE_SIZE is exact.

usage: python tools/make_traffic_json.py PMC_DIR OUT_JSON
"""
import csv
import glob
import json
import os
import re
import sys

d, out = sys.argv[1], sys.argv[2]
fetch = write = 0.0
per_kernel = {}
for f in glob.glob(os.path.join(d, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if not re.search(r"rt::k_(paths|bounce|extend|shade|generate|accumulate)", name):
            continue
        k = re.search(r"rt::(k_[a-z_]+)", name).group(1)
        v = float(row["Counter_Value"])
        if row["Counter_Name"] == "FETCH_SIZE":
            per_kernel.setdefault(k, [0.0, 0.0])[0] += v * 1024 * 2
        elif row["Counter_Name"] == "WRITE_SIZE":
            per_kernel.setdefault(k, [0.0, 0.0])[1] += v * 1024
rays = 0
log = open(os.path.join(d, "pass2.log"), errors="replace").read()
for m in re.finditer(r"rays (\d+) paths (\d+)", log):
    rays += int(m.group(1))
res = {"source": d, "rays": rays, "note": "HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes), summed over all "
       "dispatches of the run, divided by the rays the run traced", "kernels": {}}
for k, (r, w) in per_kernel.items():
    res["kernels"][k] = {"hbm_read_bytes_per_ray": r / rays, "hbm_write_bytes_per_ray": w / rays}
res["hbm_bytes_per_ray"] = sum(r + w for r, w in per_kernel.values()) / rays
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
