"""Turns one leg's tools/prof_legs.sh output into the profile bench.py reads: per-dispatch counter averages of the
leg's render kernel, per-ray figures derived from them, the rays of a dispatch and the digest of the kernel sources
the profiled build was made from (both taken from the `--profile-leg` line in the logs).

FETCH_SIZE / WRITE_SIZE are in KiB.  FETCH_SIZE tallies every read request to memory at 64 bytes although each
fetches a 128-byte line (checked with tools/calib/random_fetch for scattered reads of 32 to 256 bytes as well as for
streaming reads), so it is doubled; WRITE_SIZE is exact.

usage: python tools/make_pmc_json.py LEG_DIR OUT_JSON OUT_STATS_CSV
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

d, out_json, out_csv = sys.argv[1:4]
meta = None
for log in sorted(glob.glob(os.path.join(d, "*.log"))):
    for line in open(log, errors="replace"):
        if line.startswith("{") and "profile_leg" in line:
            m = json.loads(line)
            if meta is not None and (m["rays_per_launch"], m["source_digest"]) != (meta["rays_per_launch"], meta["source_digest"]):
                sys.exit("the passes did not run the same workload")
            meta = m
if meta is None:
    sys.exit("no --profile-leg line in the logs")

tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(d, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[k][row["Counter_Name"]] += 1
# A launch of a leg is one or two render kernels (LANES with bounce 0 in lockstep first: k_bounce + k_paths_lanes):
# their counters are added up and divided by the number of launches.
render = sorted((k for k in tot if "rt::k_paths" in k or "rt::k_bounce" in k or "rt::k_extend" in k or "rt::k_shade" in k
                 or "rt::k_generate" in k), key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", 0.0))
launches = meta["launches"]
names = sorted({n for k in render for n in tot[k]})
c = {name: sum(tot[k].get(name, 0.0) for k in render) / launches for name in names}
kernel = " + ".join(k.split("(")[0] for k in render)
rays = meta["rays_per_launch"]

avg_ms = None
stats = glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], out_csv)
    total_ns = 0.0
    for row in csv.DictReader(open(stats[0])):
        if any(row["Name"].split("(")[0] == k.split("(")[0] for k in render):
            total_ns += float(row["TotalDurationNs"])
    avg_ms = total_ns * 1e-6 / launches

simd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0  # the counter sums the 8 XCDs' busy cycles
res = {
    "leg": meta["profile_leg"], "workload": meta["workload"], "kernel": kernel,
    "launches_per_pass": launches, "rays_per_dispatch": rays, "source_digest": meta["source_digest"],
    "avg_dispatch_ms": avg_ms, "counters_per_dispatch": c,
    "lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]),
    "valu_busy_share_of_simd_cycles": 4.0 * c["SQ_ACTIVE_INST_VALU"] / simd_cycles,
    "waves_waiting_share": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
    "l2_hit_rate": c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
    "per_ray": {
        "valu_wave_instructions": c["SQ_INSTS_VALU"] / rays,
        "hbm_read_bytes": 2.0 * 1024.0 * c["FETCH_SIZE"] / rays,
        "hbm_write_bytes": 1024.0 * c["WRITE_SIZE"] / rays,
        "hbm_read_requests": c.get("TCC_EA0_RDREQ_sum", 0.0) / rays,
        "l2_read_requests": c.get("TCP_TCC_READ_REQ_sum", 0.0) / rays,
    },
    "note": "per launch: counters of the leg's render kernels summed over a pass and divided by its launches; GRBM_GUI_ACTIVE is summed over the 8 "
            "XCDs; the busy share divides counters of different passes",
}
json.dump(res, open(out_json, "w"), indent=1)
print(json.dumps({k: res[k] for k in ("leg", "kernel", "rays_per_dispatch", "avg_dispatch_ms", "lane_utilisation",
                                      "valu_busy_share_of_simd_cycles", "l2_hit_rate", "per_ray")}, indent=1))
