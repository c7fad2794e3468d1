"""Sums rocprofv3 --pmc counter CSVs per kernel: python tools/pmc_summary.py DIR"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(d, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][:60]
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[k][row["Counter_Name"]] += 1
for k in sorted(tot, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", 0)):
    print(k)
    for c in sorted(tot[k]):
        print(f"    {c:32s} {tot[k][c]:.6g}  (dispatches {calls[k][c]})")
