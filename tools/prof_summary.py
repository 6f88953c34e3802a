#!/usr/bin/env python3
"""Summarises a tools/prof.sh output directory: kernel stats table and per-launch averages of the SQ counters."""
import collections
import csv
import glob
import json
import sys

out = sys.argv[1]
for f in glob.glob(out + "/trace/*/*kernel_stats.csv"):
    print(open(f).read())
summary = {}
for f in sorted(glob.glob(out + "/pmc*/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        summary.setdefault(r["Kernel_Name"], {}).update({"VGPR": r["VGPR_Count"], "SGPR": r["SGPR_Count"], "LDS": r["LDS_Block_Size"]})
    for k, d in agg.items():
        summary.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in d.items()})
json.dump(summary, open(out + "/pmc_avg_per_launch.json", "w"), indent=1, sort_keys=True)
for k, d in summary.items():
    if "tile" in k:
        print(k)
        for c in sorted(d):
            print(f"   {c:28s} {d[c]}")
