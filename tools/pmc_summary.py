#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection.csv files per kernel.  Usage: pmc_summary.py <dir> [<dir> ...]"""
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            k = "k_render" if "k_render" in k else None
            if not k:
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k].add((f, r["Dispatch_Id"]))
for k, v in agg.items():
    print(k, "dispatches", len(calls[k]))
    for c in sorted(v):
        print(f"   {c:28s} {v[c]:.0f}")
