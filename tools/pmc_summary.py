#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output: per kernel, mean of each counter over its dispatches."""
import csv, glob, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
        rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in rows.items():
    if len(sys.argv) > 2 and sys.argv[2] not in k:
        continue
    print(k, " dispatches:", max(len(v) for v in d.values()))
    for c, v in sorted(d.items()):
        print(f"    {c:34s} {sum(v)/len(v):16.0f}")
