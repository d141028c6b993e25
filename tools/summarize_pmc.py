#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per kernel name (substring filter), mean counter value per dispatch."""
import csv, glob, sys, collections, json
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
flt = sys.argv[2] if len(sys.argv) > 2 else "k_traverse"
out = {}
for path in sorted(glob.glob(f"{root}/*/*/*counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row.get("Kernel_Name", "")
            if flt not in k:
                continue
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        d = out.setdefault(k, {})
        for c, v in cs.items():
            d[c] = {"mean": sum(v) / len(v), "n": len(v)}
print(json.dumps(out, indent=1))
