#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter CSVs per kernel (short name) and print a table + JSON."""
import csv, glob, json, os, sys
from collections import defaultdict
root = sys.argv[1]
tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rtmi::", "")
        if not k.startswith("k_"):
            continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
out = {}
for k in sorted(tot):
    print(k)
    out[k] = {}
    for c in sorted(tot[k]):
        print(f"   {c:34s} {tot[k][c]:.6g}   ({calls[k][c]} dispatches)")
        out[k][c] = tot[k][c]
        out[k][c + "_dispatches"] = calls[k][c]
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
