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

# HBM-side traffic of the dominant kernel per launch, corrected as MI355X_MICROARCH.md "HBM" prescribes:
# FETCH_SIZE/WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of the bytes of wide reads (x2).
for k in out:
    if k.startswith("k_trace") and "FETCH_SIZE" in out[k] and "WRITE_SIZE" in out[k]:
        n = max(out[k]["FETCH_SIZE_dispatches"], 1)
        b = (2.0 * out[k]["FETCH_SIZE"] + out[k]["WRITE_SIZE"]) * 1024.0 / n
        print(f"{k}: L2-miss (fabric) bytes per launch = {b:.4g} over {n} launches")
        json.dump({"kernel": k, "k_trace_hbm_bytes_per_launch": b, "launches": n, "FETCH_SIZE_KB": out[k]["FETCH_SIZE"],
                   "WRITE_SIZE_KB": out[k]["WRITE_SIZE"], "note": "FETCH_SIZE x2 (gfx950) + WRITE_SIZE; fabric-side requests, Infinity-Cache hits included"},
                  open(os.path.join(root, "pmc_traffic.json"), "w"), indent=1)
