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

# Per-frame figures of the dominant (closest-hit) kernel for bench.py's roofline object.  FETCH_SIZE / WRITE_SIZE are in
# KB; on gfx950 FETCH_SIZE reports half of the bytes of wide reads (MI355X_MICROARCH.md "HBM": double it); they count
# fabric-side (L2-miss) requests, Infinity-Cache hits included.  One profiled run = ONE frame (bench --steps 1 --warmup 0).
import subprocess
cfg = json.loads(sys.argv[2]) if len(sys.argv) > 2 else {"scene": "canonical", "width": 2048, "height": 2048, "spp": 64, "fast": False}
try:
    commit = subprocess.check_output(["git", "-C", os.path.dirname(os.path.abspath(__file__)), "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    commit = os.environ.get("RTMI_COMMIT", "unknown")
for k in out:
    if k.startswith("k_trace") and "FETCH_SIZE" in out[k] and "WRITE_SIZE" in out[k]:
        o = out[k]
        n = max(o["FETCH_SIZE_dispatches"], 1)
        fabric = (2.0 * o["FETCH_SIZE"] + o["WRITE_SIZE"]) * 1024.0
        res = {"kernel": k, "config": cfg, "commit": commit, "source": "rocprofv3 --pmc, 5 separate passes (tools/pmc_run.sh), one frame each",
               "launches_per_frame": n, "fabric_bytes_per_frame": fabric, "fabric_bytes_per_launch": fabric / n,
               "FETCH_SIZE_KB": o["FETCH_SIZE"], "WRITE_SIZE_KB": o["WRITE_SIZE"],
               "valu_wave_insts_per_frame": o.get("SQ_INSTS_VALU"), "salu_wave_insts_per_frame": o.get("SQ_INSTS_SALU"),
               "valu_lane_utilisation": (o["SQ_THREAD_CYCLES_VALU"] / (64.0 * o["SQ_ACTIVE_INST_VALU"])) if o.get("SQ_ACTIVE_INST_VALU") else None,
               "wait_any_frac": (o["SQ_WAIT_ANY"] / o["SQ_WAVE_CYCLES"]) if o.get("SQ_WAVE_CYCLES") else None,
               "l2_hit_rate": (o["TCC_HIT_sum"] / (o["TCC_HIT_sum"] + o["TCC_MISS_sum"])) if o.get("TCC_HIT_sum") else None,
               "note": "fabric bytes = 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE; L2-miss side requests, Infinity-Cache hits included"}
        print(json.dumps(res, indent=1))
        json.dump(res, open(os.path.join(root, "pmc_traffic.json"), "w"), indent=1)
