#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter CSVs per kernel and write summary.json (all counters) + pmc_traffic.json (the per-launch
figures bench.py's roofline object reads as profiles/pmc_latest.json).

One profiled run = ONE single-stream frame (bench.py --steps 1 --warmup 0 with RTMI_STREAMS=1: every launch has the GPU to
itself, which is also the launch set bench.py times for `roofline.achieved`).  FETCH_SIZE / WRITE_SIZE are in KB; on gfx950
FETCH_SIZE reports half of the bytes of wide reads (MI355X_MICROARCH.md "HBM": double it); they count fabric-side
(L2-miss) requests, Infinity-Cache hits included.  The counter CSV carries each dispatch's start/end timestamps:
effective clock = GRBM_GUI_ACTIVE / 8 XCDs / duration of the dispatches of the pass that collected it."""
import csv, glob, json, os, subprocess, sys
from collections import defaultdict
root = sys.argv[1]
tot = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
dur = defaultdict(lambda: defaultdict(float))   # kernel -> counter -> summed dispatch duration (ns) of the pass that has the counter
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rtmi::", "")
        if not k.startswith("k_"):
            continue
        k = k.split("<")[0]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
        try:
            dur[k][r["Counter_Name"]] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        except (KeyError, ValueError):
            pass
out = {}
for k in sorted(tot):
    print(k)
    out[k] = {}
    for c in sorted(tot[k]):
        print(f"   {c:34s} {tot[k][c]:.6g}   ({calls[k][c]} dispatches, {dur[k][c] / 1e6:.3f} ms)")
        out[k][c] = tot[k][c]
        out[k][c + "_dispatches"] = calls[k][c]
        out[k][c + "_dispatch_ms"] = dur[k][c] / 1e6
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)

cfg = json.loads(sys.argv[2]) if len(sys.argv) > 2 else {"scene": "canonical", "width": 2048, "height": 2048, "spp": 64, "fast": False}
try:
    commit = subprocess.check_output(["git", "-C", os.path.dirname(os.path.abspath(__file__)), "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    commit = os.environ.get("RTMI_COMMIT", "unknown")
res = {"config": cfg, "commit": commit, "streams": 1,
       "source": "rocprofv3 --pmc, 5 separate passes (tools/pmc_run.sh) over ONE single-stream frame each (RTMI_STREAMS=1)",
       "note": "fabric bytes = 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE; L2-miss side requests, Infinity-Cache hits included; "
               "per launch = per dispatch of the single-stream frame", "kernels": {}}
for k, o in out.items():
    if "FETCH_SIZE" not in o or "WRITE_SIZE" not in o:
        continue
    n = max(o["FETCH_SIZE_dispatches"], 1)
    fabric = (2.0 * o["FETCH_SIZE"] + o["WRITE_SIZE"]) * 1024.0
    g = o.get("GRBM_GUI_ACTIVE")
    res["kernels"][k] = {
        "launches_per_frame": n, "fabric_bytes_per_frame": fabric, "fabric_bytes_per_launch": fabric / n,
        "FETCH_SIZE_KB": o["FETCH_SIZE"], "WRITE_SIZE_KB": o["WRITE_SIZE"],
        "valu_wave_insts_per_launch": (o["SQ_INSTS_VALU"] / max(o["SQ_INSTS_VALU_dispatches"], 1)) if o.get("SQ_INSTS_VALU") else None,
        "salu_wave_insts_per_launch": (o["SQ_INSTS_SALU"] / max(o["SQ_INSTS_SALU_dispatches"], 1)) if o.get("SQ_INSTS_SALU") else None,
        "valu_lane_utilisation": (o["SQ_THREAD_CYCLES_VALU"] / (64.0 * o["SQ_ACTIVE_INST_VALU"])) if o.get("SQ_ACTIVE_INST_VALU") else None,
        "wait_any_frac": (o["SQ_WAIT_ANY"] / o["SQ_WAVE_CYCLES"]) if o.get("SQ_WAVE_CYCLES") else None,
        "l2_hit_rate": (o["TCC_HIT_sum"] / (o["TCC_HIT_sum"] + o["TCC_MISS_sum"])) if o.get("TCC_HIT_sum") else None,
        "effective_clock_GHz": round(g / 8.0 / o["GRBM_GUI_ACTIVE_dispatch_ms"] / 1e6, 4) if g and o.get("GRBM_GUI_ACTIVE_dispatch_ms") else None,
        "profiled_launch_ms": (o["GRBM_GUI_ACTIVE_dispatch_ms"] / max(o["GRBM_GUI_ACTIVE_dispatches"], 1)) if g else None}
print(json.dumps(res, indent=1))
json.dump(res, open(os.path.join(root, "pmc_traffic.json"), "w"), indent=1)
