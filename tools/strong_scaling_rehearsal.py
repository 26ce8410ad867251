#!/usr/bin/env python3
"""Rehearsal on ONE GPU of the per-rank render time of config 3 tiled over N ranks (no gather):
time of rank r's stripes, for every r, so that max_r predicts the N-GPU frame time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rust_raytrace_amd import raytrace as R, dist as rd
W = H = 2048
spp = 64
STRIPE = int(sys.argv[1]) if len(sys.argv) > 1 else 16  # stripe height
scene = R.canonical_scene(os.path.join(ROOT, "tests", "golden", "teapot_tri.obj"), gpu_build=0)
vp = R.canonical_viewport(W, H, 5, spp)
c = R.HipRayCaster(seed=1)
c.upload(scene)
st = torch.cuda.current_stream().cuda_stream
base = None
for world in (1, 2, 4, 8):
    times, rays, per = [], 0, []
    for r in range(world):
        tile = rd.rank_tile(r, world, H, STRIPE)
        buf = torch.zeros((tile[1], W, 4), dtype=torch.float32, device="cuda:0")
        c.walk_tile_device(vp, scene, tile, buf.data_ptr(), st)  # warm-up (allocations)
        best = None
        for _ in range(3):  # the best of three: the first tile after a pause also pays the clock ramp
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx = c.walk_tile_device(vp, scene, tile, buf.data_ptr(), st)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        times.append(best); rays += ctx.total_rays; per.append(ctx.total_rays)
    if base is None:
        base = max(times)
    print([round(t*1e3,1) for t in times], 'Mrays per rank', [round(p / 1e6, 2) for p in per]); print(f"N={world}: per-rank render time max {max(times)*1e3:.1f} ms min {min(times)*1e3:.1f} ms -> predicted {rays / max(times) / 1e6:.0f} Mrays/s, "
          f"efficiency {base / (world * max(times)):.2f} (before the gather)")
