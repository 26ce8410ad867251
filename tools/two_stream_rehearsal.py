#!/usr/bin/env python3
"""Experiment: render one rank's tile as NS sub-tiles on NS HIP streams from NS host threads (two scene handles),
so that the small deep passes of one sub-tile overlap the bulk of the other.  One GPU; prints time for N=1 and for
rank r of 8."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rust_raytrace_amd import raytrace as R, dist as rd
W = H = 2048
spp, S = 64, 16
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 2
obj = os.path.join(ROOT, "tests", "golden", "teapot_tri.obj")
scenes = [R.canonical_scene(obj) for _ in range(NS)]
casters = [R.HipRayCaster(seed=1) for _ in range(NS)]
for c, s in zip(casters, scenes):
    c.upload(s)
vp = R.canonical_viewport(W, H, 5, spp)
streams = [torch.cuda.Stream() for _ in range(NS)]

def run(world, r):
    tiles = [rd.rank_tile(r + t * world, world * NS, H, S) for t in range(NS)]
    bufs = [torch.zeros((tl[1], W, 4), dtype=torch.float32, device="cuda:0") for tl in tiles]
    rays = [0] * NS
    def work(t):
        rays[t] = casters[t].walk_tile_device(vp, scenes[t], tiles[t], bufs[t].data_ptr(), streams[t].cuda_stream).total_rays
    def once():
        th = [threading.Thread(target=work, args=(t,)) for t in range(NS)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for x in th: x.start()
        for x in th: x.join()
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    once()
    return min(once(), once()), sum(rays)

dt, rays = run(1, 0)
print(f"NS={NS} N=1: {dt*1e3:.1f} ms, {rays/dt/1e6:.0f} Mrays/s")
ts = [run(8, r)[0] for r in range(8)]
print(f"NS={NS} N=8 per-rank ms: {[round(t*1e3,1) for t in ts]} max {max(ts)*1e3:.1f}")
