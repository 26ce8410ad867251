#!/usr/bin/env python3
"""Rehearsal of BASELINE config 4 on ONE GPU: rank 0's share of 4096x4096 @ 256 spp over 8 GPUs
(interleaved 16-row stripes, 512 rows = 537 M paths) — checks memory/batching at that size and gives the per-GPU time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rust_raytrace_amd import raytrace as R, dist as rd
W = H = 4096
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
scene = R.canonical_scene(os.path.join(ROOT, "tests", "golden", "teapot_tri.obj"))
vp = R.canonical_viewport(W, H, 5, spp)
tile = rd.rank_tile(0, 8, H, 16)
buf = torch.zeros((tile[1], W, 4), dtype=torch.float32, device="cuda:0")
c = R.HipRayCaster(seed=1)
c.upload(scene)
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx = c.walk_tile_device(vp, scene, tile, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"config 4 share (rank 0 of 8): tile {tile}, {ctx.total_rays} rays in {dt:.3f} s = {ctx.total_rays / dt / 1e6:.1f} Mrays/s; "
          f"finite={bool(torch.isfinite(buf).all())} peak mem {torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0] >> 20} MiB used")
