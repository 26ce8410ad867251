#!/usr/bin/env python3
"""PCIe-inclusive rate of config 3: rtmi_render (host output buffer, one 64 MiB D2H per frame) vs the device-resident path."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rust_raytrace_amd import raytrace as R
W = H = 2048
scene = R.canonical_scene(os.path.join(ROOT, "tests", "golden", "teapot_tri.obj"))
vp = R.canonical_viewport(W, H, 5, 64)
c = R.HipRayCaster(seed=1)
img = np.zeros((H, W, 4), np.float32)
c.walk_rays(vp, scene, img)
t0 = time.perf_counter(); ctx = c.walk_rays(vp, scene, img); dt_host = time.perf_counter() - t0
buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
c.walk_tile_device(vp, scene, (0, H, H, 0), buf.data_ptr(), st); torch.cuda.synchronize()
t0 = time.perf_counter(); ctx2 = c.walk_tile_device(vp, scene, (0, H, H, 0), buf.data_ptr(), st); torch.cuda.synchronize(); dt_dev = time.perf_counter() - t0
print(f"host-buffer (PCIe-inclusive): {dt_host*1e3:.1f} ms = {ctx.total_rays/dt_host/1e6:.1f} Mrays/s; device-resident: {dt_dev*1e3:.1f} ms = {ctx2.total_rays/dt_dev/1e6:.1f} Mrays/s")
