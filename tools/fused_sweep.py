#!/usr/bin/env python3
"""Development aid: config 3 frame for a list of tunings (python dicts, one per argv; the key "options" = rtmi option bits)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rust_raytrace_amd import raytrace as R
W = H = 2048
scene = R.canonical_scene(os.path.join(ROOT, "tests", "golden", "teapot_tri.obj"), gpu_build=0)
vp = R.canonical_viewport(W, H, 5, 64)
st = torch.cuda.current_stream().cuda_stream
buf = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
for arg in sys.argv[1:]:
    tun = eval(arg)
    opts = tun.pop("options", 0)  # e.g. R.OPT_BVH = 8
    c = R.HipRayCaster(seed=1, options=opts, tuning=tun)
    c.walk_tile_device(vp, scene, (0, H, H, 0), buf.data_ptr(), st)
    best = None
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx = c.walk_tile_device(vp, scene, (0, H, H, 0), buf.data_ptr(), st)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        if best is None or dt < best[0]:
            best = (dt, ctx)
    dt, ctx = best
    s = ctx.stats
    print(f"{str(tun):70s} opt {opts} wall {dt:7.1f} ms primary {s['primary_ms']:7.1f} bounce {s['bounce_ms']:7.1f} {ctx.total_rays / dt / 1e3:7.1f} Mrays/s", flush=True)
