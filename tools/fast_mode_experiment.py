#!/usr/bin/env python3
"""RTMI_OPT_FAST (skip boxes entirely behind the ray origin) against exact mode: speed-up and differing-pixel count."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from rust_raytrace_amd import raytrace as R
W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
scene = (R.grid_scene if len(sys.argv) > 3 and sys.argv[3] == "grid" else R.canonical_scene)(os.path.join(os.getcwd(), "tests", "golden", "teapot_tri.obj"))
vp = R.canonical_viewport(W, H, 5, spp)
def run(options=0):
    img = np.zeros((H, W, 4), np.float32)
    c = R.HipRayCaster(seed=1, options=options)
    c.walk_rays(vp, scene, img)
    t0 = time.perf_counter(); ctx = c.walk_rays(vp, scene, img); dt = time.perf_counter() - t0
    return img, ctx.total_rays, dt
a, ra, ta = run()
b, rb, tb = run(R.OPT_FAST)
diff = (a.view(np.uint32) != b.view(np.uint32)).any(axis=2)
print(f"exact: {ra} rays {ta*1e3:.1f} ms; fast: {rb} rays {tb*1e3:.1f} ms; speed-up {ta/tb:.2f}; differing pixels {int(diff.sum())} of {W*H}; max abs diff {np.nanmax(np.abs(a-b)):.4f}")
