#!/usr/bin/env python3
"""Development aid: step statistics of the octree trace kernel (counting build) on the bench scene."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from rust_raytrace_amd import raytrace as R, _ffi

W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 512
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
scene = R.canonical_scene(os.path.join(ROOT, "tests", "golden", "teapot_tri.obj"))
vp = R.canonical_viewport(W, H, 5, spp)
c = R.HipRayCaster(seed=1, options=R.OPT_COUNTERS)
img = np.zeros((H, W, 4), np.float32)

ctx = c.walk_rays(vp, scene, img)
print(ctx.stats)
# the resident scene handle lives inside the C++ caster; fetch the debug counters through a tiny helper
lib = _ffi.lib()
lib.rth_debug_counters.argtypes = [C.c_void_p, C.c_void_p]
out = (C.c_ulonglong * 16)()
lib.rth_debug_counters(scene.h, out)
d = list(out)
names = ["S steps", "S lanes", "L steps", "L lanes", "refills", "refill lanes", "edge blocks", "edge lanes", "S cycles", "L cycles", "refill cycles", "wave cycles"]
for n, v in zip(names, d):
    print(f"{n:14s} {v}")
rays = ctx.stats["rays"]
print("NOTE: dbg counters cover only the LAST batch/pass sequence of the call (ctrl is reset per batch)")
print(f"S lane util {d[1] / max(d[0] * 64, 1):.3f}   L lane util {d[3] / max(d[2] * 64, 1):.3f}   edge lanes/block {d[7] / max(d[6], 1):.2f}")
print(f"per ray: S steps {d[1] / rays:.1f}  L steps {d[3] / rays:.1f}  wave-steps per ray-wave: S {d[0] * 64 / rays:.1f} L {d[2] * 64 / rays:.1f}")
tot = max(d[11], 1)
print(f"shader-clock cycles of a wave (counting build, all passes of the last batch): SELECT steps {d[8] / tot:.3f}, LEAF steps {d[9] / tot:.3f}, "
      f"refills {d[10] / tot:.3f}, vote/rest {1 - (d[8] + d[9] + d[10]) / tot:.3f} of the wave's lifetime; "
      f"{d[8] / max(d[0], 1):.0f} cycles per SELECT step, {d[9] / max(d[2], 1):.0f} per LEAF step, {d[10] / max(d[4], 1):.0f} per refill")
print(f"t < 0 decidable from signs/exponents alone: {d[12]} of {ctx.stats['tri_tests']} plane tests of the last batch's lanes "
      f"({d[12] / max(ctx.stats['tri_tests'], 1):.3f} if the call was one batch); wave level: {d[13]} of {d[14]} (LEAF step, reference) slots "
      f"have EVERY working lane decidable ({d[13] / max(d[14], 1):.4f}); LEAF steps with all four slots so: {d[15]} of {d[2]} ({d[15] / max(d[2], 1):.4f})")
