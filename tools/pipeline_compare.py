#!/usr/bin/env python3
"""Development aid: config 3 frame (or one rank's tile of N), per-pass pipeline vs fused path kernels, per stream count:
device time of the frame, sum of the closest-hit launches, the primary / bounce kernels apart."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rust_raytrace_amd import raytrace as R, dist as rd
world = int(sys.argv[1]) if len(sys.argv) > 1 else 1
PIPES = [int(x) for x in os.environ.get("PIPES", "1,2").split(",")]
streams = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 3]
extra = eval(sys.argv[3]) if len(sys.argv) > 3 else {}
W = H = 2048
scene = R.canonical_scene(os.path.join(ROOT, "tests", "golden", "teapot_tri.obj"), gpu_build=0)
vp = R.canonical_viewport(W, H, 5, 64)
st = torch.cuda.current_stream().cuda_stream
tile = rd.rank_tile(min(3, world - 1), world, H, 16)
buf = torch.zeros((tile[1], W, 4), dtype=torch.float32, device="cuda:0")
for pipe in PIPES:
    for ns in streams:
        c = R.HipRayCaster(seed=1, tuning=dict({"pipeline": pipe, "streams": ns}, **extra))
        c.walk_tile_device(vp, scene, tile, buf.data_ptr(), st)
        best = None
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx = c.walk_tile_device(vp, scene, tile, buf.data_ptr(), st)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
            if best is None or dt < best[0]:
                best = (dt, ctx)
        dt, ctx = best
        s = ctx.stats
        print(f"tile 1/{world} pipeline {pipe} streams {ns} {extra}: wall {dt:7.1f} ms  kernel {s['kernel_ms']:7.1f}  trace-sum {s['trace_ms']:7.1f}  "
              f"primary {s['primary_ms']:7.1f}  bounce {s['bounce_ms']:7.1f}  launches {s['trace_launches']}  {ctx.total_rays / dt / 1e3:7.1f} Mrays/s", flush=True)
