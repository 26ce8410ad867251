#!/usr/bin/env python3
"""Development aid: per-pass trace times (RTMI_VERBOSE) of ONE rank's tile of config 3 at N ranks, rendered on this GPU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RTMI_VERBOSE"] = "1"
import torch
from rust_raytrace_amd import raytrace as R, dist as rd
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
W = H = 2048
scene = R.canonical_scene(os.path.join(ROOT, "tests", "golden", "teapot_tri.obj"))
vp = R.canonical_viewport(W, H, 5, 64)
c = R.HipRayCaster(seed=1)
c.upload(scene)
st = torch.cuda.current_stream().cuda_stream
tile = rd.rank_tile(3, world, H, 16)
buf = torch.zeros((tile[1], W, 4), dtype=torch.float32, device="cuda:0")
os.environ.pop("RTMI_VERBOSE")
c.walk_tile_device(vp, scene, tile, buf.data_ptr(), st)
torch.cuda.synchronize(); t0 = time.perf_counter()
ctx = c.walk_tile_device(vp, scene, tile, buf.data_ptr(), st)
torch.cuda.synchronize(); print(f"tile of rank 3 of {world}: {(time.perf_counter() - t0) * 1e3:.1f} ms, kernel_ms {ctx.stats['kernel_ms']:.1f}, trace_ms (sum over streams) {ctx.stats['trace_ms']:.1f}")
