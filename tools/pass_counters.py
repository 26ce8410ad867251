#!/usr/bin/env python3
"""Development aid: per-pass work counters and step statistics (RTMI_VERBOSE + counting build) on the bench scene."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RTMI_VERBOSE"] = "1"
os.environ["RTMI_STREAMS"] = "1"
import numpy as np
from rust_raytrace_amd import raytrace as R
W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
scene = R.canonical_scene(os.path.join(ROOT, "tests", "golden", "teapot_tri.obj"), gpu_build=0)
vp = R.canonical_viewport(W, H, 5, spp)
img = np.zeros((H, W, 4), np.float32)
R.HipRayCaster(seed=1, options=R.OPT_COUNTERS).walk_rays(vp, scene, img)
