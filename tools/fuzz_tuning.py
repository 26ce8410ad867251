#!/usr/bin/env python3
"""Development aid: one canonical-scene frame under random tuning (batch size, streams, waves per CU, refill thresholds,
XCD mode) must equal the oracle's image bit for bit -- no tuning knob may change a pixel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import build_pair, recipe_canonical, assert_bits_equal
from oracle import orc
from rust_raytrace_amd import raytrace as R
so, sp = build_pair(recipe_canonical())
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(11)
for it in range(n):
    w, h, spp, depth = int(rng.integers(8, 80)), int(rng.integers(3, 70)), int(rng.integers(1, 5)), int(rng.integers(1, 6))
    vo = orc.canonical_viewport(w, h)
    vp = R.canonical_viewport(w, h, depth, spp)
    ref, cn = so.render(w, h, vo, depth, spp, seed=it, threads=8)
    tune = {"batch_paths": int(rng.integers(1, 4 * w * h * spp)), "streams": int(rng.integers(1, 5)), "subtile_min_paths": 1,
            "oct_waves_per_cu": int(rng.choice([0, 1, 3, 8, 16, 24, 32])), "refill_min0": int(rng.integers(1, 65)),
            "refill_min": int(rng.integers(1, 65)), "xcd_aware": int(rng.integers(0, 3)), "pipeline": int(rng.integers(1, 4))}
    img = np.zeros((h, w, 4), np.float32)
    ctx = R.HipRayCaster(seed=it, tuning=tune).walk_rays(vp, sp, img, 1, False)
    assert_bits_equal(ref, img, f"{w}x{h} spp {spp} depth {depth} tuning {tune}")
    assert ctx.total_rays == cn["rays"], tune
print(f"{n} random tunings: every frame equals the oracle's")
