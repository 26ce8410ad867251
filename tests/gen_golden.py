#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/ from the CPU oracle.

The reference (Rust, nightly, un-vendored crates) cannot be built in this pipeline and holds no
golden vectors for the render path, so these fixtures pin the ORACLE's outputs:
they protect against regressions of the oracle itself and let the GPU tests run against files even
if the oracle library were unavailable.  Run from the repo root:  python tests/gen_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import GOLDEN, OracleApi, recipe_axis_box, recipe_canonical, recipe_circles  # noqa: E402
from oracle import orc  # noqa: E402


def main():
    api = OracleApi(orc)
    out = {}
    # (1) deterministic image: canonical scene, Solid teapot, spp 1, 64x64 (main.rs default size) — no RNG involved
    s = recipe_canonical(solid_teapot=True)(api)
    vp = orc.canonical_viewport(64, 64)
    img, cn = s.render(64, 64, vp, 5, 1, threads=8)
    np.save(os.path.join(GOLDEN, "canonical_solid_64x64_spp1.npy"), img)
    out["canonical_solid_64x64_spp1"] = cn
    # (2) first-hit (tri, t, face) map for the same primary rays
    o4, d4 = orc.primary_rays(64, 64, vp, 1)
    tri, t, face, _ = s.trace(o4, d4)
    np.savez_compressed(os.path.join(GOLDEN, "canonical_64x64_first_hits.npz"), tri=tri, t=t, face=face)
    # (3) seeded Matte/Reflective image 32x32 x 4 spp, canonical scene, seed 1
    s = recipe_canonical()(api)
    vp = orc.canonical_viewport(32, 32)
    img, cn = s.render(32, 32, vp, 5, 4, seed=1, threads=8)
    np.save(os.path.join(GOLDEN, "canonical_32x32_spp4_seed1.npy"), img)
    out["canonical_32x32_spp4_seed1"] = cn
    out["canonical_tree"] = s.tree_stats()
    # (4) make_triangle records of hand-picked faces (first 4 teapot faces, last disc faces, sentinel)
    rec, kinds, surf = s.triangles()
    pick = [0, 1, 2, 3, 4, 3000, 6320, 6321, 6322, 6323, 6324, 6720]
    np.savez_compressed(os.path.join(GOLDEN, "canonical_triangle_records.npz"), idx=np.array(pick), rec=rec[pick],
                        kinds=kinds[pick], surf=surf[pick])
    out["canonical_num_tris"] = int(rec.shape[0])
    # (5) circles scene (BASELINE config 1 at reduced size) and the axis-aligned edge-case scene
    s = recipe_circles()(api)
    vp = orc.canonical_viewport(64, 64)
    img, cn = s.render(64, 64, vp, 5, 2, seed=3, threads=8)
    np.save(os.path.join(GOLDEN, "circles_64x64_spp2_seed3.npy"), img)
    out["circles_64x64_spp2_seed3"] = cn
    out["circles_tree"] = s.tree_stats()
    # (6) RNG: Philox4x32-10 published known-answer vectors (Random123 kat_vectors) and the draw mapping
    out["philox_kat"] = [
        {"ctr": [0, 0, 0, 0], "key": [0, 0], "out": [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]},
        {"ctr": [0xffffffff] * 4, "key": [0xffffffff] * 2, "out": [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]},
        {"ctr": [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], "key": [0xa4093822, 0x299f31d0],
         "out": [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]}]
    out["rng_block_seed1_pixel5_sample2_block3"] = [int(x) for x in orc.rng_block(1, 5, 2, 3)]
    json.dump(out, open(os.path.join(GOLDEN, "golden.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1)[:600])


if __name__ == "__main__":
    main()
