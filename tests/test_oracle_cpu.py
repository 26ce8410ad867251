"""CPU: the oracle against known-answer tests and the committed golden fixtures."""
import json
import os

import numpy as np

from conftest import GOLDEN, OracleApi, assert_bits_equal, recipe_axis_box, recipe_canonical, recipe_circles

G = json.load(open(os.path.join(GOLDEN, "golden.json")))


def _orc():
    from oracle import orc
    return orc


def test_reference_face_collision_known_answer():
    # the reference's only hot-path-adjacent unit test (raytrace.rs:735-750)
    assert _orc().kat_face_collision() == 1


def test_philox_published_vectors():
    orc = _orc()
    for v in G["philox_kat"]:
        assert [int(x) for x in orc.philox4x32_10(v["ctr"], v["key"])] == v["out"]
    assert [int(x) for x in orc.rng_block(1, 5, 2, 3)] == G["rng_block_seed1_pixel5_sample2_block3"]


def test_u32_to_unit_float_is_rand_standard():
    orc = _orc()
    assert orc.u32_to_unit_f32(0) == 0.0
    assert orc.u32_to_unit_f32(0xFFFFFFFF) == float(np.float32(0xFFFFFF) * np.float32(2.0 ** -24))
    assert orc.u32_to_unit_f32(0x80000000) == 0.5
    assert orc.u32_to_unit_f32(0x000000FF) == 0.0  # low 8 bits are dropped


def test_canonical_scene_counts(canonical_pair):
    so, _ = canonical_pair
    assert so.num_tris() == G["canonical_num_tris"] == 6721  # 1 sentinel + 6320 teapot + 2*50*4 disc
    st = so.tree_stats()
    assert st == G["canonical_tree"]
    # SURVEY.md §8(a) float64 estimate: 35 156 / 169 864 / 2 936 429 (small f32 deviations expected)
    assert abs(st["inner"] - 35156) < 200 and abs(st["leaves"] - 169864) < 800 and abs(st["refs"] - 2936429) < 8000


def test_triangle_records_golden(canonical_pair):
    so, _ = canonical_pair
    z = np.load(os.path.join(GOLDEN, "canonical_triangle_records.npz"))
    rec, kinds, surf = so.triangles()
    assert_bits_equal(rec[z["idx"]], z["rec"], "triangle records")
    assert np.array_equal(kinds[z["idx"]], z["kinds"])
    assert_bits_equal(surf[z["idx"]], z["surf"], "surfaces")
    # the sentinel of main.rs:117: corners (1,0,0) (0,1,0) (0,0,1), centroid (1/3,1/3,1/3)
    assert np.allclose(rec[0, 0:3], 1.0 / 3.0, atol=1e-6)


def test_render_goldens():
    orc = _orc()
    api = OracleApi(orc)
    s = recipe_canonical(solid_teapot=True)(api)
    img, cn = s.render(64, 64, orc.canonical_viewport(64, 64), 5, 1, threads=8)
    assert_bits_equal(img, np.load(os.path.join(GOLDEN, "canonical_solid_64x64_spp1.npy")), "solid image")
    assert cn == G["canonical_solid_64x64_spp1"]
    o4, d4 = orc.primary_rays(64, 64, orc.canonical_viewport(64, 64), 1)
    tri, t, face, _ = s.trace(o4, d4)
    z = np.load(os.path.join(GOLDEN, "canonical_64x64_first_hits.npz"))
    assert np.array_equal(tri, z["tri"]) and np.array_equal(face, z["face"])
    assert_bits_equal(t, z["t"], "hit times")
    s = recipe_circles()(api)
    img, cn = s.render(64, 64, orc.canonical_viewport(64, 64), 5, 2, seed=3, threads=8)
    assert_bits_equal(img, np.load(os.path.join(GOLDEN, "circles_64x64_spp2_seed3.npy")), "circles image")
    assert cn == G["circles_64x64_spp2_seed3"] and s.tree_stats() == G["circles_tree"]


def test_seeded_render_golden_and_thread_independence(canonical_pair):
    orc = _orc()
    so, _ = canonical_pair
    vp = orc.canonical_viewport(32, 32)
    img1, cn1 = so.render(32, 32, vp, 5, 4, seed=1, threads=1)
    img8, cn8 = so.render(32, 32, vp, 5, 4, seed=1, threads=8)
    assert_bits_equal(img1, img8, "threads")
    assert cn1 == cn8 == G["canonical_32x32_spp4_seed1"]
    assert_bits_equal(img1, np.load(os.path.join(GOLDEN, "canonical_32x32_spp4_seed1.npy")), "seeded image")
    other, _ = so.render(32, 32, vp, 5, 4, seed=2, threads=8)
    assert not np.array_equal(other, img1)
    # row bands reproduce the full image (the RNG is keyed by pixel, not by thread or order)
    band, _ = so.render(32, 32, vp, 5, 4, seed=1, row0=10, nrows=7, threads=3)
    assert_bits_equal(band, img1[10:17], "row band")


def test_octree_matches_linear_list_on_primary_rays(canonical_pair):
    # on the canonical view the octree closest hit equals the brute-force closest hit
    orc = _orc()
    so, _ = canonical_pair
    lin = recipe_canonical(accel="trivial")(OracleApi(orc))
    o4, d4 = orc.primary_rays(48, 48, orc.canonical_viewport(48, 48), 1)
    tri_o, t_o, _, _ = so.trace(o4, d4)
    tri_l, t_l, _, cn = lin.trace(o4, d4)
    assert np.array_equal(tri_o, tri_l)
    assert cn["tri_tests"] == 6720 * len(tri_l)  # every ray tests every triangle once


def test_depth_zero_and_sky():
    orc = _orc()
    s = recipe_axis_box()(OracleApi(orc))
    vp = orc.canonical_viewport(8, 8)
    img, cn = s.render(8, 8, vp, 0, 1)
    assert cn["rays"] == 0 and not img.any()  # project_ray returns black at depth 0 (raytrace.rs:1261-1263)
    # a camera looking away from everything sees only the sky colour (128,180,255)/255
    vp = orc.create_viewport(8, 8, (1.0, 1.0), [0.0, 0.0, 100.0], orc.unit([0.0, 0.0, 1.0]), 90.0, 0.0)  # scene is behind
    img, cn = s.render(8, 8, vp, 5, 1)
    sky = orc.make_color(128, 180, 255)
    assert cn["rays"] == 64 and np.array_equal(img[..., :3], np.broadcast_to(sky, (8, 8, 3))) and not img[..., 3].any()


def test_quantize_is_rust_as_u8():
    orc = _orc()
    x = np.array([[0.0, 1.0, 0.5, 0], [-1.0, 2.0, np.nan, 0], [0.999, 1e-9, np.inf, 0], [254.9 / 255, 255.1 / 255, -np.inf, 0]], np.float32)
    q = orc.quantize(x)
    assert q.tolist() == [[0, 255, 127], [0, 255, 0], [254, 0, 255], [254, 255, 0]]


def test_render_window_equals_rows():
    """orc_render_window (pixel windows for the full-size config-4 checks) is the same per-pixel loop as orc_render."""
    import pytest
    from conftest import TEAPOT_TRI
    orc = _orc()
    so = orc.canonical_scene(TEAPOT_TRI, maxdepth=6)
    vo = orc.canonical_viewport(64, 48)
    rows, cr = so.render(64, 48, vo, 5, 3, seed=9, row0=20, nrows=6, threads=4)
    win, cw = so.render_window(64, 48, vo, 5, 3, 20, 6, 10, 30, seed=9, threads=3)
    assert_bits_equal(rows[:, 10:40], win, "window")
    full, cf = so.render_window(64, 48, vo, 5, 3, 20, 6, 0, 64, seed=9, threads=2)
    assert_bits_equal(rows, full, "full-width window")
    assert cf == cr and cw["rays"] < cr["rays"]
    with pytest.raises(RuntimeError):
        so.render_window(64, 48, vo, 5, 3, 44, 6, 0, 64)


def test_analytic_sphere_definition_known_answers():
    """a12: the analytic sphere is a BUILD-DEFINED primitive (the reference at this revision only tessellates spheres,
    raytrace.rs:464-529) -- parity with the Rust binary is unpinned by construction.  These known answers pin the
    definition itself (oracle/rt_oracle.cpp, struct Sphere): exact roots on exactly representable inputs, the inside
    (Back-face) hit, the `t < 0` miss rule, strict-closer replacement in list order."""
    orc = _orc()
    s = orc.Scene(with_dummy=True)
    m = orc.Surface(orc.SOLID, orc.make_color(10, 20, 30))
    s.add_analytic_sphere([0.0, 0.0, 5.0], 1.0, m)
    s.add_analytic_sphere([0.0, 0.0, 9.0], 2.0, m)
    s.add_analytic_sphere([0.0, 0.0, 5.0], 1.0, m)      # a duplicate of sphere 1: never wins (strict <)
    o = np.array([[0, 0, 0, 0], [0, 0, 5, 0], [0, 0, 20, 0], [3, 0, 0, 0], [0, 0, 6.5, 0], [1, 0, 0, 0]], np.float32)
    d = np.array([[0, 0, 1, 0], [0, 0, 1, 0], [0, 0, 1, 0], [0, 0, 1, 0], [0, 0, 1, 0], [0, 0, 1, 0]], np.float32)
    idx, t, face = s.trace_spheres(o, d)
    assert idx.tolist() == [1, 1, 0, 0, 2, 1]            # from inside sphere 1; behind every sphere; off axis; between; tangent
    assert t.tolist() == [4.0, 1.0, 0.0, 0.0, 0.5, 5.0]
    assert face.tolist() == [0, 1, 0, 0, 0, 0]           # 1 = Back: the far root seen from inside
