"""-m gpu: the HIP path (through the C ABI) against the CPU oracle, bit for bit."""
import os

import numpy as np
import pytest

from conftest import (TEAPOT, assert_bits_equal, build_pair, recipe_axis_box, recipe_canonical, recipe_circles, recipe_grid)

pytestmark = pytest.mark.gpu


def _orc():
    from oracle import orc
    return orc


def _R():
    from rust_raytrace_amd import raytrace as R
    return R


def _viewports(w, h, maxdepth, spp):
    return _orc().canonical_viewport(w, h), _R().canonical_viewport(w, h, maxdepth, spp)


def _render_both(pair, w, h, maxdepth, spp, seed=1, options=0):
    so, sp = pair
    vo, vp = _viewports(w, h, maxdepth, spp)
    assert_bits_equal(vo, vp.vp12, "viewport")
    ref, cn = so.render(w, h, vo, maxdepth, spp, seed=seed, threads=8)
    img = np.zeros((h, w, 4), np.float32)
    ctx = _R().HipRayCaster(seed=seed, options=options).walk_rays(vp, sp, img, 1, False)
    return ref, cn, img, ctx


def _compare_hits(so, sp, o4, d4):
    tri_o, t_o, face_o, cn = so.trace(o4, d4)
    tri_g, t_g, face_g, st = _R().HipRayCaster(options=_R().OPT_COUNTERS).trace(sp, o4, d4)
    bad = np.nonzero(tri_o != tri_g)[0]
    assert len(bad) == 0, f"{len(bad)} hit ids differ, first ray {bad[:5]}: {tri_o[bad[:5]]} vs {tri_g[bad[:5]]}"
    hit = tri_o != 0
    assert_bits_equal(t_o[hit], t_g[hit], "hit time")
    assert np.array_equal(face_o[hit], face_g[hit]), "face"
    for k in ("rays", "box_tests", "tri_tests", "full_tests", "nodes", "leaves"):
        assert st[k] == cn[k], f"work counter {k}: device {st[k]} vs oracle {cn[k]}"
    return tri_o, cn


def test_trace_primary_rays_canonical(canonical_pair):
    so, sp = canonical_pair
    vo, _ = _viewports(64, 64, 5, 1)
    o4, d4 = _orc().primary_rays(64, 64, vo, 1)
    tri, _ = _compare_hits(so, sp, o4, d4)
    assert (tri != 0).sum() > 500  # the teapot is in view


def test_trace_random_rays_canonical(canonical_pair):
    so, sp = canonical_pair
    rng = np.random.default_rng(7)
    n = 20000
    o4 = np.zeros((n, 4), np.float32)
    d4 = np.zeros((n, 4), np.float32)
    o4[:, :3] = rng.uniform(-6, 6, (n, 3)) + np.array([0, 0, 6])
    d = rng.normal(size=(n, 3))
    d4[:, :3] = d / np.linalg.norm(d, axis=1, keepdims=True)
    _compare_hits(so, sp, o4, d4)


def test_trace_edge_case_rays():
    so, sp = build_pair(recipe_axis_box())
    rays = []
    for ox in (-1.0, -0.5, 0.0, 0.25, 1.0):
        for oy in (-1.0, 0.0, 0.25, 0.5):
            for dvec in ((0, 0, 1), (0, 0, -1), (1, 0, 0), (0, 1, 0), (0, -1, 0), (-1, 0, 0), (0.6, 0, 0.8), (0, 0.6, 0.8),
                         (-0.0, 0.0, 1.0), (1e-30, 0, 1), (0.57735026, 0.57735026, 0.57735026)):
                rays.append(((ox, oy, 3.5, 0.0), (*dvec, 0.0)))
                rays.append(((ox, oy, 5.0, 0.0), (*dvec, 0.0)))     # origin on the z = 5 plane
                rays.append(((-1.0, oy, 4.0, 0.0), (*dvec, 0.0)))   # origin on the x = -1 plane
    # poisoned rays: NaN / inf components and a NaN lane 3
    rays += [((0, 0, 0, 0), (np.nan, 0, 1, 0)), ((np.nan, 0, 0, 0), (0, 0, 1, 0)), ((0, 0, 0, np.nan), (0, 0, 1, 0)),
             ((0, 0, 0, 0), (0, 0, 1, np.nan)), ((np.inf, 0, 0, 0), (0, 0, 1, 0)), ((0, 0, 0, 0), (0, 0, 0, 0))]
    o4 = np.array([r[0] for r in rays], np.float32)
    d4 = np.array([r[1] for r in rays], np.float32)
    _compare_hits(so, sp, o4, d4)


def test_render_canonical_solid_spp1_is_rng_free():
    # spp == 1 and all-Solid teapot: the reference itself is deterministic here
    pair = build_pair(recipe_canonical(solid_teapot=True))
    ref, cn, img, ctx = _render_both(pair, 64, 64, 5, 1)
    assert_bits_equal(ref, img, "image")
    assert ctx.total_rays == cn["rays"]


def test_render_canonical_spp1(canonical_pair):
    ref, cn, img, ctx = _render_both(canonical_pair, 64, 64, 5, 1)
    assert_bits_equal(ref, img, "image")
    assert ctx.total_rays == cn["rays"]


@pytest.mark.parametrize("seed", [1, 2, 0xDEADBEEFCAFE])
def test_render_canonical_spp4_seeded(canonical_pair, seed):
    ref, cn, img, ctx = _render_both(canonical_pair, 48, 32, 5, 4, seed=seed)
    assert_bits_equal(ref, img, "image")
    assert ctx.total_rays == cn["rays"]


def test_generic_tree_kernel_same_results(canonical_pair):
    # RTMI_OPT_GENERIC: the kernel for trees that are not exact octrees (loads every child box)
    R = _R()
    ref, cn, img, ctx = _render_both(canonical_pair, 40, 24, 5, 3, seed=11, options=R.OPT_GENERIC | R.OPT_COUNTERS)
    assert_bits_equal(ref, img, "image")
    for k in ("rays", "box_tests", "tri_tests", "full_tests", "nodes", "leaves"):
        assert ctx.stats[k] == cn[k], k


def test_render_counters_match_oracle(canonical_pair):
    ref, cn, img, ctx = _render_both(canonical_pair, 32, 32, 5, 2, options=_R().OPT_COUNTERS)
    assert_bits_equal(ref, img, "image")
    for k in ("rays", "box_tests", "tri_tests", "full_tests", "nodes", "leaves"):
        assert ctx.stats[k] == cn[k], k


@pytest.mark.parametrize("maxdepth", [0, 1, 2, 8])
def test_render_depth_limits(canonical_pair, maxdepth):
    ref, cn, img, ctx = _render_both(canonical_pair, 32, 32, maxdepth, 2)
    assert_bits_equal(ref, img, "image")
    assert ctx.total_rays == cn["rays"]


def test_render_circles_config1(circles_pair):
    # BASELINE config 1: 256x256, 1 spp (plumbing)
    ref, cn, img, ctx = _render_both(circles_pair, 256, 256, 5, 1)
    assert_bits_equal(ref, img, "image")
    assert ctx.total_rays == cn["rays"]


def test_render_circles_seeded(circles_pair):
    ref, cn, img, ctx = _render_both(circles_pair, 64, 64, 5, 8, seed=5)
    assert_bits_equal(ref, img, "image")


@pytest.mark.parametrize("generic", [False, True])
def test_render_linear_list_config2_small(generic):
    # BASELINE config 2 shape: teapot.obj, trivial bounding box (one leaf = linear list), reduced size.
    # generic=False runs the LDS-streamed k_trace_linear, True the generic-tree kernel on the same scene.
    R = _R()
    pair = build_pair(recipe_canonical(accel="trivial", obj=TEAPOT))
    ref, cn, img, ctx = _render_both(pair, 33, 31, 5, 2, options=R.OPT_COUNTERS | (R.OPT_GENERIC if generic else 0))
    assert_bits_equal(ref, img, "image")
    for k in ("rays", "box_tests", "tri_tests", "full_tests", "nodes", "leaves"):
        assert ctx.stats[k] == cn[k], k
    assert cn["tri_tests"] == 6720 * cn["rays"]


def test_full_size_config2_rows_vs_oracle():
    """BASELINE config 2 at its STATED size: teapot.obj, trivial bounding box (build_trivial_bounding_box,
    raytrace.rs:847-856: one list of 6 720 triangles scanned in order, raytrace.rs:1012-1050), 512 x 512 @ 16 spp, depth 5,
    seed 1.  Four rows of the frame -- sky, teapot lid, teapot body + mirror disks, below the teapot -- image bits and
    all six work counters against the oracle (k_trace_linear, the LDS-streamed list kernel)."""
    import os
    orc, R = _orc(), _R()
    so, sp = build_pair(recipe_canonical(accel="trivial", obj=TEAPOT))
    W = H = 512
    vo = orc.canonical_viewport(W, H)
    vp = R.canonical_viewport(W, H, 5, 16)
    threads = max(1, len(os.sched_getaffinity(0)))
    c = R.HipRayCaster(seed=1, options=R.OPT_COUNTERS)
    bounced = 0
    for row in (3, 180, 256, 340):
        ref, cn = so.render(W, H, vo, 5, 16, seed=1, row0=row, nrows=1, threads=threads)
        got = np.zeros((1, W, 4), np.float32)
        ctx = c.walk_rows(vp, sp, row, 1, got)
        assert_bits_equal(ref, got, f"config 2 row {row}")
        for k in ("rays", "box_tests", "tri_tests", "full_tests", "nodes", "leaves"):
            assert ctx.stats[k] == cn[k], (row, k)
        assert cn["tri_tests"] == 6720 * cn["rays"]
        bounced += cn["rays"] - W * 16
    assert bounced > 0
    # the whole frame through the default (uncounted) kernel reproduces those rows
    img = np.zeros((H, W, 4), np.float32)
    R.HipRayCaster(seed=1).walk_rays(vp, sp, img, 1, False)
    for row in (3, 256):
        ref, _ = so.render(W, H, vo, 5, 16, seed=1, row0=row, nrows=1, threads=threads)
        assert_bits_equal(ref[0], img[row], f"config 2 whole frame, row {row}")


def test_trace_linear_list_edge_rays():
    # the axis-aligned scene as ONE list: NaN / inf hit times and exact ties through k_trace_linear
    from conftest import OracleApi, ProductApi
    orc, R = _orc(), _R()

    def relist(api):
        s = recipe_axis_box()(api)
        s.build_trivial_bounding_box([0.0, 0.0, 4.0], 4.0)
        return s
    so, sp = relist(OracleApi(orc)), relist(ProductApi(R))
    rng = np.random.default_rng(3)
    n = 700  # not a multiple of 256: ragged last block
    o4 = np.zeros((n, 4), np.float32)
    d4 = np.zeros((n, 4), np.float32)
    o4[:, :3] = rng.integers(-2, 3, (n, 3)) * 0.5 + np.array([0, 0, 4])
    d = rng.integers(-1, 2, (n, 3)).astype(np.float32)
    d[(d == 0).all(axis=1)] = [0, 0, 1]
    d4[:, :3] = d / np.linalg.norm(d, axis=1, keepdims=True)
    _compare_hits(so, sp, o4, d4)


def test_render_axis_aligned_scene():
    pair = build_pair(recipe_axis_box())
    so, sp = pair
    orc, R = _orc(), _R()
    # camera looking straight down +z through the box: many rays parallel to walls
    w = h = 33  # odd: the centre column has dir.x == 0 exactly
    vo = orc.create_viewport(w, h, (1.0, 1.0), [0.0, 0.0, 0.0], orc.unit([0.0, 0.0, 1.0]), 90.0, 0.0)
    vp = R.create_viewport((w, h), (1.0, 1.0), [0.0, 0.0, 0.0], R.unit([0.0, 0.0, 1.0]), 90.0, 0.0, 5, 1)
    assert_bits_equal(vo, vp.vp12, "viewport")
    ref, cn = so.render(w, h, vo, 5, 1)
    img = np.zeros((h, w, 4), np.float32)
    R.HipRayCaster().walk_rays(vp, sp, img, 1, False)
    assert_bits_equal(ref, img, "image")


def test_row_partition_invariance(canonical_pair):
    so, sp = canonical_pair
    R = _R()
    vp = R.canonical_viewport(40, 40, 5, 3)
    whole = np.zeros((40, 40, 4), np.float32)
    c = R.HipRayCaster(seed=9)
    total = c.walk_rays(vp, sp, whole, 1, False).total_rays
    parts = np.zeros_like(whole)
    rays = 0
    for row0, n in ((0, 7), (7, 20), (27, 13)):
        rays += c.walk_rows(vp, sp, row0, n, parts[row0:row0 + n]).total_rays
    assert_bits_equal(whole, parts, "row tiles")
    assert rays == total
    empty = np.zeros((0, 40, 4), np.float32)
    assert c.walk_rows(vp, sp, 5, 0, empty).total_rays == 0


def test_small_batches_same_image(canonical_pair):
    so, sp = canonical_pair
    R = _R()
    vp = R.canonical_viewport(32, 32, 5, 4)
    a = np.zeros((32, 32, 4), np.float32)
    R.HipRayCaster(seed=3).walk_rays(vp, sp, a, 1, False)
    b = np.zeros_like(a)
    R.HipRayCaster(seed=3, tuning={"batch_paths": 1000}).walk_rays(vp, sp, b, 1, False)  # forces many ragged batches
    assert_bits_equal(a, b, "batched")
    # launch tuning never changes a pixel: waves per CU, refill thresholds, ray-queue ranges
    for tn in ({"oct_waves_per_cu": 3, "refill_min0": 1, "refill_min": 64}, {"xcd_aware": 0}, {"xcd_aware": 1, "refill_min": 8}, {"xcd_aware": 2, "refill_min0": 17}):
        c = np.zeros_like(a)
        R.HipRayCaster(seed=3, tuning=tn).walk_rays(vp, sp, c, 1, False)
        assert_bits_equal(a, c, f"tuning {tn}")
    with pytest.raises(RuntimeError):
        R.HipRayCaster(seed=3, tuning={"streams": 9}).walk_rays(vp, sp, b, 1, False)  # more than RTMI_MAX_STREAMS


def test_quantize_matches_oracle(canonical_pair):
    rng = np.random.default_rng(0)
    x = rng.uniform(-0.2, 1.2, (1000, 4)).astype(np.float32)
    x[0] = [np.nan, np.inf, -np.inf, 0]
    x[1] = [1.0, 0.999999, 0.0, 0]
    assert np.array_equal(_orc().quantize(x), _R().quantize(x))


def test_error_behaviour(canonical_pair):
    so, sp = canonical_pair
    R = _R()
    vp = R.canonical_viewport(16, 16, 5, 0)  # spp 0
    with pytest.raises(RuntimeError):
        R.HipRayCaster().walk_rays(vp, sp, np.zeros((16, 16, 4), np.float32), 1, False)
    vp = R.canonical_viewport(16, 16, 5, 1)
    with pytest.raises(RuntimeError):
        R.HipRayCaster().walk_rows(vp, sp, 10, 10, np.zeros((10, 16, 4), np.float32))  # rows outside the image
    empty = R.Scene()
    with pytest.raises(RuntimeError):
        R.HipRayCaster().walk_rays(vp, empty, np.zeros((16, 16, 4), np.float32), 1, False)  # no bounding box


def test_striped_tiles_equal_whole_image(canonical_pair):
    import torch
    from rust_raytrace_amd import dist as rd
    so, sp = canonical_pair
    R = _R()
    H, W = 40, 24
    vp = R.canonical_viewport(W, H, 5, 2)
    whole = np.zeros((H, W, 4), np.float32)
    c = R.HipRayCaster(seed=4)
    total = c.walk_rays(vp, sp, whole, 1, False).total_rays
    for world, S in ((2, 4), (3, 8), (8, 2)):
        frame = np.zeros_like(whole)
        rays = 0
        for r in range(world):
            tile = rd.rank_tile(r, world, H, S)
            buf = torch.zeros((tile[1], W, 4), dtype=torch.float32, device="cuda:0")
            rays += c.walk_tile_device(vp, sp, tile, buf.data_ptr(), torch.cuda.current_stream().cuda_stream).total_rays
            torch.cuda.synchronize()
            frame[rd.tile_rows(tile, H)] = buf.cpu().numpy()
        assert_bits_equal(whole, frame, f"world {world} stripes {S}")
        assert rays == total


def test_gpu_against_committed_golden_fixtures(canonical_pair):
    """The HIP path against files (tests/golden, written by tests/gen_golden.py from the oracle)."""
    import os
    from conftest import GOLDEN
    R = _R()
    _, sp = canonical_pair
    img = np.zeros((32, 32, 4), np.float32)
    R.HipRayCaster(seed=1).walk_rays(R.canonical_viewport(32, 32, 5, 4), sp, img, 1, False)
    assert_bits_equal(img, np.load(os.path.join(GOLDEN, "canonical_32x32_spp4_seed1.npy")), "seeded golden")
    _, sps = build_pair(recipe_canonical(solid_teapot=True))
    img = np.zeros((64, 64, 4), np.float32)
    R.HipRayCaster().walk_rays(R.canonical_viewport(64, 64, 5, 1), sps, img, 1, False)
    assert_bits_equal(img, np.load(os.path.join(GOLDEN, "canonical_solid_64x64_spp1.npy")), "solid golden")
    z = np.load(os.path.join(GOLDEN, "canonical_64x64_first_hits.npz"))
    o4, d4 = _orc().primary_rays(64, 64, _orc().canonical_viewport(64, 64), 1)
    tri, t, face, _ = R.HipRayCaster().trace(sps, o4, d4)
    assert np.array_equal(tri, z["tri"])
    hit = tri != 0
    assert_bits_equal(t[hit], z["t"][hit], "hit time")
    assert np.array_equal(face[hit], z["face"][hit])


def test_full_size_properties_config3():
    """BASELINE config 3 at full size (2048x2048 @ 64 spp is the bench; here 2048x2048 @ 2 spp to stay in
    seconds): size-independent properties — a striped re-render reproduces the frame bit for bit, a checksum
    of per-tile checksums agrees, the ray count is partition-invariant, every pixel is finite in [0, 1]."""
    import torch
    from rust_raytrace_amd import dist as rd
    R = _R()
    s = R.canonical_scene(__import__("conftest").TEAPOT_TRI)
    W = H = 2048
    vp = R.canonical_viewport(W, H, 5, 2)
    c = R.HipRayCaster(seed=1)
    whole = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    total = c.walk_tile_device(vp, s, (0, H, H, 0), whole.data_ptr(), st).total_rays
    torch.cuda.synchronize()
    assert torch.isfinite(whole).all() and whole.min() >= 0 and whole.max() <= 1 and not whole[..., 3].any()
    assert H * W * 2 <= total <= H * W * 2 * 5
    frame = torch.zeros_like(whole)
    rays = 0
    for r in range(4):
        tile = rd.rank_tile(r, 4, H, 16)
        buf = torch.zeros((tile[1], W, 4), dtype=torch.float32, device="cuda:0")
        rays += c.walk_tile_device(vp, s, tile, buf.data_ptr(), st).total_rays
        torch.cuda.synchronize()
        frame[torch.as_tensor(rd.tile_rows(tile, H), device="cuda:0")] = buf
    assert rays == total
    assert torch.equal(frame.view(torch.int32), whole.view(torch.int32))
    # the centre of the image shows the teapot (orange Matte mixed with bounces), the top-left corner the sky
    sky = torch.tensor(_orc().make_color(128, 180, 255), device="cuda:0")
    assert torch.equal(whole[0, 0, :3], sky)
    assert not torch.equal(whole[H // 2 + 100, W // 2, :3], sky)


def test_render_grid_config5_reduced():
    # BASELINE config 5 scene (8 teapot instances, 50 561 triangles) with a shallower octree so that the
    # single-threaded oracle builds it in seconds; mixed Matte / Reflective / Solid instances
    pair = build_pair(recipe_grid(maxdepth=7, minobjs=19))
    ref, cn, img, ctx = _render_both(pair, 48, 48, 5, 2, seed=21, options=_R().OPT_COUNTERS)
    assert_bits_equal(ref, img, "image")
    for k in ("rays", "box_tests", "tri_tests", "full_tests", "nodes", "leaves"):
        assert ctx.stats[k] == cn[k], k


_SOUP_SEEDS = [1, 2, 3, 4] + [int(x) for x in os.environ.get("RTMI_TEST_EXTRA_SOUP_SEEDS", "").split(",") if x]


@pytest.mark.parametrize("seed", _SOUP_SEEDS)
def test_random_triangle_soups(seed):
    """Random scenes: triangle soup with random surfaces and edge thickness, random octree parameters and camera.
    Exercises boxes with many colliding children, leaves of all sizes, deep and shallow trees."""
    from conftest import OracleApi, ProductApi
    orc, R = _orc(), _R()
    rng = np.random.default_rng(1000 + seed)
    ntri = int(rng.integers(40, 400))
    centre = rng.uniform(-3, 3, (ntri, 3)) + np.array([0, 0, 8.0])
    pts = (centre[:, None, :] + rng.normal(scale=rng.uniform(0.2, 1.2), size=(ntri, 3, 3))).astype(np.float32)
    kinds = rng.integers(0, 3, ntri)
    cols = rng.integers(0, 256, (ntri, 3))
    alphas = rng.uniform(0.05, 0.95, ntri)
    scat = rng.uniform(0.0, 0.3, ntri)
    edges = rng.choice([0.0, 0.05, 0.3, -1.0], ntri)
    maxdepth, minobjs = int(rng.integers(2, 9)), int(rng.integers(2, 24))

    def recipe(api):
        s = api.scene()
        for i in range(ntri):
            c = tuple(int(x) for x in cols[i])
            surf = (api.solid(c), api.matte(c, float(alphas[i])), api.reflective(float(scat[i]), c, float(alphas[i])))[kinds[i]]
            try:
                api.add_triangle(s, pts[i], surf, float(edges[i]))
            except RuntimeError:
                pass  # degenerate triangle: rejected identically by both implementations
        s.populate_triangle_numbers()
        s.build_bounding_box([0.0, 0.0, 8.0], 8.0, maxdepth, minobjs)
        return s
    so, sp = recipe(OracleApi(orc)), recipe(ProductApi(R))
    assert so.num_tris() == sp.num_tris()
    pos = rng.uniform(-1, 1, 3).astype(np.float32)
    aim = [float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.3, 0.3)), 1.0]
    assert_bits_equal(orc.unit(aim), R.unit(aim), "unit")
    w, h, spp, depth = 40, 28, int(rng.integers(1, 5)), int(rng.integers(1, 7))
    vo = orc.create_viewport(w, h, (1.0, 0.7), pos, orc.unit(aim), 75.0, 0.1)
    vp = R.create_viewport((w, h), (1.0, 0.7), pos, R.unit(aim), 75.0, 0.1, depth, spp)  # unit() once on either side
    assert_bits_equal(vo, vp.vp12, "viewport")
    ref, cn = so.render(w, h, vo, depth, spp, seed=seed, threads=8)
    img = np.zeros((h, w, 4), np.float32)
    ctx = R.HipRayCaster(seed=seed, options=R.OPT_COUNTERS).walk_rays(vp, sp, img, 1, False)
    assert_bits_equal(ref, img, f"image (ntri {ntri}, octree ({maxdepth},{minobjs}), spp {spp}, depth {depth})")
    for k in ("rays", "box_tests", "tri_tests", "full_tests", "nodes", "leaves"):
        assert ctx.stats[k] == cn[k], k
    # the kernels without the counting code are separate instantiations: the same image from them too
    img2 = np.zeros((h, w, 4), np.float32)
    ctx2 = R.HipRayCaster(seed=seed).walk_rays(vp, sp, img2, 1, False)
    assert_bits_equal(ref, img2, f"image, plain build (ntri {ntri}, octree ({maxdepth},{minobjs}), spp {spp}, depth {depth})")
    assert ctx2.total_rays == cn["rays"]


_CAMERA_SEEDS = [1, 2, 3] + [int(x) for x in os.environ.get("RTMI_TEST_EXTRA_CAMERA_SEEDS", "").split(",") if x]


@pytest.mark.parametrize("seed", _CAMERA_SEEDS)
def test_random_cameras_canonical_scene(canonical_pair, seed):
    """The canonical scene from random viewpoints, inside and outside the teapot and the root box, looking anywhere:
    primary rays then also start in the middle of the tree (boxes behind the origin, rays leaving the root box at once,
    grazing directions) -- what only bounce rays do from the fixed camera."""
    so, sp = canonical_pair
    orc, R = _orc(), _R()
    rng = np.random.default_rng(7000 + seed)
    pos = (rng.uniform(-6, 6, 3) + np.array([0, 0, 5.0])).astype(np.float32) if seed % 3 else rng.uniform(-30, 30, 3).astype(np.float32)
    aim = [float(x) for x in rng.normal(size=3)]
    w, h, spp, depth = 36, 26, int(rng.integers(1, 4)), int(rng.integers(1, 6))
    fov, roll = float(rng.uniform(30, 120)), float(rng.uniform(-1, 1))
    vo = orc.create_viewport(w, h, (1.0, 0.7), pos, orc.unit(aim), fov, roll)
    vp = R.create_viewport((w, h), (1.0, 0.7), pos, R.unit(aim), fov, roll, depth, spp)
    assert_bits_equal(vo, vp.vp12, "viewport")
    ref, cn = so.render(w, h, vo, depth, spp, seed=seed, threads=8)
    for opts in (0, R.OPT_COUNTERS):
        img = np.zeros((h, w, 4), np.float32)
        ctx = R.HipRayCaster(seed=seed, options=opts, tuning={"subtile_min_paths": 1}).walk_rays(vp, sp, img, 1, False)
        assert_bits_equal(ref, img, f"camera {pos} -> {aim}, spp {spp}, depth {depth}, options {opts}")
        assert ctx.total_rays == cn["rays"]
        if opts:
            for k in ("box_tests", "tri_tests", "full_tests", "nodes", "leaves"):
                assert ctx.stats[k] == cn[k], k


def test_quantize_device_matches_oracle(canonical_pair):
    import torch
    R = _R()
    _, sp = canonical_pair
    H, W = 24, 40
    vp = R.canonical_viewport(W, H, 5, 2)
    c = R.HipRayCaster(seed=2)
    frame = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    c.walk_tile_device(vp, sp, (0, H, H, 0), frame.data_ptr(), st)
    rgb = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0")
    c.quantize_device(sp, frame.data_ptr(), H * W, rgb.data_ptr(), st)
    torch.cuda.synchronize()
    assert np.array_equal(rgb.cpu().numpy().reshape(-1, 3), _orc().quantize(frame.cpu().numpy()))


def test_make_triangles_gpu(canonical_pair):
    # the GPU make_triangle kernel (k_make_triangles) against the ORACLE's make_triangle (raytrace.rs:340-383): every
    # record of the canonical scene, bit for bit, and against the committed fixture written from the oracle
    import os
    from conftest import GOLDEN
    R = _R()
    so, _ = canonical_pair
    rec, kinds, surf = so.triangles()
    s = R.Scene(False)
    matte = R.SurfaceKind.Matte(R.make_color(252, 119, 0), 0.2)
    s.extend_make_triangles_gpu(rec[:, 20:29].reshape(-1, 3, 3), matte, 0.05)
    got, _, _ = s.triangles()
    assert got.shape == rec.shape
    assert_bits_equal(got[:, :19], rec[:, :19], "geometric fields vs oracle")  # incenter norm r2 sides side_lens
    assert_bits_equal(got[:, 20:], rec[:, 20:], "corners")
    z = np.load(os.path.join(GOLDEN, "canonical_triangle_records.npz"))
    idx = z["idx"]
    idx = idx[idx != 0]  # record 0 is make_dummy_triangle(), not a make_triangle() result
    assert_bits_equal(got[idx][:, :19], z["rec"][z["idx"] != 0][:, :19], "geometric fields vs golden fixture")
    with pytest.raises(RuntimeError, match="degenerate triangle 1"):
        s.extend_make_triangles_gpu(np.array([[[0, 0, 1], [1, 0, 1], [0, 1, 1]], [[0, 0, 0], [0, 0, 0], [0, 0, 0]]], np.float32), matte, 0.0)


def test_odd_sizes_and_many_samples(circles_pair):
    so, sp = circles_pair
    orc, R = _orc(), _R()
    for (w, h, spp, depth) in ((1, 1, 1, 5), (17, 5, 3, 2), (3, 29, 65, 3), (64, 2, 130, 5)):
        vo = orc.canonical_viewport(w, h)
        vp = R.canonical_viewport(w, h, depth, spp)
        assert_bits_equal(vo, vp.vp12, "viewport")
        ref, cn = so.render(w, h, vo, depth, spp, seed=6, threads=8)
        img = np.zeros((h, w, 4), np.float32)
        ctx = R.HipRayCaster(seed=6).walk_rays(vp, sp, img, 1, False)
        assert_bits_equal(ref, img, f"image {w}x{h}x{spp}")
        assert ctx.total_rays == cn["rays"]


def test_two_stream_subtiles_small_and_ragged(canonical_pair):
    """The library deals a tile's rows out to 1..4 internal streams (automatic: 3 for tiles below 2^26 paths, else 1).  Force
    2 and 3 on small, odd-sized images (contiguous bands, striped tiles with a partial last stripe, fewer rows than streams)
    and compare with the one-stream result."""
    import torch
    from rust_raytrace_amd import dist as rd
    so, sp = canonical_pair
    orc, R = _orc(), _R()
    for (w, h, spp) in ((24, 37, 2), (9, 50, 3), (16, 3, 4)):
        vo = orc.canonical_viewport(w, h)
        vp = R.canonical_viewport(w, h, 5, spp)
        ref, cn = so.render(w, h, vo, 5, spp, seed=4, threads=8)
        for k in (2, 3):
            multi = {"subtile_min_paths": 1, "streams": k}
            img = np.zeros((h, w, 4), np.float32)
            ctx = R.HipRayCaster(seed=4, tuning=multi).walk_rays(vp, sp, img, 1, False)
            assert ctx.stats["streams"] == min(k, h)   # the tile's rows are dealt to the streams one by one
            assert_bits_equal(ref, img, f"{k} streams {w}x{h}")
            assert ctx.total_rays == cn["rays"]
            # a striped tile (rank 1 of 3, 4-row stripes) on the same streams
            tile = rd.rank_tile(1, 3, h, 4)
            if tile[1]:
                buf = torch.zeros((tile[1], w, 4), dtype=torch.float32, device="cuda:0")
                R.HipRayCaster(seed=4, tuning=multi).walk_tile_device(vp, sp, tile, buf.data_ptr(), torch.cuda.current_stream().cuda_stream)
                torch.cuda.synchronize()
                assert_bits_equal(ref[rd.tile_rows(tile, h)], buf.cpu().numpy(), f"striped, {k} streams {w}x{h}")
        one = np.zeros((h, w, 4), np.float32)
        ctx1 = R.HipRayCaster(seed=4, tuning={"subtile_min_paths": 1, "streams": 1}).walk_rays(vp, sp, one, 1, False)
        assert ctx1.stats["streams"] == 1
        assert_bits_equal(ref, one, f"one stream {w}x{h}")


def test_fast_option_is_opt_in_and_close(canonical_pair):
    """RTMI_OPT_FAST is NOT the reference's traversal (it skips boxes behind the ray origin) and is never the default.
    Pinned on this frame (canonical scene, 96 x 96 @ 4 spp, seed 1): 0 differing pixels and the same ray count; at the
    full config-3 size 17 of 4 194 304 pixels differ (include/rtmi.h)."""
    so, sp = canonical_pair
    R = _R()
    vp = R.canonical_viewport(96, 96, 5, 4)
    exact = np.zeros((96, 96, 4), np.float32)
    fast = np.zeros_like(exact)
    r0 = R.HipRayCaster(seed=1).walk_rays(vp, sp, exact, 1, False).total_rays
    r1 = R.HipRayCaster(seed=1, options=R.OPT_FAST).walk_rays(vp, sp, fast, 1, False).total_rays
    differing = int((exact.view(np.uint32) != fast.view(np.uint32)).any(axis=2).sum())
    assert differing == 0 and r1 == r0 == 51644
    ref, _ = so.render(96, 96, _orc().canonical_viewport(96, 96), 5, 4, seed=1, threads=8)
    assert_bits_equal(ref, exact, "default mode stays exact")


def test_full_size_config3_sample_rows_vs_oracle(canonical_pair):
    """BASELINE config 3 at its full size (2048 x 2048 @ 64 spp, depth 5, seed 1): six image rows, three of them through
    the teapot, rendered by the HIP path and by the oracle — bit for bit, including the ray counts."""
    import os
    so, sp = canonical_pair
    orc, R = _orc(), _R()
    W = H = 2048
    vo = orc.canonical_viewport(W, H)
    vp = R.canonical_viewport(W, H, 5, 64)
    c = R.HipRayCaster(seed=1)
    threads = max(1, len(os.sched_getaffinity(0)))
    for row in (0, 700, 1024, 1100, 1333, 2047):
        ref, cn = so.render(W, H, vo, 5, 64, seed=1, row0=row, nrows=1, threads=threads)
        got = np.zeros((1, W, 4), np.float32)
        ctx = c.walk_rows(vp, sp, row, 1, got)
        assert_bits_equal(ref, got, f"row {row}")
        assert ctx.total_rays == cn["rays"], row


# ---------------------------------------------------------------- BASELINE configs 4 and 5 at their stated sizes
# image features of the canonical view (64 x 64 first-hit map in tests/golden): teapot rows 25-46, mirror disk A at
# cols 41-60 / rows 14-29, mirror disk B at cols 9-19 / rows 14-25, sky elsewhere; x 64 for a 4096 x 4096 frame.
_C4_WINDOWS = [  # (rank of 8, first row, first col) of 8 x 32 pixel windows; row // 16 % 8 == rank
    (0, 2048, 2000),   # teapot body
    (2, 1952, 1264),   # teapot spout / silhouette
    (3, 1200, 3072),   # mirror disk A (reflective, scattering 0.0002)
    (7, 1136, 768),    # mirror disk B (reflective, scattering 0.002)
    (5, 80, 0),        # sky
]


def test_full_size_config4_windows_vs_oracle(canonical_pair):
    """BASELINE config 4 at its stated size (4096 x 4096 @ 256 spp, depth 5, seed 1, image tiled over 8 ranks in
    interleaved 16-row stripes): the WHOLE tile of a rank is rendered through its rtmi_tile_t (one eighth of the
    frame, ~750 M rays) and pixel windows on the teapot, both mirror disks and the sky are compared with the oracle
    bit for bit (the oracle cannot render 4.3 G samples; it renders the windows, orc_render_window)."""
    import os
    import torch
    from rust_raytrace_amd import dist as rd
    so, sp = canonical_pair
    orc, R = _orc(), _R()
    W = H = 4096
    spp = 256
    vo = orc.canonical_viewport(W, H)
    vp = R.canonical_viewport(W, H, 5, spp)
    c = R.HipRayCaster(seed=1)
    threads = max(1, len(os.sched_getaffinity(0)))
    st = torch.cuda.current_stream().cuda_stream
    for rank, row0, col0 in _C4_WINDOWS:
        tile = rd.rank_tile(rank, 8, H, 16)
        rows = rd.tile_rows(tile, H)
        assert (row0 // 16) % 8 == rank
        buf = torch.zeros((tile[1], W, 4), dtype=torch.float32, device="cuda:0")
        ctx = c.walk_tile_device(vp, sp, tile, buf.data_ptr(), st)
        torch.cuda.synchronize()
        assert ctx.total_rays >= tile[1] * W * spp
        lr = int(np.nonzero(rows == row0)[0][0])  # local row of the window's first row inside the rank's buffer
        got = buf[lr:lr + 8, col0:col0 + 32].cpu().numpy()
        del buf
        ref, _ = so.render_window(W, H, vo, 5, spp, row0, 8, col0, 32, seed=1, threads=threads)
        assert_bits_equal(ref, got, f"config 4 window rank {rank} rows {row0}.. cols {col0}..")


def test_full_size_config4_tiles_reproduce_frame(canonical_pair):
    """4096 x 4096 (config 4's frame, 2 spp to stay in seconds): the 8 interleaved 16-row-stripe tiles, de-interleaved,
    equal the single-tile frame bit for bit; ray counts add up."""
    import torch
    from rust_raytrace_amd import dist as rd
    _, sp = canonical_pair
    R = _R()
    W = H = 4096
    vp = R.canonical_viewport(W, H, 5, 2)
    c = R.HipRayCaster(seed=1)
    st = torch.cuda.current_stream().cuda_stream
    whole = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    total = c.walk_tile_device(vp, sp, (0, H, H, 0), whole.data_ptr(), st).total_rays
    frame = torch.zeros_like(whole)
    rays = 0
    for r in range(8):
        tile = rd.rank_tile(r, 8, H, 16)
        buf = torch.zeros((tile[1], W, 4), dtype=torch.float32, device="cuda:0")
        rays += c.walk_tile_device(vp, sp, tile, buf.data_ptr(), st).total_rays
        torch.cuda.synchronize()
        frame[torch.as_tensor(rd.tile_rows(tile, H), device="cuda:0")] = buf
    assert rays == total
    assert torch.equal(frame.view(torch.int32), whole.view(torch.int32))


@pytest.fixture(scope="module")
def grid_pair_gpu_built():
    """BASELINE config 5 scene: the oracle's copy with the ORACLE's octree builder (about 10 s), the product's copy with
    the octree built on the GPU (rtmi_builder_filter / k_box_contains), which is what bench.py renders from."""
    from conftest import OracleApi, TEAPOT_TRI
    so = recipe_grid()(OracleApi(_orc()))
    sp = _R().grid_scene(TEAPOT_TRI, gpu_build=0)
    return so, sp


def test_full_size_config5_grid_rows_vs_oracle(grid_pair_gpu_built):
    """BASELINE config 5 at its stated octree (maxdepth 10, minobjs 19; 8 x teapot_tri.obj = 50 561 triangles,
    988 380 boxes, depth-10 LDS stack) and size (2048 x 2048 @ 64 spp).  The product renders from the GPU-BUILT tree
    (what bench.py uses), the oracle from the tree of its own builder (raytrace.rs:790-845): sampled rows, image bits
    and all six work counters."""
    import os
    orc, R = _orc(), _R()
    so, sp = grid_pair_gpu_built
    threads = max(1, len(os.sched_getaffinity(0)))
    assert so.num_tris() == sp.num_tris() == 8 * 6320 + 1
    ro, _, _ = so.triangles()
    rp, _, _ = sp.triangles()
    assert_bits_equal(ro, rp, "triangle records")
    assert int(sp.tree()[1][:, 3].max()) == 10
    W = H = 2048
    vo = orc.canonical_viewport(W, H)
    vp = R.canonical_viewport(W, H, 5, 64)
    c = R.HipRayCaster(seed=1, options=R.OPT_COUNTERS)
    bounced = 0
    for row in (420, 1024, 1290, 1700):
        ref, cn = so.render(W, H, vo, 5, 64, seed=1, row0=row, nrows=1, threads=threads)
        got = np.zeros((1, W, 4), np.float32)
        ctx = c.walk_rows(vp, sp, row, 1, got)
        assert_bits_equal(ref, got, f"config 5 row {row}")
        for k in ("rays", "box_tests", "tri_tests", "full_tests", "nodes", "leaves"):
            assert ctx.stats[k] == cn[k], (row, k)
        bounced += cn["rays"] - W * 64
    assert bounced > 0  # the sampled rows do cross the teapots


def recipe_grid_no_tree():
    """recipe_grid() without the bounding-box build."""
    from conftest import TEAPOT_TRI

    def r(api):
        s = api.scene()
        surfs = [api.matte((252, 119, 0), 0.2), api.reflective(0.01, (200, 200, 220), 0.6), api.solid((30, 160, 60)),
                 api.matte((200, 40, 40), 0.35)]
        k = 0
        for iz in range(2):
            for iy in range(2):
                for ix in range(2):
                    off = [-4.5 + 9.0 * ix, -4.5 + 9.0 * iy, 9.0 + 9.0 * iz]
                    api.add_obj(s, TEAPOT_TRI, off, 1.0, api.transform([0.0, 0.3, 1.0], 270.0 + 20.0 * k), surfs[k % 4], 0.05 if k % 2 == 0 else 0.0)
                    k += 1
        s.populate_triangle_numbers()
        return s
    return r


def test_caster_multi_device_fanout_in_library(canonical_pair):
    """HipRayCaster with a device list: walk_rays() stripes the frame over the caster's devices INSIDE the library
    (rtmi_render_frame_multi; the fan-out DefaultRayCaster does over threads, raytrace.rs:1175-1196).  The test box has
    one GPU, so the list names it twice (two resident copies, two host threads); the frame equals the single-device
    one and the oracle bit for bit, also as RGB8 (each band quantised on its device before it crosses to the root)."""
    so, sp = canonical_pair
    orc, R = _orc(), _R()
    w, h, spp = 64, 50, 2
    vo = orc.canonical_viewport(w, h)
    vp = R.canonical_viewport(w, h, 5, spp)
    ref, cn = so.render(w, h, vo, 5, spp, seed=12, threads=8)
    img = np.zeros((h, w, 4), np.float32)
    ctx = R.HipRayCaster(seed=12, devices=[0, 0]).walk_rays(vp, sp, img, 1, False)
    assert_bits_equal(ref, img, "two handles on one device")
    assert ctx.total_rays == cn["rays"]
    assert len(ctx.per_device) == 2 and all(d["rays"] > 0 for d in ctx.per_device)
    assert all(d["peer_access"] == 1 and d["render_ms"] > 0 for d in ctx.per_device)
    # tuning reaches EVERY handle of a multi-device caster (it used to stop at device 0), and so does the reset
    big = R.canonical_viewport(256, 192, 5, spp)  # enough paths per handle for sub-tiles (subtile_min_paths)
    bimg = np.zeros((192, 256, 4), np.float32)
    c1 = R.HipRayCaster(seed=12, devices=[0, 0], tuning={"streams": 1}).walk_rays(big, sp, bimg, 1, False)
    assert [d["streams"] for d in c1.per_device] == [1, 1]
    c3 = R.HipRayCaster(seed=12, devices=[0, 0]).walk_rays(big, sp, bimg, 1, False)
    assert [d["streams"] for d in c3.per_device] == [3, 3]   # the automatic choice for a tile this small
    q = np.zeros((h, w, 3), np.uint8)
    R.HipRayCaster(seed=12, devices=[0, 0]).walk_frame_multi(vp, sp, q, rgb8=True, stripe_rows=8)
    assert np.array_equal(q.reshape(-1, 3), orc.quantize(ref))
    # back to one device: same scene object, the caster drops the extra copy
    one = np.zeros_like(img)
    R.HipRayCaster(seed=12).walk_rays(vp, sp, one, 1, False)
    assert_bits_equal(ref, one, "single device again")
    with pytest.raises(RuntimeError):
        R.HipRayCaster(seed=12, devices=[0, 99]).walk_rays(vp, sp, img, 1, False)


@pytest.mark.parametrize("which", ["canonical", "grid"])
def test_octree_build_on_gpu_equals_oracle_build(which, canonical_pair, grid_pair_gpu_built):
    """f2: build_bounding_box (raytrace.rs:790-845) with every level's box_contains_polygon tests (raytrace.rs:753-779) on
    the GPU (rtmi_builder_filter / k_box_contains).  The flattened tree -- box geometry, topology, leaf lists -- is compared
    DIRECTLY with the tree of the oracle's builder (orc.Scene.tree_flatten), bit for bit, at the stated octree (10, 19), for
    the canonical scene and the 8-teapot grid (config 5); and with the product's host builder, whose time is printed."""
    import os
    import time
    from conftest import TEAPOT_TRI
    R = _R()
    threads = max(1, len(os.sched_getaffinity(0)))
    mk = R.canonical_scene if which == "canonical" else R.grid_scene
    so = canonical_pair[0] if which == "canonical" else grid_pair_gpu_built[0]
    t0 = time.time()
    host = mk(TEAPOT_TRI, threads=threads)
    t1 = time.time()
    gpu = mk(TEAPOT_TRI, gpu_build=0)
    t2 = time.time()
    go, to, ro = so.tree_flatten()
    gh, th, rh = host.tree()
    gg, tg, rg = gpu.tree()
    assert go.shape == gg.shape and to.shape == tg.shape and ro.shape == rg.shape
    assert_bits_equal(go, gg, "box geometry, GPU build vs oracle build")
    assert np.array_equal(to, tg), "topology, GPU build vs oracle build"
    assert np.array_equal(ro, rg), "leaf lists, GPU build vs oracle build"
    assert_bits_equal(gh, gg, "box geometry, GPU build vs host build")
    assert np.array_equal(th, tg) and np.array_equal(rh, rg), "GPU build vs host build"
    print(f"[{which}] host build ({threads} threads) {t1 - t0:.2f} s, GPU build {t2 - t1:.2f} s (includes OBJ load + make_triangle), "
          f"{len(gg)} boxes, {len(rg)} references")
    # a frame from the GPU-built tree is the oracle's frame
    vo, vp = _viewports(48, 48, 5, 2)
    ref, cn = so.render(48, 48, vo, 5, 2, seed=3, threads=8)
    b = np.zeros((48, 48, 4), np.float32)
    ctx = R.HipRayCaster(seed=3).walk_rays(vp, gpu, b, 1, False)
    assert_bits_equal(ref, b, "frame from the GPU-built tree vs the oracle")
    assert ctx.total_rays == cn["rays"]


def test_fused_and_per_pass_pipelines_same_image(canonical_pair, circles_pair):
    """The default pipeline renders an octree scene with the fused path kernels (k_path_primary: pixel_ray + closest hit +
    color_ray of the primary rays; k_path_bounce: every bounce of every path in one persistent launch, shaded in place);
    tuning pipeline=1 runs the same frame with one launch per bounce pass (k_gen, k_trace_oct + k_shade per pass), pipeline=3
    (the default) k_path_primary and then the bounce passes one launch each.  Same
    device functions, so: image bits, "Rays" and all work counters equal each other AND the oracle -- odd sizes, sample
    counts that are not powers of two (the path -> (pixel, sample) mapping divides by them), depth limits 1..7, batches
    smaller than a wave, refill thresholds 1..64, striped tiles."""
    orc, R = _orc(), _R()
    for pair, (w, h, spp, depth, seed) in ((canonical_pair, (97, 61, 3, 5, 5)), (canonical_pair, (64, 64, 1, 5, 1)),
                                           (canonical_pair, (33, 47, 7, 1, 2)), (canonical_pair, (50, 20, 65, 3, 3)),
                                           (circles_pair, (71, 53, 5, 7, 9)), (canonical_pair, (40, 36, 2, 2, 11))):
        so, sp = pair
        vo, vp = _viewports(w, h, depth, spp)
        ref, cn = so.render(w, h, vo, depth, spp, seed=seed, threads=8)
        imgs = {}
        for pipe in (1, 2, 3):
            img = np.zeros((h, w, 4), np.float32)
            ctx = R.HipRayCaster(seed=seed, options=R.OPT_COUNTERS, tuning={"pipeline": pipe}).walk_rays(vp, sp, img, 1, False)
            assert ctx.stats["pipeline"] == pipe
            assert_bits_equal(ref, img, f"pipeline {pipe} {w}x{h}@{spp} depth {depth}")
            for k in ("rays", "box_tests", "tri_tests", "full_tests", "nodes", "leaves"):
                assert ctx.stats[k] == cn[k], (pipe, k)
            imgs[pipe] = img
            assert ctx.stats["trace_launches"] == (min(depth, 2) if pipe == 2 else depth) * ctx.stats["streams"]
        # uncounted kernels (the ones that are timed)
        for pipe in (1, 2, 3):
            img = np.zeros((h, w, 4), np.float32)
            ctx = R.HipRayCaster(seed=seed, tuning={"pipeline": pipe}).walk_rays(vp, sp, img, 1, False)
            assert_bits_equal(ref, img, f"uncounted pipeline {pipe}")
            assert ctx.total_rays == cn["rays"]
    # tuning never changes a pixel: tiny batches (less than a wave of paths), every refill threshold, stream counts
    so, sp = canonical_pair
    w, h, spp = 45, 38, 3
    vo, vp = _viewports(w, h, 5, spp)
    ref, cn = so.render(w, h, vo, 5, spp, seed=21, threads=8)
    for tun in ({"batch_paths": 40}, {"batch_paths": 1000, "streams": 2}, {"refill_min": 1, "refill_min0": 1}, {"refill_min": 64, "refill_min0": 64},
                {"refill_min": 23, "refill_min0": 17, "streams": 4, "subtile_min_paths": 1}, {"oct_waves_per_cu": 1}, {"xcd_aware": 1},
                {"xcd_aware": 2, "streams": 1}):
        img = np.zeros((h, w, 4), np.float32)
        for pipe in (2, 3):
            ctx = R.HipRayCaster(seed=21, tuning=dict(tun, pipeline=pipe)).walk_rays(vp, sp, img, 1, False)
            assert_bits_equal(ref, img, f"pipeline {pipe}, tuning {tun}")
            assert ctx.total_rays == cn["rays"], (pipe, tun)
    # a striped tile (one rank of three) through the fused kernels equals those rows of the frame
    import torch
    from rust_raytrace_amd import dist as rd
    tile = rd.rank_tile(1, 3, h, 4)
    buf = torch.zeros((tile[1], w, 4), dtype=torch.float32, device="cuda:0")
    R.HipRayCaster(seed=21, tuning={"pipeline": 2}).walk_tile_device(vp, sp, tile, buf.data_ptr())
    torch.cuda.synchronize()
    assert_bits_equal(ref[rd.tile_rows(tile, h)], buf.cpu().numpy(), "striped tile, fused")


def test_slow_path_for_zero_direction_components(canonical_pair):
    """A ray whose unit direction has an exactly-zero component skips that axis' slab in BoundingBox::collides
    (raytrace.rs:872, :882, :892) and enters every box of the perpendicular plane -- ~150 x the work of an ordinary ray.
    The default pipeline sets such rays aside (SlowQ) and k_path_slow traces their paths beside the ordinary passes.
    One-row and one-column images at 1 spp put EVERY primary ray on the camera axis plane (row + 0.5 over a height of 1
    is exactly the axis), so every path takes the slow route: image bits, "Rays" and all six work counters equal the
    oracle's and the renders with the slow path switched off / the per-pass pipeline (which has none)."""
    so, sp = canonical_pair
    orc, R = _orc(), _R()
    for (w, h) in ((96, 1), (1, 80), (1, 1)):
        # image plane 1 x 1: with one row (column) the pixel centre (0 + 0.5) * 1 lies exactly on the camera axis
        vo = orc.create_viewport(w, h, (1.0, 1.0), [2.0, 0.0, 0.0], orc.unit([0.0, 0.0, 1.0]), 90.0, orc.to_radians(0.0))
        vp = R.create_viewport((w, h), (1.0, 1.0), [2.0, 0.0, 0.0], R.unit([0.0, 0.0, 1.0]), 90.0, R.to_radians(0.0), 5, 1)
        assert_bits_equal(vo, vp.vp12, "viewport")
        ref, cn = so.render(w, h, vo, 5, 1, seed=6, threads=8)
        o4, d4 = orc.primary_rays(w, h, vo, 1, seed=6)
        nzero = int(((d4[:, :3] == 0).any(axis=1)).sum())
        assert nzero == w * h  # the premise of this test
        for tun, want_slow in (({"pipeline": 3}, True), ({"pipeline": 2}, True), ({"pipeline": 3, "slow_path_off": 1}, False),
                               ({"pipeline": 1}, False), ({"pipeline": 3, "streams": 2, "subtile_min_paths": 1, "batch_paths": 7}, True)):
            img = np.zeros((h, w, 4), np.float32)
            ctx = R.HipRayCaster(seed=6, options=R.OPT_COUNTERS, tuning=tun).walk_rays(vp, sp, img, 1, False)
            assert_bits_equal(ref, img, f"{w}x{h} {tun}")
            for k in ("rays", "box_tests", "tri_tests", "full_tests", "nodes", "leaves"):
                assert ctx.stats[k] == cn[k], (tun, k, ctx.stats[k], cn[k])
            assert (ctx.stats["slow_paths"] >= w * h) == want_slow, (tun, ctx.stats["slow_paths"])
            img2 = np.zeros((h, w, 4), np.float32)
            ctx2 = R.HipRayCaster(seed=6, tuning=tun).walk_rays(vp, sp, img2, 1, False)   # the uncounted kernels
            assert_bits_equal(ref, img2, f"uncounted {w}x{h} {tun}")
            assert ctx2.total_rays == cn["rays"]
    # a frame where only SOME paths may be slow: jittered samples next to the camera axis (`row + v_off` rounds to the axis row
    # with probability 2^-19 per sample here: usually none -- the frame must be right either way)
    w, h = 40, 32
    vo = orc.create_viewport(w, h, (1.0, 1.0), [2.0, 0.0, 0.0], orc.unit([0.0, 0.0, 1.0]), 90.0, orc.to_radians(0.0))
    vp = R.create_viewport((w, h), (1.0, 1.0), [2.0, 0.0, 0.0], R.unit([0.0, 0.0, 1.0]), 90.0, R.to_radians(0.0), 5, 65)
    vp.samples_per_pixel = 65
    o4, d4 = orc.primary_rays(w, h, vo, 65, seed=2)
    nzero = int(((d4[:, :3] == 0).any(axis=1)).sum())
    ref, cn = so.render(w, h, vo, 5, 65, seed=2, threads=8)
    img = np.zeros((h, w, 4), np.float32)
    ctx = R.HipRayCaster(seed=2).walk_rays(vp, sp, img, 1, False)
    assert_bits_equal(ref, img, "mixed frame")
    assert ctx.total_rays == cn["rays"] and ctx.stats["slow_paths"] >= nzero
    print(f"mixed frame: {nzero} zero-component primary rays of {w * h}, {ctx.stats['slow_paths']} slow paths")


def test_progress_tuples_while_rendering(canonical_pair):
    """f4 / raytrace.rs:1411, :1429-1435: the reference's caster reports progress per finished row.  With a callback
    HipRayCaster.walk_rays renders in row bands and reports (thread, last row, pixels, {"Rays"}) after each: the tuples
    cover every pixel once, their ray counts add up to the frame's, and the image is the one-piece image (== the oracle)."""
    so, sp = canonical_pair
    R = _R()
    w, h, spp = 70, 53, 3
    vo, vp = _viewports(w, h, 5, spp)
    ref, cn = so.render(w, h, vo, 5, spp, seed=4, threads=8)
    for bands in (16, 5, 1, 200):
        seen = []
        img = np.zeros((h, w, 4), np.float32)
        ctx = R.HipRayCaster(seed=4).walk_rays(vp, sp, img, 1, False, progress=lambda t, row, px, st: seen.append((t, row, px, st["Rays"])), bands=bands)
        assert_bits_equal(ref, img, f"{bands} bands")
        assert len(seen) == min(bands, h) and seen[-1][1] == h - 1
        assert [r for _, r, _, _ in seen] == sorted(r for _, r, _, _ in seen)
        assert sum(px for _, _, px, _ in seen) == w * h and sum(n for _, _, _, n in seen) == cn["rays"] == ctx.total_rays
        assert ctx.stats["rays"] == cn["rays"]


def test_pool_kernel_is_bit_exact(canonical_pair):
    """tuning kernel=2 selects k_trace_pool (per-wave ray pool in LDS, free ray-to-lane assignment each step); measured
    slower than the default on MI355X and therefore opt-in, but it stays exact: image bits and all six work
    counters against the oracle, plus edge-case rays."""
    so, sp = canonical_pair
    orc, R = _orc(), _R()
    w, h, spp = 48, 40, 3
    vo = orc.canonical_viewport(w, h)
    vp = R.canonical_viewport(w, h, 5, spp)
    ref, cn = so.render(w, h, vo, 5, spp, seed=5, threads=8)
    img = np.zeros((h, w, 4), np.float32)
    ctx = R.HipRayCaster(seed=5, options=R.OPT_COUNTERS, tuning={"kernel": 2}).walk_rays(vp, sp, img, 1, False)
    assert_bits_equal(ref, img, "pool kernel image")
    for k in ("rays", "box_tests", "tri_tests", "full_tests", "nodes", "leaves"):
        assert ctx.stats[k] == cn[k], k
    pair = build_pair(recipe_axis_box())
    vo = orc.create_viewport(33, 33, (1.0, 1.0), [0.0, 0.0, 0.0], orc.unit([0.0, 0.0, 1.0]), 90.0, 0.0)
    vp = R.create_viewport((33, 33), (1.0, 1.0), [0.0, 0.0, 0.0], R.unit([0.0, 0.0, 1.0]), 90.0, 0.0, 5, 1)
    ref, _ = pair[0].render(33, 33, vo, 5, 1)
    img = np.zeros((33, 33, 4), np.float32)
    R.HipRayCaster(tuning={"kernel": 2}).walk_rays(vp, pair[1], img, 1, False)
    assert_bits_equal(ref, img, "pool kernel, axis-aligned scene")


def test_bvh_fast_mode_equals_linear_list_oracle(canonical_pair):
    """RTMI_OPT_BVH (opt-in "fast mode", never the default): the closest hit over ALL triangles with the lowest index
    winning ties, found through the library's SAH BVH.  Its oracle is the REFERENCE's own linear-list semantics: the same
    triangles under build_trivial_bounding_box (raytrace.rs:847-856, one leaf scanned in index order, :1012-1050), with
    the mode's one stated exception: "hits" with a +-inf / NaN time (rays exactly parallel to a triangle's plane, which
    the reference accepts wherever the triangle is) are ignored -- orc_set_finite_hits_only.  Bit-equal hit ids / times /
    faces for primary and random rays and a bit-equal frame; how many pixels that exception and the octree's own
    deviations cost is counted."""
    from conftest import TEAPOT_TRI
    orc, R = _orc(), _R()
    orc.set_finite_hits_only(True)
    try:
        _bvh_vs_linear(canonical_pair, orc, R, TEAPOT_TRI)
    finally:
        orc.set_finite_hits_only(False)


def _bvh_vs_linear(canonical_pair, orc, R, TEAPOT_TRI):
    so_oct, sp = canonical_pair
    so_lin = orc.canonical_scene(TEAPOT_TRI, accel="trivial")      # same triangles, the reference's linear list
    vo = orc.canonical_viewport(64, 64)
    o4, d4 = orc.primary_rays(64, 64, vo, 1)
    rng = np.random.default_rng(11)
    n = 6000
    ro = np.zeros((n, 4), np.float32)
    rd = np.zeros((n, 4), np.float32)
    ro[:, :3] = rng.uniform(-6, 6, (n, 3)) + np.array([0, 0, 6])
    d = rng.normal(size=(n, 3))
    rd[:, :3] = d / np.linalg.norm(d, axis=1, keepdims=True)
    o4, d4 = np.concatenate([o4, ro]), np.concatenate([d4, rd])
    tri_o, t_o, face_o, _ = so_lin.trace(o4, d4)
    tri_g, t_g, face_g, st = R.HipRayCaster(options=R.OPT_BVH).trace(sp, o4, d4)
    assert np.array_equal(tri_o, tri_g), f"{(tri_o != tri_g).sum()} hit ids differ"
    hit = tri_o != 0
    assert hit.sum() > 1000
    assert_bits_equal(t_o[hit], t_g[hit], "hit time")
    assert np.array_equal(face_o[hit], face_g[hit])
    # a seeded frame: BVH mode == the oracle's linear-list render, bit for bit, same ray count
    w, h, spp = 48, 40, 3
    vo = orc.canonical_viewport(w, h)
    vp = R.canonical_viewport(w, h, 5, spp)
    ref_lin, cn = so_lin.render(w, h, vo, 5, spp, seed=7, threads=8)
    img = np.zeros((h, w, 4), np.float32)
    ctx = R.HipRayCaster(seed=7, options=R.OPT_BVH).walk_rays(vp, sp, img, 1, False)
    assert_bits_equal(ref_lin, img, "BVH mode vs linear-list oracle")
    assert ctx.total_rays == cn["rays"]
    # the stated exception, counted: the reference's linear list WITH its parallel-plane hits
    orc.set_finite_hits_only(False)
    ref_lin_all, _ = so_lin.render(w, h, vo, 5, spp, seed=7, threads=8)
    n_inf = int((ref_lin_all.view(np.uint32) != img.view(np.uint32)).any(axis=2).sum())
    # and the default stays the exact octree traversal
    ref_oct, _ = so_oct.render(w, h, vo, 5, spp, seed=7, threads=8)
    exact = np.zeros_like(img)
    R.HipRayCaster(seed=7).walk_rays(vp, sp, exact, 1, False)
    assert_bits_equal(ref_oct, exact, "default mode stays exact")
    differing = int((exact.view(np.uint32) != img.view(np.uint32)).any(axis=2).sum())
    print(f"BVH mode: {n_inf} of {w * h} pixels differ from the reference's linear list (parallel-plane hits), {differing} from the octree render")
    assert n_inf <= 8 and differing <= 8


def test_bvh_fast_mode_other_scenes():
    """BVH mode on the circles scene (tessellated spheres, Solid / Matte / Reflective) and on random triangle soups,
    against the oracle's linear list of the same triangles."""
    from conftest import OracleApi, ProductApi
    orc, R = _orc(), _R()
    orc.set_finite_hits_only(True)
    try:
        _bvh_other(orc, R, OracleApi, ProductApi)
    finally:
        orc.set_finite_hits_only(False)


def _bvh_other(orc, R, OracleApi, ProductApi):
    so, sp = build_pair(recipe_circles(accel="trivial"))
    vo = orc.canonical_viewport(64, 64)
    vp = R.canonical_viewport(64, 64, 5, 2)
    ref, cn = so.render(64, 64, vo, 5, 2, seed=3, threads=8)
    img = np.zeros((64, 64, 4), np.float32)
    ctx = R.HipRayCaster(seed=3, options=R.OPT_BVH).walk_rays(vp, sp, img, 1, False)
    assert_bits_equal(ref, img, "circles, BVH mode")
    assert ctx.total_rays == cn["rays"]
    rng = np.random.default_rng(99)
    pts = (rng.uniform(-3, 3, (300, 1, 3)) + np.array([0, 0, 8.0]) + rng.normal(scale=0.6, size=(300, 3, 3))).astype(np.float32)

    def recipe(api):
        s = api.scene()
        for i in range(len(pts)):
            try:
                api.add_triangle(s, pts[i], api.matte((200, 100, 50), 0.4) if i % 2 else api.solid((20, 200, 40)), 0.05 if i % 3 else 0.0)
            except RuntimeError:
                pass
        s.populate_triangle_numbers()
        s.build_trivial_bounding_box([0.0, 0.0, 8.0], 8.0)
        return s
    so, sp = recipe(OracleApi(orc)), recipe(ProductApi(R))
    ref, cn = so.render(40, 28, orc.canonical_viewport(40, 28), 4, 3, seed=5, threads=8)
    img = np.zeros((28, 40, 4), np.float32)
    R.HipRayCaster(seed=5, options=R.OPT_BVH).walk_rays(R.canonical_viewport(40, 28, 4, 3), sp, img, 1, False)
    assert_bits_equal(ref, img, "triangle soup, BVH mode")


def test_analytic_spheres_config1_circles():
    """a12 / BASELINE config 1 ("circles scene (few spheres), 256 x 256, 1 spp"): ANALYTIC spheres -- a build-defined
    primitive (the reference at this revision has none: parity with the Rust binary is unpinned; the oracle states the
    definition, tests/test_oracle_cpu.py pins it with known answers).  HIP path vs oracle, bit for bit: the plumbing
    frame of config 1, a seeded multi-sample frame (bounces off and inside spheres, one sphere contains the camera), the
    sphere list changing under a resident scene, explicit rays through rtmi_trace (hit index = ntris + sphere)."""
    from conftest import recipe_circles_analytic, OracleApi, ProductApi
    orc, R = _orc(), _R()
    so, sp = build_pair(recipe_circles_analytic())
    for (w, h, spp, depth, seed) in ((256, 256, 1, 5, 1), (64, 48, 5, 5, 3), (33, 17, 2, 2, 9)):
        vo = orc.canonical_viewport(w, h)
        vp = R.canonical_viewport(w, h, depth, spp)
        ref, cn = so.render(w, h, vo, depth, spp, seed=seed, threads=8)
        img = np.zeros((h, w, 4), np.float32)
        ctx = R.HipRayCaster(seed=seed).walk_rays(vp, sp, img, 1, False)
        assert_bits_equal(ref, img, f"analytic spheres {w}x{h}x{spp}")
        assert ctx.total_rays == cn["rays"]
    assert not np.array_equal(ref[..., :3], np.broadcast_to(orc.make_color(128, 180, 255), ref[..., :3].shape))  # not all sky
    # explicit rays: spheres beat / lose against the triangles of the ground disk
    ntris = sp.num_tris()
    rng = np.random.default_rng(4)
    n = 4000
    o4 = np.zeros((n, 4), np.float32)
    d4 = np.zeros((n, 4), np.float32)
    o4[:, :3] = rng.uniform(-3, 3, (n, 3)) + np.array([0.5, 0.5, 5.0])
    d = rng.normal(size=(n, 3))
    d4[:, :3] = d / np.linalg.norm(d, axis=1, keepdims=True)
    tri_t, t_t, face_t, _ = so.trace(o4, d4)              # the oracle's tree (triangles only) ...
    idx_s, t_s, face_s = so.trace_spheres(o4, d4)         # ... and its sphere list, merged by the stated rule
    take = (idx_s != 0) & ((tri_t == 0) | (t_s < t_t))
    exp_tri = np.where(take, ntris + idx_s - 1, tri_t).astype(np.uint32)
    exp_t = np.where(take, t_s, t_t)
    exp_face = np.where(take, face_s, face_t)
    tri_g, t_g, face_g, _ = R.HipRayCaster().trace(sp, o4, d4)
    assert np.array_equal(exp_tri, tri_g)
    hit = exp_tri != 0
    assert (tri_g >= ntris).sum() > 300 and ((tri_g > 0) & (tri_g < ntris)).sum() > 100
    assert_bits_equal(exp_t[hit], t_g[hit], "hit time")
    assert np.array_equal(exp_face[hit], face_g[hit])
    # the sphere list can change under a resident scene (generation counter), and the BVH mode sees spheres too
    sp.push_analytic_sphere([0.0, 0.0, 4.0], 0.5, R.SurfaceKind.Solid(R.make_color(1, 2, 3)))
    so.add_analytic_sphere([0.0, 0.0, 4.0], 0.5, orc.Surface(orc.SOLID, orc.make_color(1, 2, 3)))
    ref, _ = so.render(40, 40, orc.canonical_viewport(40, 40), 5, 2, seed=2, threads=8)
    img = np.zeros((40, 40, 4), np.float32)
    R.HipRayCaster(seed=2).walk_rays(R.canonical_viewport(40, 40, 5, 2), sp, img, 1, False)
    assert_bits_equal(ref, img, "after adding a sphere")
    with pytest.raises(RuntimeError):
        sp.push_analytic_sphere([0.0, 0.0, 4.0], 0.5, R.SurfaceKind(7, R.make_color(1, 2, 3)))
