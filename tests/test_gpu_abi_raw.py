"""-m gpu: the C ABI called directly (ctypes on include/rtmi.h structs): hand-made trees, error codes,
and the generic-tree kernel on trees that are NOT octrees, against the oracle running the same tree."""
import ctypes as C

import numpy as np
import pytest

from conftest import OracleApi, assert_bits_equal, recipe_axis_box, recipe_circles

pytestmark = pytest.mark.gpu

RTMI_OK, RTMI_ERR_INVALID, RTMI_ERR_NO_DEVICE, RTMI_ERR_UNSUPPORTED, RTMI_ERR_OOM, RTMI_ERR_DEVICE = 0, 1, 2, 3, 4, 5


class Tri(C.Structure):
    _fields_ = [("incenter", C.c_float * 3), ("norm", C.c_float * 3), ("bounding_r2", C.c_float), ("sides", (C.c_float * 3) * 3),
                ("side_lens", C.c_float * 3), ("edge_thickness", C.c_float), ("surface_kind", C.c_uint32), ("color", C.c_float * 3),
                ("alpha", C.c_float), ("scattering", C.c_float)]


class Box(C.Structure):
    _fields_ = [("orig", C.c_float * 3), ("len2", C.c_float), ("first", C.c_uint32), ("count", C.c_uint32), ("is_leaf", C.c_uint32),
                ("depth", C.c_uint32)]


class Vp(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("orig", C.c_float * 3), ("cam", C.c_float * 3), ("vu", C.c_float * 3),
                ("vv", C.c_float * 3), ("maxdepth", C.c_uint32), ("samples_per_pixel", C.c_uint32)]


def _lib():
    from rust_raytrace_amd import _ffi
    L = _ffi.lib()
    L.rtmi_scene_create.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p]
    L.rtmi_scene_destroy.argtypes = [C.c_void_p]
    L.rtmi_scene_set_options.argtypes = [C.c_void_p, C.c_uint32]
    L.rtmi_render.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    L.rtmi_trace.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return L, _ffi


def _abi_arrays(oscene):
    """rtmi_triangle_t[] / rtmi_box_t[] / refs from an oracle scene (records are 29 floats, see orc_get_triangles)."""
    rec, kinds, surf = oscene.triangles()
    tris = (Tri * len(rec))()
    for i in range(len(rec)):
        t = tris[i]
        t.incenter[:] = rec[i, 0:3]; t.norm[:] = rec[i, 3:6]; t.bounding_r2 = rec[i, 6]
        for k in range(3):
            t.sides[k][:] = rec[i, 7 + 3 * k:10 + 3 * k]
        t.side_lens[:] = rec[i, 16:19]; t.edge_thickness = rec[i, 19]
        t.surface_kind = int(kinds[i]); t.color[:] = surf[i, 0:3]; t.alpha = surf[i, 3]; t.scattering = surf[i, 4]
    geo, topo, refs = oscene.tree_flatten()
    return tris, geo, topo, refs


def _boxes(geo, topo):
    boxes = (Box * len(geo))()
    for i in range(len(geo)):
        boxes[i].orig[:] = geo[i, 0:3]; boxes[i].len2 = geo[i, 3]
        boxes[i].first, boxes[i].count, boxes[i].is_leaf, boxes[i].depth = (int(x) for x in topo[i])
    return boxes


def _create(L, tris, boxes, refs):
    h = C.c_void_p()
    refs = np.ascontiguousarray(refs, np.uint32)
    rc = L.rtmi_scene_create(tris, len(tris), boxes, len(boxes), refs.ctypes.data_as(C.c_void_p), len(refs), 0, C.byref(h))
    return rc, h


def _render(L, ffi, h, vp12, w, hgt, maxdepth, spp, seed):
    vp = Vp(w, hgt, (C.c_float * 3)(*vp12[0:3]), (C.c_float * 3)(*vp12[3:6]), (C.c_float * 3)(*vp12[6:9]), (C.c_float * 3)(*vp12[9:12]),
            maxdepth, spp)
    out = np.zeros((hgt, w, 4), np.float32)
    st = ffi.Stats()
    rc = L.rtmi_render(h, C.byref(vp), seed, 0, hgt, out.ctypes.data_as(C.c_void_p), C.byref(st))
    assert rc == RTMI_OK, L.rtmi_last_error()
    return out, st


def test_non_octree_tree_runs_the_generic_kernel():
    from oracle import orc
    L, ffi = _lib()
    so = recipe_circles(maxdepth=4, minobjs=6)(OracleApi(orc))
    tris, geo, topo, refs = _abi_arrays(so)
    # make it a NON-octree: nudge every box centre and grow every box a little (still encloses what it did)
    rng = np.random.default_rng(5)
    geo = geo.copy()
    geo[:, 0:3] += rng.uniform(-0.01, 0.01, (len(geo), 3)).astype(np.float32)
    geo[:, 3] *= np.float32(1.07)
    so.set_tree(geo, topo, refs)
    rc, h = _create(L, tris, _boxes(geo, topo), refs)
    assert rc == RTMI_OK, L.rtmi_last_error()
    try:
        L.rtmi_scene_set_options(h, 1)  # RTMI_OPT_COUNTERS
        vp12 = orc.canonical_viewport(40, 40)
        img, st = _render(L, ffi, h, vp12, 40, 40, 5, 3, 17)
        ref, cn = so.render(40, 40, vp12, 5, 3, seed=17, threads=8)
        assert_bits_equal(ref, img, "image")
        for k in ("rays", "box_tests", "tri_tests", "full_tests", "nodes", "leaves"):
            assert getattr(st, k) == cn[k], k
    finally:
        L.rtmi_scene_destroy(h)


def test_overlapping_two_child_tree():
    # a hand-made tree: root with two overlapping children that both list every triangle (hits tie between leaves)
    from oracle import orc
    L, ffi = _lib()
    so = recipe_axis_box()(OracleApi(orc))
    tris, _, _, _ = _abi_arrays(so)
    n = len(tris)
    all_refs = np.arange(1, n, dtype=np.uint32)
    geo = np.array([[0, 0, 4, 4], [-0.5, 0, 4, 3], [0.5, 0.25, 4.25, 3]], np.float32)
    topo = np.array([[1, 2, 0, 0], [0, n - 1, 1, 1], [n - 1, n - 1, 1, 1]], np.uint32)
    refs = np.concatenate([all_refs, all_refs[::-1]])  # second leaf lists them in reverse order
    so.set_tree(geo, topo, refs)
    rc, h = _create(L, tris, _boxes(geo, topo), refs)
    assert rc == RTMI_OK, L.rtmi_last_error()
    try:
        vp12 = orc.create_viewport(31, 31, (1.0, 1.0), [0.0, 0.0, 0.0], orc.unit([0.0, 0.0, 1.0]), 90.0, 0.0)
        img, st = _render(L, ffi, h, vp12, 31, 31, 4, 2, 3)
        ref, cn = so.render(31, 31, vp12, 4, 2, seed=3, threads=4)
        assert_bits_equal(ref, img, "image")
        assert st.rays == cn["rays"]
    finally:
        L.rtmi_scene_destroy(h)


def test_scene_create_rejects_malformed_input():
    from oracle import orc
    L, ffi = _lib()
    so = recipe_axis_box()(OracleApi(orc))
    tris, geo, topo, refs = _abi_arrays(so)

    def expect(code, geo=geo, topo=topo, refs=refs, tris=tris, frag=None):
        rc, h = _create(L, tris, _boxes(geo, topo), refs)
        msg = L.rtmi_last_error().decode()
        if rc == RTMI_OK:
            L.rtmi_scene_destroy(h)
        assert rc == code, (rc, msg)
        if frag:
            assert frag in msg, msg

    bad = topo.copy(); bad[0, 0] = 0                     # root's child range starts at the root itself
    expect(RTMI_ERR_INVALID, topo=bad, frag="follow")
    bad = topo.copy(); bad[0, 1] = 9                     # nine children: the reference's boxmap has 8 slots
    expect(RTMI_ERR_INVALID, topo=bad, frag="1..8")
    leaf = int(np.nonzero(topo[:, 2] == 1)[0][0])
    bad = topo.copy(); bad[leaf, 1] = len(refs) + 5      # leaf list runs past the reference array
    expect(RTMI_ERR_INVALID, topo=bad, frag="out of bounds")
    badr = refs.copy(); badr[0] = len(tris) + 3          # triangle index out of range
    expect(RTMI_ERR_INVALID, refs=badr, frag="out of range")
    t2 = (Tri * len(tris))(); C.memmove(t2, tris, C.sizeof(tris)); t2[1].surface_kind = 7
    expect(RTMI_ERR_INVALID, tris=t2, frag="surface kind")
    h = C.c_void_p()
    assert L.rtmi_scene_create(None, 0, None, 0, None, 0, 0, C.byref(h)) == RTMI_ERR_INVALID
    assert L.rtmi_scene_create(tris, len(tris), _boxes(geo, topo), len(geo), refs.ctypes.data_as(C.c_void_p), len(refs), 99,
                               C.byref(h)) == RTMI_ERR_INVALID  # no such device
    # a leaf that lists the sentinel triangle 0 is legal for the generic kernel (hit index 0 then reads as a miss)
    z = refs.copy(); z[0] = 0
    rc, h = _create(L, tris, _boxes(geo, topo), z)
    assert rc == RTMI_OK
    L.rtmi_scene_destroy(h)


def test_trace_empty_and_null_arguments():
    from oracle import orc
    L, ffi = _lib()
    so = recipe_axis_box()(OracleApi(orc))
    tris, geo, topo, refs = _abi_arrays(so)
    rc, h = _create(L, tris, _boxes(geo, topo), refs)
    assert rc == RTMI_OK
    try:
        assert L.rtmi_trace(h, 0, None, None, None, None, None, None) == RTMI_OK          # empty input
        assert L.rtmi_trace(h, 4, None, None, None, None, None, None) == RTMI_ERR_INVALID  # NULL buffers
        assert L.rtmi_trace(None, 0, None, None, None, None, None, None) == RTMI_ERR_INVALID
        assert b"NULL" in L.rtmi_last_error() or b"scene" in L.rtmi_last_error()
    finally:
        L.rtmi_scene_destroy(h)


@pytest.mark.parametrize("options", [0, 2])  # 2 = RTMI_OPT_GENERIC
def test_deep_chain_tree(options):
    """A hand-made exact octree 14 levels deep (the reference builder cannot make deep trees cheaply: every triangle is
    listed in all boxes its plane crosses).  Each inner box has the octant containing P as its main child plus, on even
    levels, the mirrored octant as a sibling leaf; exercises deep LDS stacks in the octree kernel and the 64-thread
    blocks of the generic kernel."""
    from oracle import orc
    L, ffi = _lib()
    so = recipe_axis_box()(OracleApi(orc))
    tris, _, _, _ = _abi_arrays(so)
    all_refs = list(range(1, len(tris)))
    f32 = np.float32
    P = np.array([0.3, 0.2, 4.4], f32)
    depth = 14

    def make(c, h, d):
        if d == depth:
            return dict(c=c, h=h, d=d, refs=all_refs)
        hm = f32(h / f32(2.0))
        main = np.array([f32(c[a] + (hm if P[a] >= c[a] else f32(-1.0) * hm)) for a in range(3)], f32)
        kids = [make(main, hm, d + 1)]
        if d % 2 == 0:
            sib = np.array([f32(c[a] + (f32(-1.0) * hm if P[a] >= c[a] else hm)) for a in range(3)], f32)
            kids.append(dict(c=sib, h=hm, d=d + 1, refs=all_refs[::2] if d % 4 == 0 else all_refs[1::2]))
        kids.sort(key=lambda k: sum(1 << a for a in range(3) if k["c"][a] > c[a]))  # octant order
        return dict(c=c, h=h, d=d, kids=kids)

    root = make(np.array([0.0, 0.0, 4.0], f32), f32(4.0), 0)
    geo, topo, refs, queue = [], [], [], [root]
    for node in queue:  # breadth-first; the list grows while we walk it
        geo.append([*node["c"], node["h"]])
        if "refs" in node:
            topo.append([len(refs), len(node["refs"]), 1, node["d"]])
            refs += node["refs"]
        else:
            topo.append([len(queue), len(node["kids"]), 0, node["d"]])
            queue += node["kids"]
    geo = np.array(geo, f32); topo = np.array(topo, np.uint32); refs = np.array(refs, np.uint32)
    assert topo[:, 3].max() == depth
    so.set_tree(geo, topo, refs)
    rc, h = _create(L, tris, _boxes(geo, topo), refs)
    assert rc == RTMI_OK, L.rtmi_last_error()
    try:
        L.rtmi_scene_set_options(h, 1 | options)
        vp12 = orc.create_viewport(29, 29, (1.0, 1.0), [0.0, 0.0, 0.0], orc.unit([0.0, 0.0, 1.0]), 90.0, 0.0)
        img, st = _render(L, ffi, h, vp12, 29, 29, 5, 3, 12)
        ref, cn = so.render(29, 29, vp12, 5, 3, seed=12, threads=4)
        assert_bits_equal(ref, img, "image")
        for k in ("rays", "box_tests", "tri_tests", "full_tests", "nodes", "leaves"):
            assert getattr(st, k) == cn[k], k
        assert cn["nodes"] > cn["rays"]  # the chain is really descended
    finally:
        L.rtmi_scene_destroy(h)


def test_render_frame_multi_raw_abi():
    """rtmi_render_frame_multi through the raw C ABI: ONE handle (n = 1), then THREE handles of the same scene on the one
    GPU of the test box (three stripes sets, three host threads, three band copies to the root, de-interleave kernel),
    f32 and RGB8 output; each equals rtmi_render of the whole image bit for bit (and the oracle), ray counts add up.
    Height 45 with 4-row stripes: ragged last stripe, unequal band sizes."""
    from oracle import orc
    L, ffi = _lib()
    L.rtmi_render_frame_multi.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p,
                                          C.c_void_p, C.c_void_p]
    so = recipe_circles()(OracleApi(orc))
    tris, geo, topo, refs = _abi_arrays(so)
    boxes = _boxes(geo, topo)
    hs = []
    for _ in range(3):
        rc, h = _create(L, tris, boxes, refs)
        assert rc == RTMI_OK
        hs.append(h)
    w, hgt, spp, seed = 37, 45, 3, 8
    vp12 = orc.canonical_viewport(w, hgt)
    ref, cn = so.render(w, hgt, vp12, 5, spp, seed=seed, threads=8)
    whole, st1 = _render(L, ffi, hs[0], vp12, w, hgt, 5, spp, seed)
    assert_bits_equal(ref, whole, "single-device render vs oracle")
    vp = Vp(w, hgt, (C.c_float * 3)(*vp12[0:3]), (C.c_float * 3)(*vp12[3:6]), (C.c_float * 3)(*vp12[6:9]), (C.c_float * 3)(*vp12[9:12]), 5, spp)
    # (3, 32): stripes of 32 rows on a 45-row image -> scene 0 gets 32 rows, scene 1 gets 13, scene 2 gets NONE
    for n, S in ((1, 0), (3, 4), (2, 16), (3, 1), (3, 32)):
        arr = (C.c_void_p * n)(*[h.value for h in hs[:n]])
        out = np.zeros((hgt, w, 4), np.float32)
        sts = (ffi.Stats * n)()
        rc = L.rtmi_render_frame_multi(arr, n, C.byref(vp), seed, S, 0, out.ctypes.data_as(C.c_void_p), None, sts)
        assert rc == RTMI_OK, L.rtmi_last_error()
        assert L.rtmi_last_error() == b"", "no warning: every handle sits on the root device"
        assert_bits_equal(whole, out, f"multi n={n} S={S}")
        assert sum(s.rays for s in sts) == st1.rays == cn["rays"]
        # per-device diagnostics: which scene rendered / copied for how long, and whether it reaches the root directly
        assert all(s.peer_access == 1 for s in sts)
        assert sts[0].deinterleave_ms > 0 and all(s.deinterleave_ms == 0 for s in sts[1:])
        for i, s_ in enumerate(sts):
            has_rows = (n, S, i) != (3, 32, 2)
            assert (s_.rays > 0) == has_rows and (s_.render_ms > 0) == has_rows and (s_.band_copy_ms > 0) == has_rows
        q = np.zeros((hgt, w, 3), np.uint8)
        rc = L.rtmi_render_frame_multi(arr, n, C.byref(vp), seed, S, 1, q.ctypes.data_as(C.c_void_p), None, None)
        assert rc == RTMI_OK, L.rtmi_last_error()
        assert np.array_equal(q.reshape(-1, 3), orc.quantize(ref)), f"rgb8 n={n} S={S}"
    # RTMI_FRAME_RCCL (flag 2): the bands cross with ONE ncclGather on a communicator of the scenes' devices.  This box has
    # one GPU: one handle runs the whole path (librccl loaded on first use, ncclCommInitAll, grouped ncclGather into the
    # staging buffer, de-interleave) twice (the communicator is kept), f32 and RGB8; two handles on ONE device are refused
    # (RCCL wants a device per rank) without disturbing the handles.  Untested here: more than one physical device.
    arr = (C.c_void_p * 1)(hs[0].value)
    for _ in range(2):
        out = np.zeros((hgt, w, 4), np.float32)
        sts = (ffi.Stats * 1)()
        rc = L.rtmi_render_frame_multi(arr, 1, C.byref(vp), seed, 4, 2, out.ctypes.data_as(C.c_void_p), None, sts)
        assert rc == RTMI_OK, L.rtmi_last_error()
        assert_bits_equal(whole, out, "one handle, ncclGather")
        assert sts[0].rays == cn["rays"] and sts[0].band_copy_ms > 0
    q = np.zeros((hgt, w, 3), np.uint8)
    assert L.rtmi_render_frame_multi(arr, 1, C.byref(vp), seed, 4, 2 | 1, q.ctypes.data_as(C.c_void_p), None, None) == RTMI_OK, L.rtmi_last_error()
    assert np.array_equal(q.reshape(-1, 3), orc.quantize(ref)), "rgb8 through ncclGather"
    arr = (C.c_void_p * 2)(hs[0].value, hs[1].value)
    out = np.zeros((hgt, w, 4), np.float32)
    assert L.rtmi_render_frame_multi(arr, 2, C.byref(vp), seed, 4, 2, out.ctypes.data_as(C.c_void_p), None, None) == RTMI_ERR_UNSUPPORTED
    assert b"device of its own" in L.rtmi_last_error()
    assert L.rtmi_render_frame_multi(arr, 2, C.byref(vp), seed, 4, 0, out.ctypes.data_as(C.c_void_p), None, None) == RTMI_OK
    assert_bits_equal(whole, out, "peer copies again after the refused RCCL call")
    # errors: the same handle twice, no output, unknown flag
    arr = (C.c_void_p * 2)(hs[0].value, hs[0].value)
    out = np.zeros((hgt, w, 4), np.float32)
    assert L.rtmi_render_frame_multi(arr, 2, C.byref(vp), seed, 4, 0, out.ctypes.data_as(C.c_void_p), None, None) == RTMI_ERR_INVALID
    arr = (C.c_void_p * 1)(hs[0].value)
    assert L.rtmi_render_frame_multi(arr, 1, C.byref(vp), seed, 4, 0, None, None, None) == RTMI_ERR_INVALID
    assert L.rtmi_render_frame_multi(arr, 1, C.byref(vp), seed, 4, 8, out.ctypes.data_as(C.c_void_p), None, None) == RTMI_ERR_INVALID
    for h in hs:
        L.rtmi_scene_destroy(h)


def test_peer_access_is_asked_once_and_leaves_nothing_pending():
    """rtmi_render_frame_multi used to call hipDeviceEnablePeerAccess on every frame; from the second
    frame on it answers hipErrorPeerAccessAlreadyEnabled and leaves that error pending for the next hipGetLastError()
    poll of the launch checks.  The step now runs once per (device, root) pair, is cached on the handle and clears what it
    leaves.  A one-GPU box cannot take the cross-device branch for real, so the step is driven through its test hook:
    against the own device, against it again, after forgetting the cached answer, and against a device that does not
    exist (a refusal: recorded, not an error, nothing pending) -- and a frame renders bit-equal right after each."""
    from oracle import orc
    L, ffi = _lib()
    L.rtmi_debug_peer_access.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    so = recipe_circles()(OracleApi(orc))
    tris, geo, topo, refs = _abi_arrays(so)
    rc, h = _create(L, tris, _boxes(geo, topo), refs)
    assert rc == RTMI_OK
    try:
        vp12 = orc.canonical_viewport(24, 20)
        ref, _ = so.render(24, 20, vp12, 5, 2, seed=4, threads=4)
        for root, forget, want_ok in ((0, 0, 1), (0, 0, 1), (0, 1, 1), (99, 1, 0), (99, 0, 0), (0, 1, 1)):
            ok, pending = C.c_int(-1), C.c_int(-1)
            assert L.rtmi_debug_peer_access(h, root, forget, C.byref(ok), C.byref(pending)) == RTMI_OK, L.rtmi_last_error()
            assert ok.value == want_ok and pending.value == 0, (root, forget, ok.value, pending.value)
            img, _ = _render(L, ffi, h, vp12, 24, 20, 5, 2, 4)
            assert_bits_equal(ref, img, f"render after the peer step (root {root})")
    finally:
        L.rtmi_scene_destroy(h)


def test_oom_and_device_error_codes():
    """RTMI_ERR_OOM provoked for real: an allocation no MI355X can satisfy (2^36 pixels x 16 B = 1 TiB; the request fails
    before anything is read or written).  The error must not stay pending in the HIP runtime either: PyTorch polls
    hipGetLastError() after its own calls and would report this library's failure as its own (it did, before
    hip_code() cleared it).  RTMI_ERR_DEVICE is what every other HIP runtime failure maps to; a real one (a faulting
    kernel, a destroyed stream) cannot be provoked safely on a shared GPU host -- a stale hipStream_t segfaults inside
    the runtime, a faulting kernel can reset the GPU for everyone -- so the mapping itself is checked through the
    library's test hook for the statuses that matter."""
    from oracle import orc
    import torch
    L, ffi = _lib()
    L.rtmi_quantize.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.rtmi_debug_status_of.argtypes = [C.c_int]
    so = recipe_circles()(OracleApi(orc))
    tris, geo, topo, refs = _abi_arrays(so)
    rc, h = _create(L, tris, _boxes(geo, topo), refs)
    assert rc == RTMI_OK
    try:
        small = np.zeros(16, np.float32)
        assert L.rtmi_quantize(h, small.ctypes.data_as(C.c_void_p), 1 << 36, small.ctypes.data_as(C.c_void_p)) == RTMI_ERR_OOM
        assert b"memory" in L.rtmi_last_error().lower()
        dev = torch.zeros((8, 16, 4), dtype=torch.float32, device="cuda:0")  # raises if the OOM were still pending
        torch.cuda.synchronize()
        assert float(dev.sum()) == 0.0
        # hipError_t -> ABI status (hip_runtime_api.h: 2 OutOfMemory, 100 NoDevice, 101 InvalidDevice, 700 IllegalAddress,
        # 719 LaunchFailure, 1 InvalidValue, 400 InvalidHandle, 900 StreamCaptureUnsupported)
        for hip_err, want in ((2, RTMI_ERR_OOM), (100, RTMI_ERR_NO_DEVICE), (101, RTMI_ERR_NO_DEVICE), (700, RTMI_ERR_DEVICE),
                              (719, RTMI_ERR_DEVICE), (1, RTMI_ERR_DEVICE), (400, RTMI_ERR_DEVICE), (900, RTMI_ERR_DEVICE)):
            assert L.rtmi_debug_status_of(hip_err) == want, hip_err
            assert L.rtmi_last_error() != b""
        # the handle still renders, and equals the oracle
        w, hgt = 16, 8
        vp12 = orc.canonical_viewport(w, hgt)
        ref, _ = so.render(w, hgt, vp12, 5, 2, seed=1, threads=4)
        img, _ = _render(L, ffi, h, vp12, w, hgt, 5, 2, 1)
        assert_bits_equal(ref, img, "render after the provoked errors")
    finally:
        L.rtmi_scene_destroy(h)


def test_corners_are_optional_and_only_tighten_the_bvh():
    """rtmi_scene_set_corners: the opt-in BVH mode gives the same hits with the disc boxes alone (no corners) and with the
    corners' boxes -- rtmi_trace on primary + random rays, ids / times / faces bit-equal -- and equals the exact octree
    traversal wherever neither a tie nor a lost triangle is involved (here: everywhere on this scene's primary rays).
    Wrong sizes and NULL are RTMI_ERR_INVALID and leave the scene usable."""
    from oracle import orc
    from conftest import recipe_circles
    L, ffi = _lib()
    L.rtmi_scene_set_corners.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    so = recipe_circles()(OracleApi(orc))
    tris, geo, topo, refs = _abi_arrays(so)
    rec, _, _ = so.triangles()
    corners = np.ascontiguousarray(rec[:, 20:29], np.float32)
    rc, h = _create(L, tris, _boxes(geo, topo), refs)
    assert rc == RTMI_OK
    try:
        vp12 = orc.canonical_viewport(48, 48)
        o4, d4 = orc.primary_rays(48, 48, vp12, 1)
        rng = np.random.default_rng(3)
        ro = np.zeros((4000, 4), np.float32); rd = np.zeros((4000, 4), np.float32)
        ro[:, :3] = rng.uniform(-4, 4, (4000, 3)) + np.array([0, 0, 8]); d = rng.normal(size=(4000, 3)); rd[:, :3] = d / np.linalg.norm(d, axis=1, keepdims=True)
        o4 = np.concatenate([o4, ro]); d4 = np.concatenate([d4, rd])
        n = len(o4)

        def trace(options):
            L.rtmi_scene_set_options(h, options)
            tri, t, face = np.zeros(n, np.uint32), np.zeros(n, np.float32), np.zeros(n, np.uint32)
            assert L.rtmi_trace(h, n, o4.ctypes.data_as(C.c_void_p), d4.ctypes.data_as(C.c_void_p), tri.ctypes.data_as(C.c_void_p),
                                t.ctypes.data_as(C.c_void_p), face.ctypes.data_as(C.c_void_p), None) == RTMI_OK, L.rtmi_last_error()
            return tri, t, face
        a = trace(8)                                   # RTMI_OPT_BVH, boxes from the bounding-radius discs
        assert L.rtmi_scene_set_corners(h, corners.ctypes.data_as(C.c_void_p), len(corners) + 1) == RTMI_ERR_INVALID
        assert L.rtmi_scene_set_corners(h, None, len(corners)) == RTMI_ERR_INVALID
        assert L.rtmi_scene_set_corners(h, corners.ctypes.data_as(C.c_void_p), len(corners)) == RTMI_OK, L.rtmi_last_error()
        b = trace(8)                                   # the same mode, boxes tightened by the corners
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[2][a[0] != 0], b[2][b[0] != 0])
        assert_bits_equal(a[1][a[0] != 0], b[1][b[0] != 0], "hit times, BVH with and without corners")
        e = trace(0)                                   # exact octree traversal
        same = e[0] == b[0]
        assert same.mean() > 0.995, same.mean()       # ties / lost triangles aside (none expected on these rays)
        assert_bits_equal(e[1][same & (e[0] != 0)], b[1][same & (e[0] != 0)], "hit times where the modes agree on the triangle")
    finally:
        L.rtmi_scene_destroy(h)
