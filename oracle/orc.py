"""ctypes binding of the CPU oracle (oracle/librt_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by rust_raytrace_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librt_oracle.so")

SOLID, MATTE, REFLECTIVE = 0, 1, 2
COUNTER_NAMES = ("rays", "box_tests", "tri_tests", "full_tests", "nodes", "leaves")


def build(force=False):
    src = os.path.join(_HERE, "rt_oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "librt_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_scene_new.restype = C.c_void_p
        L.orc_scene_new.argtypes = [C.c_int]
        L.orc_scene_free.argtypes = [C.c_void_p]
        L.orc_last_error.restype = C.c_char_p
        L.orc_last_error.argtypes = [C.c_void_p]
        L.orc_num_tris.restype = C.c_uint64
        L.orc_num_tris.argtypes = [C.c_void_p]
        L.orc_to_radians.restype = C.c_float
        L.orc_to_radians.argtypes = [C.c_float]
        L.orc_u32_to_unit_f32.restype = C.c_float
        L.orc_u32_to_unit_f32.argtypes = [C.c_uint32]
        L.orc_tree_flatten_sizes.restype = C.c_uint64
        _lib = L
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def make_color(r, g, b):
    out = np.zeros(3, np.float32)
    lib().orc_make_color(C.c_uint8(r), C.c_uint8(g), C.c_uint8(b), _p(out))
    return out


def unit(v):
    out = np.zeros(3, np.float32)
    lib().orc_unit(_p(_f(v)), _p(out))
    return out


def to_radians(deg):
    return float(lib().orc_to_radians(C.c_float(deg)))


def create_transform(direction, d_roll):
    out = np.zeros(9, np.float32)
    lib().orc_create_transform(_p(_f(direction)), C.c_float(d_roll), _p(out))
    return out


def create_viewport(w, h, size, pos, direction, fov, c_roll):
    out = np.zeros(12, np.float32)
    lib().orc_create_viewport(C.c_uint32(w), C.c_uint32(h), C.c_float(size[0]), C.c_float(size[1]),
                              _p(_f(pos)), _p(_f(direction)), C.c_float(fov), C.c_float(c_roll), _p(out))
    return out


class Surface:
    def __init__(self, kind, color, alpha=0.0, scattering=0.0):
        self.kind, self.color, self.alpha, self.scattering = kind, _f(color), float(alpha), float(scattering)

    def args(self):
        return (C.c_int(self.kind), _p(self.color), C.c_float(self.alpha), C.c_float(self.scattering))


class Scene:
    def __init__(self, with_dummy=True):
        self.h = C.c_void_p(lib().orc_scene_new(1 if with_dummy else 0))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_scene_free(self.h)
            self.h = None

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(lib().orc_last_error(self.h).decode())

    def add_triangle(self, pts, surface, edge):
        self._check(lib().orc_add_triangle(self.h, _p(_f(pts).reshape(9)), *surface.args(), C.c_float(edge)))

    def add_obj(self, path, offset, scale, basis, surface, edge):
        self._check(lib().orc_add_obj(self.h, path.encode(), _p(_f(offset)), C.c_float(scale), _p(_f(basis)),
                                      *surface.args(), C.c_float(edge)))

    def add_disk(self, orig, norm, r, d, n, surface, side_surface, edge):
        self._check(lib().orc_add_disk(self.h, _p(_f(orig)), _p(_f(norm)), C.c_float(r), C.c_float(d), C.c_uint64(n),
                                       *surface.args(), *side_surface.args(), C.c_float(edge)))

    def add_sphere(self, orig, r, lat_lon, surface, edge):
        self._check(lib().orc_add_sphere(self.h, _p(_f(orig)), C.c_float(r), C.c_uint64(lat_lon[0]),
                                         C.c_uint64(lat_lon[1]), *surface.args(), C.c_float(edge)))

    def add_analytic_sphere(self, center, r, surface):
        """Build-defined extension (the reference has no analytic sphere: parity unpinned, see rt_oracle.cpp struct Sphere)."""
        self._check(lib().orc_add_analytic_sphere(self.h, _p(_f(center)), C.c_float(r), *surface.args()))

    def trace_spheres(self, o4, d4):
        o4, d4 = _f(o4).reshape(-1, 4), _f(d4).reshape(-1, 4)
        n = o4.shape[0]
        idx, t, face = np.zeros(n, np.uint32), np.zeros(n, np.float32), np.zeros(n, np.uint32)
        lib().orc_trace_spheres(self.h, C.c_uint64(n), _p(o4), _p(d4), _p(idx), _p(t), _p(face))
        return idx, t, face

    def populate_triangle_numbers(self):
        lib().orc_populate_triangle_numbers(self.h)

    def num_tris(self):
        return int(lib().orc_num_tris(self.h))

    def build_bounding_box(self, orig, len2, maxdepth, minobjs):
        self._check(lib().orc_build_bounding_box(self.h, _p(_f(orig)), C.c_float(len2), C.c_uint64(maxdepth),
                                                 C.c_uint64(minobjs)))

    def build_trivial_bounding_box(self, orig, len2):
        lib().orc_build_trivial_bounding_box(self.h, _p(_f(orig)), C.c_float(len2))

    def triangles(self):
        n = self.num_tris()
        rec = np.zeros((n, 29), np.float32)
        kinds = np.zeros(n, np.int32)
        surf = np.zeros((n, 5), np.float32)
        lib().orc_get_triangles(self.h, _p(rec), _p(kinds), _p(surf))
        return rec, kinds, surf

    def tree_stats(self):
        out = np.zeros(4, np.uint64)
        lib().orc_tree_stats(self.h, _p(out))
        return dict(inner=int(out[0]), leaves=int(out[1]), refs=int(out[2]), maxdepth=int(out[3]))

    def tree_flatten(self):
        nrefs = C.c_uint64(0)
        nb = int(lib().orc_tree_flatten_sizes(self.h, C.byref(nrefs)))
        geo = np.zeros((nb, 4), np.float32)
        topo = np.zeros((nb, 4), np.uint32)
        refs = np.zeros(max(int(nrefs.value), 1), np.uint32)
        lib().orc_tree_flatten(self.h, _p(geo), _p(topo), _p(refs))
        return geo, topo, refs[: int(nrefs.value)]

    def set_tree(self, geo, topo, refs):
        geo = _f(geo).reshape(-1, 4)
        topo = np.ascontiguousarray(topo, np.uint32).reshape(-1, 4)
        refs = np.ascontiguousarray(refs, np.uint32)
        self._check(lib().orc_set_tree(self.h, _p(geo), _p(topo), _p(refs), C.c_uint64(geo.shape[0]), C.c_uint64(refs.shape[0])))

    def render(self, w, h, vp12, maxdepth, spp, seed=1, row0=0, nrows=None, threads=1):
        nrows = h - row0 if nrows is None else nrows
        out = np.zeros((nrows, w, 4), np.float32)
        cn = np.zeros(6, np.uint64)
        self._check(lib().orc_render(self.h, C.c_uint32(w), C.c_uint32(h), _p(_f(vp12)), C.c_uint64(maxdepth),
                                     C.c_uint64(spp), C.c_uint64(seed), C.c_uint64(row0), C.c_uint64(nrows),
                                     C.c_int(threads), _p(out), _p(cn)))
        return out, dict(zip(COUNTER_NAMES, (int(x) for x in cn)))

    def render_window(self, w, h, vp12, maxdepth, spp, row0, nrows, col0, ncols, seed=1, threads=1):
        """Pixels [row0, row0+nrows) x [col0, col0+ncols) of the w x h frame -> (nrows, ncols, 4), counters."""
        out = np.zeros((nrows, ncols, 4), np.float32)
        cn = np.zeros(6, np.uint64)
        self._check(lib().orc_render_window(self.h, C.c_uint32(w), C.c_uint32(h), _p(_f(vp12)), C.c_uint64(maxdepth),
                                            C.c_uint64(spp), C.c_uint64(seed), C.c_uint64(row0), C.c_uint64(nrows),
                                            C.c_uint64(col0), C.c_uint64(ncols), C.c_int(threads), _p(out), _p(cn)))
        return out, dict(zip(COUNTER_NAMES, (int(x) for x in cn)))

    def trace(self, o4, d4):
        o4, d4 = _f(o4).reshape(-1, 4), _f(d4).reshape(-1, 4)
        n = o4.shape[0]
        tri = np.zeros(n, np.uint32)
        t = np.zeros(n, np.float32)
        face = np.zeros(n, np.uint32)
        cn = np.zeros(6, np.uint64)
        self._check(lib().orc_trace(self.h, C.c_uint64(n), _p(o4), _p(d4), _p(tri), _p(t), _p(face), _p(cn)))
        return tri, t, face, dict(zip(COUNTER_NAMES, (int(x) for x in cn)))

    def box_contains_polygon(self, orig, len2, tri):
        return bool(lib().orc_box_contains_polygon(self.h, _p(_f(orig)), C.c_float(len2), C.c_uint64(tri)))


def primary_rays(w, h, vp12, spp, seed=1, row0=0, nrows=None):
    nrows = h - row0 if nrows is None else nrows
    n = nrows * w * spp
    o4 = np.zeros((n, 4), np.float32)
    d4 = np.zeros((n, 4), np.float32)
    lib().orc_primary_rays(C.c_uint32(w), C.c_uint32(h), _p(_f(vp12)), C.c_uint64(spp), C.c_uint64(seed),
                           C.c_uint64(row0), C.c_uint64(nrows), _p(o4), _p(d4))
    return o4, d4


def philox4x32_10(ctr, key):
    out = np.zeros(4, np.uint32)
    lib().orc_philox4x32_10(_p(np.asarray(ctr, np.uint32)), _p(np.asarray(key, np.uint32)), _p(out))
    return out


def rng_block(seed, pixel, sample, blk):
    out = np.zeros(4, np.uint32)
    lib().orc_rng_block(C.c_uint64(seed), C.c_uint32(pixel), C.c_uint32(sample), C.c_uint32(blk), _p(out))
    return out


def u32_to_unit_f32(u):
    return float(lib().orc_u32_to_unit_f32(C.c_uint32(u)))


def kat_face_collision():
    return int(lib().orc_kat_face_collision())


def set_finite_hits_only(on):
    """NOT the reference: ignore "hits" with a +-inf / NaN time (rays exactly parallel to a triangle's plane).  This is
    the definition of the product's opt-in RTMI_OPT_BVH mode; only the test of that mode switches it on (and off again)."""
    lib().orc_set_finite_hits_only(C.c_int(1 if on else 0))


def quantize(rgba):
    rgba = _f(rgba).reshape(-1, 4)
    out = np.zeros((rgba.shape[0], 3), np.uint8)
    lib().orc_quantize(_p(rgba), C.c_uint64(rgba.shape[0]), _p(out))
    return out


# ---------------------------------------------------------------- scenes
def canonical_scene(obj_path, accel="octree", maxdepth=10, minobjs=19, teapot_surface=None):
    """The scene of raytrace/src/main.rs:116-164."""
    s = Scene(with_dummy=True)
    tsurf = teapot_surface or Surface(MATTE, make_color(252, 119, 0), 0.2)
    s.add_obj(obj_path, [0.0, 0.5, 5.0], 1.0, create_transform(unit([0.0, 0.3, 1.0]), to_radians(270.0)), tsurf, 0.05)
    refl1 = Surface(REFLECTIVE, make_color(230, 230, 230), 0.7, 0.0002)
    refl2 = Surface(REFLECTIVE, make_color(230, 230, 230), 0.7, 0.002)
    side = Surface(MATTE, make_color(40, 40, 40), 0.2)
    s.add_disk([4.0, 4.0, 7.0], unit([-0.3, -0.55, -0.5]), 2.0, 0.1, 50, refl1, side, -1.0)
    s.add_disk([4.0, -3.0, 5.0], unit([-0.5, 2.0, -0.5]), 1.0, 0.04, 50, refl2, side, -1.0)
    s.populate_triangle_numbers()
    if accel == "octree":
        s.build_bounding_box([0.0, 0.0, 20.1], 20.0, maxdepth, minobjs)
    elif accel == "trivial":
        s.build_trivial_bounding_box([0.0, 0.0, 0.0], 20.0)
    return s


def canonical_viewport(w, h):
    """main.rs:166-173: size (1, 1*aspect), pos (2,0,0), dir (0,0,1), fov 90, roll 0."""
    aspect = np.float32(h) / np.float32(w)
    return create_viewport(w, h, (1.0, float(np.float32(1.0) * aspect)), [2.0, 0.0, 0.0], unit([0.0, 0.0, 1.0]), 90.0,
                           to_radians(0.0))
