"""Python view of the host-side mirror of rust_raytrace's `raytrace` module.

Names follow raytrace_lib/src/raytrace.rs: make_color, create_transform,
create_viewport, Scene (tris + boxes), make_disk / make_sphere / parse_obj
(as Scene.extend_* helpers, since triangles live in the C++ Scene),
build_bounding_box, build_trivial_bounding_box, and the RayCaster plug-in
`HipRayCaster` whose walk_rays() runs the MI355X kernels through the C ABI.
"""
import ctypes as C
import time

import numpy as np

from . import _ffi

SOLID, MATTE, REFLECTIVE = 0, 1, 2
OPT_COUNTERS, OPT_GENERIC, OPT_FAST, OPT_BVH = 1, 2, 4, 8


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _chk(rc):
    if rc != 0:
        raise RuntimeError(_ffi.lib().rth_last_error().decode())


def make_color(r, g, b):
    """raytrace.rs:176-180"""
    out = np.zeros(3, np.float32)
    _ffi.lib().rth_make_color(r, g, b, _p(out))
    return out


def unit(v):
    out = np.zeros(3, np.float32)
    _ffi.lib().rth_unit(_p(_f(v)), _p(out))
    return out


def to_radians(deg):
    return float(_ffi.lib().rth_to_radians(deg))


def create_transform(direction, d_roll):
    """raytrace.rs:1320-1341 -> 9 floats (three basis rows)."""
    out = np.zeros(9, np.float32)
    _ffi.lib().rth_create_transform(_p(_f(direction)), d_roll, _p(out))
    return out


class SurfaceKind:
    """raytrace.rs:303-308"""

    def __init__(self, kind, color, alpha=0.0, scattering=0.0):
        self.kind, self.color, self.alpha, self.scattering = kind, _f(color), float(alpha), float(scattering)

    @staticmethod
    def Solid(color):
        return SurfaceKind(SOLID, color)

    @staticmethod
    def Matte(color, alpha):
        return SurfaceKind(MATTE, color, alpha)

    @staticmethod
    def Reflective(scattering, color, alpha):
        return SurfaceKind(REFLECTIVE, color, alpha, scattering)

    def args(self):
        return (self.kind, _p(self.color), self.alpha, self.scattering)


class Viewport:
    """raytrace.rs:1305-1318"""

    def __init__(self, width, height, vp12, maxdepth, samples_per_pixel):
        self.width, self.height = int(width), int(height)
        self.vp12 = _f(vp12)
        self.maxdepth, self.samples_per_pixel = int(maxdepth), int(samples_per_pixel)


def create_viewport(px, size, pos, direction, fov, c_roll, maxdepth, samples):
    """raytrace.rs:1343-1370"""
    out = np.zeros(12, np.float32)
    _ffi.lib().rth_create_viewport(px[0], px[1], size[0], size[1], _p(_f(pos)), _p(_f(direction)), fov, c_roll, _p(out))
    return Viewport(px[0], px[1], out, maxdepth, samples)


class Scene:
    """raytrace.rs:1297-1303: tris + boxes (held by the C++ mirror)."""

    def __init__(self, with_dummy=True):
        self.h = C.c_void_p(_ffi.lib().rth_scene_new(1 if with_dummy else 0))

    def __del__(self):
        if getattr(self, "h", None) and _ffi is not None:  # module globals are None while the interpreter shuts down
            _ffi.lib().rth_scene_free(self.h)
            self.h = None

    # --- Scene.tris
    def num_tris(self):
        return int(_ffi.lib().rth_num_tris(self.h))

    def push_triangle(self, points, surface, edge_thickness):
        """obj_data.push(make_triangle(points, surface, edge_thickness))"""
        _chk(_ffi.lib().rth_add_triangle(self.h, _p(_f(points).reshape(9)), *surface.args(), edge_thickness))

    def extend_make_triangles_gpu(self, points, surface, edge_thickness, device=0):
        """obj_data.extend(points.map(make_triangle)) computed by the GPU kernel (rtmi_make_triangles); points: (n, 3, 3)."""
        pts = _f(points).reshape(-1, 9)
        _chk(_ffi.lib().rth_add_triangles_gpu(self.h, _p(pts), pts.shape[0], *surface.args(), edge_thickness, device))

    def extend_parse_obj(self, path, offset, scale, transform, surface, edge_thickness, robust=False):
        """obj_data.extend(obj_parser::parse_obj(...)) — obj_parser.rs:47-73.  robust=True is an opt-in loader extension
        (fan-triangulated polygons, negative indices, degenerate triangles skipped); the default is the reference's loader."""
        _chk(_ffi.lib().rth_add_obj_mode(self.h, path.encode(), _p(_f(offset)), scale, _p(_f(transform)), *surface.args(),
                                         edge_thickness, 1 if robust else 0))

    def extend_make_disk(self, orig, norm, r, d, num_tris, surface, side_surface, edge_thickness):
        """obj_data.extend(make_disk(...)) — raytrace.rs:531-592"""
        _chk(_ffi.lib().rth_add_disk(self.h, _p(_f(orig)), _p(_f(norm)), r, d, num_tris, *surface.args(),
                                     *side_surface.args(), edge_thickness))

    def extend_make_sphere(self, orig, r, lat_lon, surface, edge_thickness):
        """obj_data.extend(make_sphere(...)) — raytrace.rs:464-529"""
        _chk(_ffi.lib().rth_add_sphere(self.h, _p(_f(orig)), r, lat_lon[0], lat_lon[1], *surface.args(), edge_thickness))

    def push_analytic_sphere(self, center, radius, surface):
        """Analytic sphere primitive: a build-defined extension (the reference only tessellates, raytrace.rs:464-529);
        a flat list tested against every ray after the box tree.  Semantics: include/rtmi.h (rtmi_sphere_t)."""
        _chk(_ffi.lib().rth_add_analytic_sphere(self.h, _p(_f(center)), radius, *surface.args()))

    def populate_triangle_numbers(self):
        _ffi.lib().rth_populate_triangle_numbers(self.h)

    # --- Scene.boxes
    def build_bounding_box(self, orig, len2, maxdepth, minobjs, threads=0, gpu_device=None):
        """raytrace.rs:790-845.  gpu_device: evaluate the box/triangle overlap tests of every level on that GPU
        (rtmi_builder_*, k_box_contains) instead of on `threads` host threads; the tree is bit-equal either way."""
        if gpu_device is not None:
            _chk(_ffi.lib().rth_build_bounding_box_gpu(self.h, _p(_f(orig)), len2, maxdepth, minobjs, int(gpu_device)))
        else:
            _chk(_ffi.lib().rth_build_bounding_box(self.h, _p(_f(orig)), len2, maxdepth, minobjs, threads))

    def build_trivial_bounding_box(self, orig, len2):
        """raytrace.rs:847-856"""
        _chk(_ffi.lib().rth_build_trivial_bounding_box(self.h, _p(_f(orig)), len2))

    def box_contains_polygon(self, orig, len2, tri):
        return bool(_ffi.lib().rth_box_contains_polygon(self.h, _p(_f(orig)), len2, tri))

    def face_contains_triangle(self, p, norm, len2, tri):
        return bool(_ffi.lib().rth_face_contains_triangle(self.h, _p(_f(p)), _p(_f(norm)), len2, tri))

    # --- inspection
    def triangles(self):
        n = self.num_tris()
        rec = np.zeros((n, 29), np.float32)
        kinds = np.zeros(n, np.int32)
        surf = np.zeros((n, 5), np.float32)
        _ffi.lib().rth_get_triangles(self.h, _p(rec), _p(kinds), _p(surf))
        return rec, kinds, surf

    def tree(self):
        nb, nr = C.c_uint64(0), C.c_uint64(0)
        _ffi.lib().rth_tree_sizes(self.h, C.byref(nb), C.byref(nr))
        geo = np.zeros((nb.value, 4), np.float32)
        topo = np.zeros((nb.value, 4), np.uint32)
        refs = np.zeros(max(nr.value, 1), np.uint32)
        _ffi.lib().rth_tree_get(self.h, _p(geo), _p(topo), _p(refs))
        return geo, topo, refs[: nr.value]

    def tree_stats(self):
        _, topo, refs = self.tree()
        leaf = topo[:, 2] == 1
        return dict(inner=int((~leaf).sum()), leaves=int(leaf.sum()), refs=int(len(refs)),
                    maxdepth=int(topo[:, 3].max()) if len(topo) else 0)


class ProgressCtx:
    """What progress.rs:157-184 reports."""

    def __init__(self, total_rays, seconds, stats):
        self.total_rays, self.seconds, self.stats = total_rays, seconds, stats

    def stats_line(self):
        m = self.total_rays / 1e6
        return f"Processed {m:.3f} million rays in {self.seconds:.3f} seconds. {m / max(self.seconds, 1e-12):.3f} million rays/s"


class HipRayCaster:
    """impl RayCaster (raytrace.rs:1128-1165) on MI355X.

    walk_rays(v, s, data, threads, show_progress) fills `data` ((H, W, 4) f32,
    the reference's `&mut [Color]`) and returns a ProgressCtx.  `threads` is
    accepted and ignored, as the reference's CudaRayCaster does.
    """

    def __init__(self, seed=1, device=0, options=0, tuning=None, devices=None):
        """tuning: dict of rtmi_tuning_t fields (batch_paths, streams, subtile_min_paths, oct_waves_per_cu,
        refill_min0, refill_min, xcd_aware); fields not given keep the library default.  Never changes a pixel.
        devices: list of device indices for the in-library multi-GPU fan-out (rtmi_render_frame_multi); entry 0 is the
        root, an entry may repeat a device.  With more than one entry walk_rays() stripes the frame over them."""
        self.seed, self.device, self.options = int(seed), int(device), int(options)
        self.tuning = dict(tuning) if tuning else None
        self.devices = [int(d) for d in devices] if devices else None
        if self.devices:
            self.device = self.devices[0]

    def _config(self, s):
        _chk(_ffi.lib().rth_caster_config(s.h, self.seed, self.device, self.options))
        devs = self.devices or [self.device]
        arr = (C.c_int32 * len(devs))(*devs)
        _chk(_ffi.lib().rth_caster_set_devices(s.h, arr, len(devs)))
        if self.tuning is None:
            if getattr(s, "_tuned", False):
                _chk(_ffi.lib().rth_caster_set_tuning(s.h, None))
                s._tuned = False
        else:
            t = _ffi.Tuning()
            for k, v in self.tuning.items():
                setattr(t, k, int(v) + 1 if k == "xcd_aware" else int(v))
            _chk(_ffi.lib().rth_caster_set_tuning(s.h, C.byref(t)))
            s._tuned = True

    def walk_rays(self, v, s, data, threads=1, show_progress=False, progress=None, bands=16):
        """progress: callable(thread, row, pixels, stats) -- what DefaultRayCaster sends over its channel per finished row
        (raytrace.rs:1411, :1429-1435).  With a callback (or show_progress=True, which prints one line per band like the
        reference's TUI redraw) the frame is rendered in `bands` row bands, one render call and one tuple each; same image, a
        few per cent slower than in one piece (a band's launches do not overlap the next band's)."""
        if self.devices and len(self.devices) > 1:
            return self.walk_frame_multi(v, s, data)
        if progress is None and not show_progress:
            return self.walk_rows(v, s, 0, v.height, data)
        flat = data.reshape(v.height, v.width * 4)
        total, secs, summed = 0, 0.0, None
        bands = max(1, min(int(bands), v.height))
        for b in range(bands):
            r0, r1 = v.height * b // bands, v.height * (b + 1) // bands
            if r1 == r0:
                continue
            ctx = self.walk_rows(v, s, r0, r1 - r0, flat[r0:r1])
            total += ctx.total_rays
            secs += ctx.seconds
            if summed is None:
                summed = dict(ctx.stats)
            else:
                for k in ("rays", "box_tests", "tri_tests", "full_tests", "nodes", "leaves", "kernel_ms", "trace_ms", "trace_launches",
                          "primary_ms", "bounce_ms", "slow_paths"):
                    summed[k] += ctx.stats[k]
            if progress is not None:
                progress(0, r1 - 1, (r1 - r0) * v.width, {"Rays": ctx.total_rays})
            if show_progress:
                print(f"rows {r0}..{r1 - 1} done: {ctx.stats_line()}", flush=True)
        return ProgressCtx(total, secs, summed)

    def walk_frame_multi(self, v, s, data=None, rgb8=False, stripe_rows=0, out_device_ptr=None, rccl=False):
        """One frame striped over self.devices inside the library.  data: (H, W, 4) f32, or (H, W, 3) u8 with rgb8=True
        (each band is quantised on its device before it crosses to the root).  rccl=True: the bands cross with one ncclGather
        (RTMI_FRAME_RCCL; every entry of self.devices must then be a different device).  Returns ProgressCtx; .per_device
        holds the stats of every device."""
        want = (np.uint8, 3) if rgb8 else (np.float32, 4)
        if data is not None and (data.dtype != want[0] or not data.flags.c_contiguous or data.size != v.height * v.width * want[1]):
            raise ValueError("data must be C-contiguous (H, W, 4) float32, or (H, W, 3) uint8 with rgb8")
        self._config(s)
        n = len(self.devices or [self.device])
        st = _ffi.Stats()
        per = (_ffi.Stats * n)()
        wall = C.c_double(0)
        _chk(_ffi.lib().rth_caster_walk_frame_multi(s.h, v.width, v.height, _p(v.vp12), v.maxdepth, v.samples_per_pixel, stripe_rows,
                                                    (1 if rgb8 else 0) | (2 if rccl else 0), _p(data) if data is not None else None,
                                                    C.c_void_p(out_device_ptr or 0), C.byref(st), per, n, C.byref(wall)))
        ctx = ProgressCtx(st.rays, wall.value, st.as_dict())
        ctx.per_device = [p.as_dict() for p in per]
        return ctx

    def walk_rows(self, v, s, row0, nrows, data):
        """Rows [row0, row0+nrows) only — the unit of multi-GPU image tiling."""
        if data.dtype != np.float32 or not data.flags.c_contiguous or data.size != nrows * v.width * 4:
            raise ValueError("data must be a C-contiguous float32 array of nrows*width*4 elements")
        self._config(s)
        st = _ffi.Stats()
        wall = C.c_double(0)
        _chk(_ffi.lib().rth_caster_walk_rows(s.h, v.width, v.height, _p(v.vp12), v.maxdepth, v.samples_per_pixel, row0,
                                             nrows, _p(data), C.byref(st), C.byref(wall)))
        return ProgressCtx(st.rays, wall.value, st.as_dict())

    def walk_rows_device(self, v, s, row0, nrows, out_ptr, stream_ptr=None):
        """Same, writing nrows*width float4 of device memory at `out_ptr` on HIP stream `stream_ptr`."""
        self._config(s)
        st = _ffi.Stats()
        wall = C.c_double(0)
        _chk(_ffi.lib().rth_caster_walk_rows_device(s.h, v.width, v.height, _p(v.vp12), v.maxdepth, v.samples_per_pixel,
                                                    row0, nrows, C.c_void_p(out_ptr), C.c_void_p(stream_ptr or 0),
                                                    C.byref(st), C.byref(wall)))
        return ProgressCtx(st.rays, wall.value, st.as_dict())

    def walk_tile_device(self, v, s, tile, out_ptr, stream_ptr=None):
        """Striped row set: tile = (row0, nrows, stripe_rows, stripe_step) — rtmi_tile_t."""
        self._config(s)
        st = _ffi.Stats()
        wall = C.c_double(0)
        t = _ffi.Tile(*[int(x) for x in tile])
        _chk(_ffi.lib().rth_caster_walk_tile_device(s.h, v.width, v.height, _p(v.vp12), v.maxdepth, v.samples_per_pixel,
                                                    C.byref(t), C.c_void_p(out_ptr), C.c_void_p(stream_ptr or 0),
                                                    C.byref(st), C.byref(wall)))
        return ProgressCtx(st.rays, wall.value, st.as_dict())

    def quantize_device(self, s, rgba_ptr, npixels, rgb_ptr, stream_ptr=None):
        """write_png's `(c * 255.) as u8` on device memory (f32x4 -> u8x3), enqueued on the stream."""
        self._config(s)
        _chk(_ffi.lib().rth_caster_quantize_device(s.h, C.c_void_p(rgba_ptr), npixels, C.c_void_p(rgb_ptr), C.c_void_p(stream_ptr or 0)))

    def upload(self, s):
        self._config(s)
        _chk(_ffi.lib().rth_caster_upload(s.h))

    def trace(self, s, orig4, dir4):
        """Closest hit per explicit ray (rtmi_trace): -> tri, t, face, stats."""
        o4, d4 = _f(orig4).reshape(-1, 4), _f(dir4).reshape(-1, 4)
        n = o4.shape[0]
        tri, t, face = np.zeros(n, np.uint32), np.zeros(n, np.float32), np.zeros(n, np.uint32)
        self._config(s)
        st = _ffi.Stats()
        _chk(_ffi.lib().rth_caster_trace(s.h, n, _p(o4), _p(d4), _p(tri), _p(t), _p(face), C.byref(st)))
        return tri, t, face, st.as_dict()


def quantize(rgba):
    """write_png's `(c * 255.) as u8` (raytrace.rs:1468-1473)."""
    rgba = _f(rgba).reshape(-1, 4)
    out = np.zeros((rgba.shape[0], 3), np.uint8)
    _ffi.lib().rth_quantize(_p(rgba), rgba.shape[0], _p(out))
    return out


# ---------------------------------------------------------------- scenes of the benchmark configs
def canonical_scene(obj_path, accel="octree", maxdepth=10, minobjs=19, teapot_surface=None, threads=0, gpu_build=None):
    """The scene of raytrace/src/main.rs:116-164."""
    s = Scene(with_dummy=True)
    tsurf = teapot_surface or SurfaceKind.Matte(make_color(252, 119, 0), 0.2)
    s.extend_parse_obj(obj_path, [0.0, 0.5, 5.0], 1.0, create_transform(unit([0.0, 0.3, 1.0]), to_radians(270.0)), tsurf, 0.05)
    side = SurfaceKind.Matte(make_color(40, 40, 40), 0.2)
    s.extend_make_disk([4.0, 4.0, 7.0], unit([-0.3, -0.55, -0.5]), 2.0, 0.1, 50,
                       SurfaceKind.Reflective(0.0002, make_color(230, 230, 230), 0.7), side, -1.0)
    s.extend_make_disk([4.0, -3.0, 5.0], unit([-0.5, 2.0, -0.5]), 1.0, 0.04, 50,
                       SurfaceKind.Reflective(0.002, make_color(230, 230, 230), 0.7), side, -1.0)
    s.populate_triangle_numbers()
    if accel == "octree":
        s.build_bounding_box([0.0, 0.0, 20.1], 20.0, maxdepth, minobjs, threads, gpu_device=gpu_build)
    elif accel == "trivial":
        s.build_trivial_bounding_box([0.0, 0.0, 0.0], 20.0)
    return s


def grid_scene(obj_path, maxdepth=10, minobjs=19, n=8, threads=0, gpu_build=None):
    """BASELINE config 5: n instances of teapot_tri.obj on a 2x2x2 grid (spacing 9 units) inside the canonical
    root box — the octree-traversal stress scene (8 x 6320 + 1 = 50 561 triangles)."""
    s = Scene(with_dummy=True)
    surfs = [SurfaceKind.Matte(make_color(252, 119, 0), 0.2), SurfaceKind.Reflective(0.01, make_color(200, 200, 220), 0.6),
             SurfaceKind.Solid(make_color(30, 160, 60)), SurfaceKind.Matte(make_color(200, 40, 40), 0.35)]
    k = 0
    for iz in range(2):
        for iy in range(2):
            for ix in range(2):
                if k >= n:
                    break
                off = [-4.5 + 9.0 * ix, -4.5 + 9.0 * iy, 9.0 + 9.0 * iz]
                s.extend_parse_obj(obj_path, off, 1.0, create_transform(unit([0.0, 0.3, 1.0]), to_radians(270.0 + 20.0 * k)),
                                   surfs[k % 4], 0.05 if k % 2 == 0 else 0.0)
                k += 1
    s.populate_triangle_numbers()
    s.build_bounding_box([0.0, 0.0, 20.1], 20.0, maxdepth, minobjs, threads, gpu_device=gpu_build)
    return s


def canonical_viewport(w, h, maxdepth=5, samples=1):
    """main.rs:166-173"""
    aspect = np.float32(h) / np.float32(w)
    return create_viewport((w, h), (1.0, float(np.float32(1.0) * aspect)), [2.0, 0.0, 0.0], unit([0.0, 0.0, 1.0]), 90.0,
                           to_radians(0.0), maxdepth, samples)
