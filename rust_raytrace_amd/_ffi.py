"""ctypes loader for librtmi.so (HIP kernels + C ABI + C++ host mirror).

The product path has no CPU fallback: if the library is missing this module
raises, and every render entry point fails with the library's own error when
no HIP device is visible.
"""
import ctypes as C
import os
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(_CSRC, "librtmi.so")


class Stats(C.Structure):
    """rtmi_stats_t (include/rtmi.h)."""
    _fields_ = [("rays", C.c_uint64), ("box_tests", C.c_uint64), ("tri_tests", C.c_uint64),
                ("full_tests", C.c_uint64), ("nodes", C.c_uint64), ("leaves", C.c_uint64),
                ("kernel_ms", C.c_double), ("trace_ms", C.c_double), ("trace_launches", C.c_uint32),
                ("streams", C.c_uint32), ("render_ms", C.c_double), ("band_copy_ms", C.c_double),
                ("deinterleave_ms", C.c_double), ("primary_ms", C.c_double), ("bounce_ms", C.c_double),
                ("peer_access", C.c_int32), ("pipeline", C.c_uint32), ("slow_paths", C.c_uint32), ("reserved", C.c_uint32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Tile(C.Structure):
    """rtmi_tile_t (include/rtmi.h)."""
    _fields_ = [("row0", C.c_uint32), ("nrows", C.c_uint32), ("stripe_rows", C.c_uint32), ("stripe_step", C.c_uint32)]


class Tuning(C.Structure):
    """rtmi_tuning_t (include/rtmi.h)."""
    _fields_ = [("batch_paths", C.c_uint64), ("streams", C.c_uint32), ("subtile_min_paths", C.c_uint32),
                ("oct_waves_per_cu", C.c_uint32), ("refill_min0", C.c_uint32), ("refill_min", C.c_uint32),
                ("xcd_aware", C.c_uint32), ("kernel", C.c_uint32), ("pipeline", C.c_uint32), ("slow_path_off", C.c_uint32)]


def build(force=False):
    """Compile librtmi.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", _CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _CSRC, "-j4"], stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C rust_raytrace_amd/csrc). There is no CPU fallback for the render path.")
    # One HIP runtime per process: torch bundles its own libamdhip64 (SONAME libamdhip64.so.7) and
    # librtmi.so needs that SONAME, so with torch loaded FIRST both share torch's runtime (streams and
    # device pointers are then interchangeable).  Loaded the other way round the process ends up with
    # two runtimes and torch sees no GPU.  Without torch the system /opt/rocm runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u64, u32, f32, i32 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_float, C.c_int
    L.rtmi_last_error.restype = C.c_char_p
    L.rth_last_error.restype = C.c_char_p
    L.rtmi_device_count.restype = i32
    L.rth_to_radians.restype = f32
    L.rth_to_radians.argtypes = [f32]
    L.rth_scene_new.restype = vp
    L.rth_scene_new.argtypes = [i32]
    L.rth_scene_free.argtypes = [vp]
    L.rth_num_tris.restype = u64
    L.rth_num_tris.argtypes = [vp]
    L.rth_make_color.argtypes = [C.c_uint8, C.c_uint8, C.c_uint8, vp]
    L.rth_unit.argtypes = [vp, vp]
    L.rth_create_transform.argtypes = [vp, f32, vp]
    L.rth_create_viewport.argtypes = [u32, u32, f32, f32, vp, vp, f32, f32, vp]
    L.rth_add_triangle.argtypes = [vp, vp, u32, vp, f32, f32, f32]
    L.rth_add_triangles_gpu.argtypes = [vp, vp, u64, u32, vp, f32, f32, f32, i32]
    L.rth_add_obj.argtypes = [vp, C.c_char_p, vp, f32, vp, u32, vp, f32, f32, f32]
    L.rth_add_obj_mode.argtypes = [vp, C.c_char_p, vp, f32, vp, u32, vp, f32, f32, f32, u32]
    L.rth_add_disk.argtypes = [vp, vp, vp, f32, f32, u64, u32, vp, f32, f32, u32, vp, f32, f32, f32]
    L.rth_add_sphere.argtypes = [vp, vp, f32, u64, u64, u32, vp, f32, f32, f32]
    L.rth_populate_triangle_numbers.argtypes = [vp]
    L.rth_add_analytic_sphere.argtypes = [vp, vp, f32, u32, vp, f32, f32]
    L.rth_build_bounding_box.argtypes = [vp, vp, f32, u64, u64, u32]
    L.rth_build_bounding_box_gpu.argtypes = [vp, vp, f32, u64, u64, i32]
    L.rth_build_trivial_bounding_box.argtypes = [vp, vp, f32]
    L.rth_box_contains_polygon.argtypes = [vp, vp, f32, u64]
    L.rth_face_contains_triangle.argtypes = [vp, vp, vp, f32, u64]
    L.rth_get_triangles.argtypes = [vp, vp, vp, vp]
    L.rth_tree_sizes.argtypes = [vp, vp, vp]
    L.rth_tree_get.argtypes = [vp, vp, vp, vp]
    L.rth_caster_config.argtypes = [vp, u64, i32, u32]
    L.rth_caster_upload.argtypes = [vp]
    L.rth_caster_set_tuning.argtypes = [vp, vp]
    L.rth_caster_set_devices.argtypes = [vp, vp, u32]
    L.rth_caster_walk_frame_multi.argtypes = [vp, u32, u32, vp, u64, u64, u32, u32, vp, vp, vp, vp, u32, vp]
    L.rth_caster_walk_rows.argtypes = [vp, u32, u32, vp, u64, u64, u64, u64, vp, vp, vp]
    L.rth_caster_walk_rows_device.argtypes = [vp, u32, u32, vp, u64, u64, u64, u64, vp, vp, vp, vp]
    L.rth_caster_walk_tile_device.argtypes = [vp, u32, u32, vp, u64, u64, vp, vp, vp, vp, vp]
    L.rth_caster_quantize_device.argtypes = [vp, vp, u64, vp, vp]
    L.rth_caster_trace.argtypes = [vp, u64, vp, vp, vp, vp, vp, vp]
    L.rth_quantize.argtypes = [vp, u64, vp]
    _lib = L
    return L


# every symbol include/rtmi.h and include/rtmi_host.h declare
RTMI_SYMBOLS = ["rtmi_device_count", "rtmi_scene_create", "rtmi_scene_destroy", "rtmi_scene_set_options",
                "rtmi_scene_get_tuning", "rtmi_scene_set_tuning", "rtmi_scene_set_spheres", "rtmi_scene_set_corners", "rtmi_render", "rtmi_render_frame_multi",
                "rtmi_render_device", "rtmi_render_tile_device", "rtmi_trace", "rtmi_quantize", "rtmi_quantize_device", "rtmi_make_triangles", "rtmi_builder_create", "rtmi_builder_filter", "rtmi_builder_destroy", "rtmi_last_error"]
RTH_SYMBOLS = ["rth_last_error", "rth_make_color", "rth_unit", "rth_to_radians", "rth_create_transform",
               "rth_create_viewport", "rth_scene_new", "rth_scene_free", "rth_num_tris", "rth_add_triangle", "rth_add_triangles_gpu", "rth_add_obj", "rth_add_obj_mode",
               "rth_add_disk", "rth_add_sphere", "rth_add_analytic_sphere", "rth_populate_triangle_numbers", "rth_build_bounding_box", "rth_build_bounding_box_gpu",
               "rth_build_trivial_bounding_box", "rth_box_contains_polygon", "rth_face_contains_triangle",
               "rth_get_triangles", "rth_tree_sizes", "rth_tree_get", "rth_caster_config", "rth_caster_walk_rows",
               "rth_caster_walk_rows_device", "rth_caster_walk_tile_device", "rth_caster_trace", "rth_caster_upload", "rth_caster_set_tuning", "rth_caster_set_devices", "rth_caster_walk_frame_multi", "rth_caster_quantize_device", "rth_quantize"]
