"""rust_raytrace_amd — MI355X-native path-tracing core for rust_raytrace.

Only the hot path lives here: csrc/device (HIP kernels + the C ABI of
include/rtmi.h) and the host-side mirror of the reference's scene / RayCaster
interface (csrc/host, raytrace.py).  Importing the package loads librtmi.so and
fails loudly when it has not been built.
"""
from . import _ffi
from . import raytrace  # noqa: F401
from .raytrace import HipRayCaster, Scene, SurfaceKind, Viewport, create_transform, create_viewport, make_color  # noqa: F401

_ffi.lib()
