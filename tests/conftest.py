import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
TEAPOT_TRI = os.path.join(GOLDEN, "teapot_tri.obj")
TEAPOT = os.path.join(GOLDEN, "teapot.obj")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def bits(a):
    """Bit pattern view for exact float comparison: distinguishes -0/+0; every NaN is
    mapped to one pattern because the sign/payload of a generated NaN is a property
    of the machine (x86 SSE makes 0xFFC00000, gfx950 makes 0x7FC00000), not of the algorithm."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).copy()
    u[np.isnan(a)] = 0x7FC00000
    return u


def assert_bits_equal(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    ba, bb = bits(a), bits(b)
    if not np.array_equal(ba, bb):
        bad = np.argwhere(ba != bb)
        first = tuple(bad[0])
        raise AssertionError(f"{what}: {len(bad)} of {ba.size} floats differ; first at {first}: {a[first]!r} vs {b[first]!r}")


# ---------------------------------------------------------------- scene recipes (same calls on oracle and product)
def build_pair(recipe):
    """recipe(api) builds a scene with either API; returns (oracle_scene, product_scene)."""
    from oracle import orc
    from rust_raytrace_amd import raytrace as R
    return recipe(OracleApi(orc)), recipe(ProductApi(R))


class OracleApi:
    def __init__(self, orc):
        self.m = orc
        self.kind = "oracle"

    def scene(self):
        return self.m.Scene(with_dummy=True)

    def solid(self, c):
        return self.m.Surface(self.m.SOLID, self.m.make_color(*c))

    def matte(self, c, a):
        return self.m.Surface(self.m.MATTE, self.m.make_color(*c), a)

    def reflective(self, sc, c, a):
        return self.m.Surface(self.m.REFLECTIVE, self.m.make_color(*c), a, sc)

    def add_obj(self, s, *a):
        s.add_obj(*a)

    def add_disk(self, s, *a):
        s.add_disk(*a)

    def add_sphere(self, s, *a):
        s.add_sphere(*a)

    def add_triangle(self, s, *a):
        s.add_triangle(*a)

    def add_analytic_sphere(self, s, *a):
        s.add_analytic_sphere(*a)

    def transform(self, d, roll_deg):
        return self.m.create_transform(self.m.unit(d), self.m.to_radians(roll_deg))

    def unit(self, v):
        return self.m.unit(v)


class ProductApi:
    def __init__(self, R):
        self.m = R
        self.kind = "product"

    def scene(self):
        return self.m.Scene(with_dummy=True)

    def solid(self, c):
        return self.m.SurfaceKind.Solid(self.m.make_color(*c))

    def matte(self, c, a):
        return self.m.SurfaceKind.Matte(self.m.make_color(*c), a)

    def reflective(self, sc, c, a):
        return self.m.SurfaceKind.Reflective(sc, self.m.make_color(*c), a)

    def add_obj(self, s, *a):
        s.extend_parse_obj(*a)

    def add_disk(self, s, *a):
        s.extend_make_disk(*a)

    def add_sphere(self, s, *a):
        s.extend_make_sphere(*a)

    def add_triangle(self, s, *a):
        s.push_triangle(*a)

    def add_analytic_sphere(self, s, *a):
        s.push_analytic_sphere(*a)

    def transform(self, d, roll_deg):
        return self.m.create_transform(self.m.unit(d), self.m.to_radians(roll_deg))

    def unit(self, v):
        return self.m.unit(v)


def recipe_canonical(accel="octree", maxdepth=10, minobjs=19, solid_teapot=False, obj=TEAPOT_TRI):
    """raytrace/src/main.rs:116-164"""
    def r(api):
        s = api.scene()
        tsurf = api.solid((252, 119, 0)) if solid_teapot else api.matte((252, 119, 0), 0.2)
        api.add_obj(s, obj, [0.0, 0.5, 5.0], 1.0, api.transform([0.0, 0.3, 1.0], 270.0), tsurf, 0.05)
        side = api.matte((40, 40, 40), 0.2)
        api.add_disk(s, [4.0, 4.0, 7.0], api.unit([-0.3, -0.55, -0.5]), 2.0, 0.1, 50, api.reflective(0.0002, (230, 230, 230), 0.7), side, -1.0)
        api.add_disk(s, [4.0, -3.0, 5.0], api.unit([-0.5, 2.0, -0.5]), 1.0, 0.04, 50, api.reflective(0.002, (230, 230, 230), 0.7), side, -1.0)
        s.populate_triangle_numbers()
        if accel == "octree":
            s.build_bounding_box([0.0, 0.0, 20.1], 20.0, maxdepth, minobjs)
        else:
            s.build_trivial_bounding_box([0.0, 0.0, 0.0], 20.0)
        return s
    return r


def recipe_grid(maxdepth=10, minobjs=19, n=8, obj=TEAPOT_TRI):
    """BASELINE config 5: `n` instances of teapot_tri.obj on a 2x2x2 grid, spacing 9 units (instances do not
    overlap; the teapot spans about +-3.4), inside the canonical root box (centre (0,0,20.1), half 20)."""
    def r(api):
        s = api.scene()
        surfs = [api.matte((252, 119, 0), 0.2), api.reflective(0.01, (200, 200, 220), 0.6), api.solid((30, 160, 60)),
                 api.matte((200, 40, 40), 0.35)]
        k = 0
        for iz in range(2):
            for iy in range(2):
                for ix in range(2):
                    if k >= n:
                        break
                    off = [-4.5 + 9.0 * ix, -4.5 + 9.0 * iy, 9.0 + 9.0 * iz]
                    api.add_obj(s, obj, off, 1.0, api.transform([0.0, 0.3, 1.0], 270.0 + 20.0 * k), surfs[k % 4], 0.05 if k % 2 == 0 else 0.0)
                    k += 1
        s.populate_triangle_numbers()
        s.build_bounding_box([0.0, 0.0, 20.1], 20.0, maxdepth, minobjs)
        return s
    return r


def recipe_circles(accel="octree", maxdepth=6, minobjs=8):
    """BASELINE config 1 ("circles"): the reference has no analytic sphere at this
    revision, so the scene is a ground disk plus tessellated make_sphere balls
     with Solid / Matte / Reflective surfaces."""
    def r(api):
        s = api.scene()
        api.add_disk(s, [0.0, -1.5, 8.0], api.unit([0.0, 1.0, 0.05]), 6.0, 0.05, 24, api.matte((90, 140, 90), 0.3), api.solid((20, 20, 20)), -1.0)
        api.add_sphere(s, [-1.6, 0.0, 7.0], 1.0, (8, 16), api.solid((200, 30, 30)), 0.0)
        api.add_sphere(s, [0.9, 0.2, 6.0], 0.8, (8, 16), api.matte((30, 60, 200), 0.4), 0.04)
        api.add_sphere(s, [2.8, 0.5, 9.0], 1.3, (8, 16), api.reflective(0.05, (220, 220, 220), 0.8), 0.0)
        api.add_sphere(s, [0.0, 2.2, 11.0], 1.1, (6, 12), api.matte((240, 200, 40), 0.15), 0.0)
        s.populate_triangle_numbers()
        if accel == "octree":
            s.build_bounding_box([0.0, 0.0, 10.0], 10.0, maxdepth, minobjs)
        else:
            s.build_trivial_bounding_box([0.0, 0.0, 0.0], 20.0)
        return s
    return r


def recipe_circles_analytic():
    """BASELINE config 1 as north_star words it ("circles scene (few spheres)") with ANALYTIC spheres -- a build-defined
    primitive the reference does not have (parity unpinned): a ground disk of triangles in a small octree plus four
    spheres, Solid / Matte / Reflective, one of them with the camera looking through its inside."""
    def r(api):
        s = api.scene()
        api.add_disk(s, [0.0, -1.5, 8.0], api.unit([0.0, 1.0, 0.05]), 6.0, 0.05, 24, api.matte((90, 140, 90), 0.3), api.solid((20, 20, 20)), -1.0)
        s.populate_triangle_numbers()
        s.build_bounding_box([0.0, 0.0, 10.0], 10.0, 5, 8)
        api.add_analytic_sphere(s, [-1.6, 0.0, 7.0], 1.0, api.solid((200, 30, 30)))
        api.add_analytic_sphere(s, [0.9, 0.2, 6.0], 0.8, api.matte((30, 60, 200), 0.4))
        api.add_analytic_sphere(s, [2.8, 0.5, 9.0], 1.3, api.reflective(0.05, (220, 220, 220), 0.8))
        api.add_analytic_sphere(s, [2.0, 0.0, 0.1], 1.5, api.matte((240, 200, 40), 0.15))   # contains the camera (2, 0, 0)
        return s
    return r


def recipe_axis_box():
    """Axis-aligned geometry: exercises rays parallel to planes (t = +-inf / NaN),
    zero direction components and exact ties."""
    def r(api):
        s = api.scene()
        m = api.matte((200, 200, 200), 0.5)
        sol = api.solid((10, 200, 10))
        q = [([-1, -1, 5], [1, -1, 5], [1, 1, 5]), ([-1, -1, 5], [1, 1, 5], [-1, 1, 5]),      # z = 5 wall
             ([-1, -1, 3], [-1, -1, 5], [-1, 1, 5]), ([-1, -1, 3], [-1, 1, 5], [-1, 1, 3]),   # x = -1 wall
             ([-1, -1, 3], [1, -1, 3], [1, -1, 5]), ([-1, -1, 3], [1, -1, 5], [-1, -1, 5]),   # y = -1 floor
             ([0, 0, 4], [0.5, 0, 4], [0, 0.5, 4]), ([0, 0, 4], [0.5, 0, 4], [0, 0.5, 4])]    # duplicate: exact tie
        for i, pts in enumerate(q):
            api.add_triangle(s, np.array(pts, np.float32), m if i % 2 == 0 else sol, 0.05 if i < 4 else 0.0)
        s.populate_triangle_numbers()
        s.build_bounding_box([0.0, 0.0, 4.0], 4.0, 4, 2)
        return s
    return r


@pytest.fixture(scope="session")
def canonical_pair():
    return build_pair(recipe_canonical())


@pytest.fixture(scope="session")
def circles_pair():
    return build_pair(recipe_circles())
