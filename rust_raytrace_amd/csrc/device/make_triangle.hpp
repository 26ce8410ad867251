// make_triangle.hpp — `make_triangle` (raytrace_lib/src/raytrace.rs:340-383) as a HIP kernel: the per-triangle
// precompute (centroid by intersecting two medians, inward unit edge normals + distances, unit normal, bounding
// radius) for meshes too big to prepare on the host.  One thread per triangle, 4-lane V4 arithmetic exactly as the
// reference (vec4.hpp); bit-identical to the host mirror (tests/test_gpu_parity.py::test_make_triangles_gpu).
// Included by rtmi_device.hip.
#pragma once

namespace rtmi {

__device__ inline V4 vcross(V4 a, V4 b) {  // raytrace.rs:80-90 (swizzles [1,2,0,3] and [2,0,1,3])
    return V4{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x, a.w * b.w - a.w * b.w};
}
struct RayF { V4 orig, dir; };
__device__ inline RayF mkray(V4 orig, V4 dir) { return RayF{orig, vunit(dir)}; }            // make_ray, :201-210
__device__ inline V4 ray_at(const RayF& r, float t) { return vadd(vmul(r.dir, t), r.orig); }  // :227-229
__device__ inline float comp(V4 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }

// ray_intersect_helper on coordinates (i, j) (raytrace.rs:212-224)
__device__ inline bool solve2(const RayF& s, const RayF& r, int i, int j, float& t1, float& t2) {
    const float det = comp(r.dir, i) * comp(s.dir, j) - comp(r.dir, j) * comp(s.dir, i);
    if (fabsf(det) < 0.0001f) return false;
    const float dx = comp(r.orig, i) - comp(s.orig, i);
    const float dy = comp(r.orig, j) - comp(s.orig, j);
    t1 = (dy * comp(r.dir, i) - dx * comp(r.dir, j)) / det;
    t2 = (dy * comp(s.dir, i) - dx * comp(s.dir, j)) / det;
    return true;
}

// out: 20 floats per triangle = incenter3 norm3 r2 sides9 side_lens3 (+ pad); ok[i] = 0 where the reference panics
__global__ void __launch_bounds__(256) k_make_triangles(uint32_t n, const float* __restrict__ pts9, float* __restrict__ out20,
                                                        uint32_t* __restrict__ ok) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float* p = pts9 + (size_t)i * 9;
        const V4 pt[3] = {mk(p[0], p[1], p[2]), mk(p[3], p[4], p[5]), mk(p[6], p[7], p[8])};
        const V4 ab = vsub(pt[1], pt[0]), ac = vsub(pt[2], pt[0]), bc = vsub(pt[2], pt[1]);
        const RayF ra = mkray(pt[0], vadd(ac, ab));
        const RayF rb = mkray(pt[1], vadd(bc, vmul(ab, -1.f)));
        float t1 = 0.f, t2 = 0.f;
        bool good = solve2(ra, rb, 0, 1, t1, t2) || solve2(ra, rb, 0, 2, t1, t2) || solve2(ra, rb, 1, 2, t1, t2);  // :231-256
        const V4 p1 = ray_at(ra, t1), p2 = ray_at(rb, t2);
        good = good && (vlen2(vsub(p2, p1)) < 0.01f);  // :261-266
        const V4 inc = p1;
        float* o = out20 + (size_t)i * 20;
        V4 sides[3];
        for (int k = 0; k < 3; k++) {
            const V4 vedge = vsub(pt[(k + 1) % 3], pt[k]);
            const V4 po = vsub(inc, pt[k]);
            const V4 pc = vmul(vedge, vdot(vedge, po) / vlen2(vedge));
            const V4 oc = vsub(pc, po);
            sides[k] = vunit(oc);
            o[7 + 3 * k] = sides[k].x; o[8 + 3 * k] = sides[k].y; o[9 + 3 * k] = sides[k].z;
            o[16 + k] = vlen(oc);
        }
        const V4 norm = vunit(vcross(sides[0], sides[1]));
        float r2 = 0.0f;
        for (int k = 0; k < 3; k++) r2 = fmaxf(r2, vlen2(vsub(pt[k], inc)));
        o[0] = inc.x; o[1] = inc.y; o[2] = inc.z;
        o[3] = norm.x; o[4] = norm.y; o[5] = norm.z;
        o[6] = r2;
        o[19] = 0.f;
        ok[i] = good ? 1u : 0u;
    }
}

}  // namespace rtmi
