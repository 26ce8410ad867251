// build_octree.hpp — the octree builder's overlap test on the GPU.
//
// build_bounding_box_helper (raytrace_lib/src/raytrace.rs:795-845) filters, for every candidate child box, the
// parent's surviving triangle list with box_contains_polygon (raytrace.rs:753-779): ~9.3 M tests for the canonical
// scene, ~75 M for the 8-teapot grid, each up to 6 face_contains_triangle evaluations (raytrace.rs:645-729) with
// divisions and square roots.  That is the whole cost of the build; the list bookkeeping around it is linear.  The
// kernel below evaluates the test for every (box, candidate) pair of one tree level, one thread per pair, in the
// reference's operation order on 4-lane values (vec4.hpp: ordered dot products, unit = x * (1/len), lane 3 carried),
// compiled with -ffp-contract=off: the flags -- and therefore the tree -- are identical to the host builder's.
// Included by rtmi_device.hip.
#pragma once

namespace rtmi {

struct BTri { V4 incenter, norm, corner[3]; };

__device__ inline float vget(const V4& v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }

// raytrace.rs:636-643
__device__ inline bool b_contains_point(const V4& orig, float len2, const V4& p) {
    const V4 op = vsub(p, orig);
    return fabsf(op.x) < len2 && fabsf(op.y) < len2 && fabsf(op.z) < len2;
}

struct BRay { V4 orig, dir, inv; };
__device__ inline BRay b_make_ray(const V4& orig, const V4& dir) {  // raytrace.rs:201-210
    const V4 du = vunit(dir);
    return BRay{orig, du, mk(1.f / du.x, 1.f / du.y, 1.f / du.z)};
}
__device__ inline V4 b_at(const BRay& r, float t) { return vadd(vmul(r.dir, t), r.orig); }  // raytrace.rs:227-229

// raytrace.rs:645-729; `axis` 0..5 = +x -x +y -y +z -z (the order of raytrace.rs:765-777)
__device__ inline bool b_face_contains_triangle(const V4& p, int axis, float len2, const BTri& t) {
    const float sgn = (axis & 1) ? -1.f : 1.f;
    const int ax = axis >> 1;
    const V4 norm = mk(ax == 0 ? sgn : 0.f, ax == 1 ? sgn : 0.f, ax == 2 ? sgn : 0.f);
    const float h1 = vdot(norm, vadd(p, vmul(norm, len2)));
    const float h2 = vdot(t.norm, t.incenter);
    const float nn = vdot(norm, t.norm);
    const float c1 = (h1 - h2 * nn) / (1.f - nn * nn);
    const float c2 = (h2 - h1 * nn) / (1.f - nn * nn);
    const BRay line_tmp = b_make_ray(vadd(vmul(norm, c1), vmul(t.norm, c2)), vcross(norm, t.norm));
    // how far before the box does the line start (raytrace.rs:659-685): the two axes the face normal is zero on
    float tmin = FLT_MAX;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (k == ax) continue;
        const float t1 = (vget(p, k) - len2 - vget(line_tmp.orig, k)) * vget(line_tmp.inv, k);
        const float t2 = (vget(p, k) + len2 - vget(line_tmp.orig, k)) * vget(line_tmp.inv, k);
        tmin = fminf(tmin, fminf(t1, t2));
    }
    const BRay line = (tmin > 0.f) ? line_tmp : b_make_ray(b_at(line_tmp, tmin * 2.f), line_tmp.dir);
    // clip against the two slabs of the face (raytrace.rs:687-716)
    tmin = -FLT_MAX;
    float tmax = FLT_MAX;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (k == ax) continue;
        const float t1 = (vget(p, k) - len2 - vget(line.orig, k)) * vget(line.inv, k);
        const float t2 = (vget(p, k) + len2 - vget(line.orig, k)) * vget(line.inv, k);
        tmin = fmaxf(tmin, fminf(t1, t2));
        tmax = fminf(tmax, fmaxf(t1, t2));
    }
    if (tmax < tmin) return false;
    // does the (infinite) line separate two corners (raytrace.rs:718-728)
    V4 off[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float tk = vdot(vsub(t.corner[k], line.orig), line.dir) / vlen2(line.dir);
        off[k] = vsub(b_at(line, tk), t.corner[k]);
    }
    return vdot(off[0], off[1]) < 0.f || vdot(off[0], off[2]) < 0.f || vdot(off[1], off[2]) < 0.f;
}

// raytrace.rs:753-779
__device__ inline bool b_box_contains_polygon(const V4& orig, float len2, const BTri& t) {
    if (b_contains_point(orig, len2, t.incenter)) return true;
#pragma unroll
    for (int k = 0; k < 3; k++)
        if (b_contains_point(orig, len2, t.corner[k])) return true;
    for (int a = 0; a < 6; a++)
        if (b_face_contains_triangle(orig, a, len2, t)) return true;
    return false;
}

// One work item = 256 consecutive candidates of one box.  items: (box, first candidate of the chunk); boxes: geometry
// float4 (centre, half edge) + uint4 (first candidate in `cand`, candidate count, first output flag lo, hi).
__global__ void __launch_bounds__(256) k_box_contains(const float* __restrict__ tris15, const float4* __restrict__ box_geo,
                                                      const uint4* __restrict__ box_rng, const uint2* __restrict__ items,
                                                      uint32_t nitems, const uint32_t* __restrict__ cand, uint8_t* __restrict__ keep) {
    for (uint32_t it = blockIdx.x; it < nitems; it += gridDim.x) {
        const uint2 w = items[it];
        const float4 g = box_geo[w.x];
        const uint4 rg = box_rng[w.x];
        const uint32_t j = w.y + threadIdx.x;
        if (j >= rg.y) continue;
        const float* p = tris15 + (size_t)cand[rg.x + j] * 15;
        BTri t;
        t.incenter = mk(p[0], p[1], p[2]);
        t.norm = mk(p[3], p[4], p[5]);
#pragma unroll
        for (int k = 0; k < 3; k++) t.corner[k] = mk(p[6 + 3 * k], p[7 + 3 * k], p[8 + 3 * k]);
        const uint64_t o = ((uint64_t)rg.w << 32 | rg.z) + j;
        keep[o] = b_box_contains_polygon(mk(g.x, g.y, g.z), g.w, t) ? 1 : 0;
    }
}

}  // namespace rtmi
