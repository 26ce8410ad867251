// rtmi_device.hip — HIP kernels (gfx950) and the C ABI of include/rtmi.h.
//
// Pipeline for one batch of N = pixels*spp paths ("wavefront" formulation of
// the reference's recursive project_ray, raytrace_lib/src/raytrace.rs:1256-1295):
//
//   k_gen     pixel_ray() for every (pixel, sample)        raytrace.rs:1374-1394
//   for pass k = 0 .. maxdepth-1 (remaining depth = maxdepth-k):
//     k_trace closest hit for every queued ray              raytrace.rs:909-1050, 400-439
//     k_shade color_ray(): terminal colour, or push the surface and emit
//             the bounce ray into the next queue, compacted with wave
//             ballot + prefix sum                           raytrace.rs:1199-1254, 278-301
//   k_accum   per pixel: ordered sum over samples * (1/spp) raytrace.rs:1414-1426
//
// The recursion `mix(c_1, mix(c_2, ...))` is evaluated inside-out when a path
// terminates (fold over the per-path surface stack), which is bit-identical to
// the nested calls; see DESIGN.md.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (strict IEEE f32:
// correctly rounded / and sqrtf are hipcc defaults, contraction is not).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is dlopen'ed when RTMI_FRAME_RCCL is asked for
#include <dlfcn.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "../../../include/rtmi.h"
#include "vec4.hpp"

// layouts the language bindings mirror (rust_raytrace_amd/_ffi.py, INTEGRATION.md ffi.rs; tests/test_host_cpu.py)
static_assert(sizeof(rtmi_stats_t) == 128 && sizeof(rtmi_tuning_t) == 48 && sizeof(rtmi_tile_t) == 16 && sizeof(rtmi_box_t) == 32 &&
              sizeof(rtmi_triangle_t) == 104 && sizeof(rtmi_viewport_t) == 64 && sizeof(rtmi_sphere_t) == 40, "ABI struct layout changed");

namespace rtmi {

// ---------------------------------------------------------------- layout
// HBM-resident scene (see DESIGN.md "Data layout"):
//  nodes   : one 32-B record per BoundingBox, children of a box contiguous
//  refs    : u32 triangle indices of all leaves, leaf ranges contiguous
//  tplane  : 2 x float4 per triangle  (incenter.xyz, r2) (norm.xyz, material)
//  tedge   : 4 x float4 per triangle  (side_k.xyz, side_len_k) x3, (thr_k x3, 0)
//            thr_k = side_len_k * (1 - edge_thickness)      raytrace.rs:419
//  mats    : 2 x float4 per distinct surface (color.rgb, alpha) (scattering, kind, 0, 0)
struct DNode {
    float cx, cy, cz, half;
    uint32_t first, count, is_leaf, pad;
};

struct DScene {
    const DNode* nodes;
    const uint32_t* refs;
    const float4* tplane;
    const float4* tedge;
    const float4* mats;
    uint32_t nnodes, ntris, nmats, levels;
    // exact-octree form (trace_oct.hpp); null when the tree is not an exact octree
    const uint4* fnodes;
    const uint4* oblocks;
    const uint32_t* wlinks;  // 8 explicit first-block indices per FN_WIDE box
    // analytic spheres (a build-defined extension, rtmi_sphere_t): 2 x float4 each, (centre, radius) (surface id, 0, 0, 0)
    const float4* spheres;
    uint32_t nspheres;
    float root_half;
    uint32_t olevels;
    uint32_t noblocks;  // reference blocks in `oblocks`
};
#define RTMI_FN_WIDE 0x10000u

// which octree kernel tune.kernel == 0 selects (measured on MI355X)
#define RTMI_DEFAULT_POOL 0
#define RTMI_MAX_PASSES 32
#define RTMI_MAX_STREAMS 4  // interleaved sub-tiles of one tile, each on its own internal stream
struct DCtrl {
    uint32_t count[RTMI_MAX_PASSES + 1];  // rays queued for pass k
    uint32_t head[RTMI_MAX_PASSES + 1];   // work-fetch cursor of pass k
    uint32_t xhead[RTMI_MAX_PASSES + 1][8];  // octree kernel: one cursor per XCD range of the queue
    // slow-path queue (rays with an exactly-zero direction component, trace_oct.hpp): entries pushed so far, and per
    // consumer launch k its range [slo[k], shi[k]) (k_slow_snapshot) and work-fetch cursor
    uint32_t scount;
    uint32_t slo[RTMI_MAX_PASSES + 1], shi[RTMI_MAX_PASSES + 1], shead[RTMI_MAX_PASSES + 1];
    unsigned long long rays;              // sum of count[] (the "Rays" statistic)
    unsigned long long counters[5];       // box_tests tri_tests full_tests nodes leaves
    unsigned long long dbg[16];           // step statistics of the counting build (tools/step_stats.py)
};

// The slow-path queue of one stream's batch.  A ray whose unit direction has an exactly-zero component skips that axis'
// slab in BoundingBox::collides (raytrace.rs:872, :882, :892: the origin is not checked against the slab), so it enters
// every box of the perpendicular plane: ~150 x the work of an ordinary ray (6 000 box + 36 000 triangle tests on the
// canonical scene), 14 ms for the one lane that traces it.  78 of the 268 M primary rays of config 3 are such rays (the
// two pixel rows and columns next to the camera axis, where `row + v_off` rounds to the axis), ~20 bounce rays per frame.
// Their work is nothing, their LATENCY is: a lane that meets one near the end of a launch holds the launch (and, in the
// primary pass, its whole wave) for up to 14 ms -- 10 % of a 1/8-frame tile.  Producers (k_path_primary, k_shade) put such
// rays here instead of the ordinary queue; k_path_slow traces their paths to the end on a side stream, one path per
// wave, concurrently with the following passes.  Same device functions, same image.
struct SlowQ {
    float4* o; float4* d; uint32_t* path; uint32_t* bounce;
    uint32_t cap;
};
#define RTMI_SLOW_CAP 16384u
__device__ inline bool has_zero_component(float x, float y, float z) { return (x == 0.f) | (y == 0.f) | (z == 0.f); }
// true when the ray was queued for the slow path (false: the queue is full, the caller keeps the ray)
__device__ inline bool slow_push(const SlowQ& q, DCtrl* ctrl, float4 o, float4 d, uint32_t path, uint32_t bounce) {
    const uint32_t slot = atomicAdd(&ctrl->scount, 1u);
    if (slot >= q.cap) return false;
    q.o[slot] = o; q.d[slot] = d; q.path[slot] = path; q.bounce[slot] = bounce;
    return true;
}
__global__ void k_slow_snapshot(DCtrl* ctrl, uint32_t k, uint32_t cap) {
    ctrl->slo[k] = k ? ctrl->shi[k - 1] : 0u;
    ctrl->shi[k] = min(ctrl->scount, cap);
}

// ---------------------------------------------------------------- ray for the hot loops
struct RayK {
    float ox, oy, oz, dx, dy, dz, ix, iy, iz;
    float ow, dw;  // lane 3 of orig / dir (normally +0)
    float qn, qd;  // lane-3 terms of norm.(incenter-orig) and norm.dir
};

__device__ inline RayK make_rayk(float4 o, float4 d) {
    RayK r;
    r.ox = o.x; r.oy = o.y; r.oz = o.z; r.ow = o.w;
    r.dx = d.x; r.dy = d.y; r.dz = d.z; r.dw = d.w;
    r.ix = 1.f / d.x; r.iy = 1.f / d.y; r.iz = 1.f / d.z;  // raytrace.rs:206-208
    r.qn = 0.f * (0.f - o.w);
    r.qd = 0.f * d.w;
    return r;
}

// BoundingBox::collides (raytrace.rs:860-907)
__device__ inline bool collides(float cx, float cy, float cz, float half, const RayK& r, float& tmin_o) {
    float tmin = -FLT_MAX, tmax = FLT_MAX;
    float a0 = (cx - r.ox) * r.ix, a1 = (cy - r.oy) * r.iy, a2 = (cz - r.oz) * r.iz;
    float b0 = r.ix * half, b1 = r.iy * half, b2 = r.iz * half;
    float t10 = a0 - b0, t20 = a0 + b0;
    float t11 = a1 - b1, t21 = a1 + b1;
    float t12 = a2 - b2, t22 = a2 + b2;
    if (r.dx != 0.f) {
        if (r.ix > 0.f) { tmin = t10; tmax = t20; } else { tmin = t20; tmax = t10; }
    }
    if (r.dy != 0.f) {
        if (r.iy > 0.f) { tmin = fmaxf(tmin, t11); tmax = fminf(tmax, t21); }
        else { tmin = fmaxf(tmin, t21); tmax = fminf(tmax, t11); }
    }
    if (r.dz != 0.f) {
        if (r.iz > 0.f) { tmin = fmaxf(tmin, t12); tmax = fminf(tmax, t22); }
        else { tmin = fmaxf(tmin, t22); tmax = fminf(tmax, t12); }
    }
    tmin_o = tmin;
    return tmin < tmax;
}

// Triangle::intersects (raytrace.rs:400-439).  Returns hit; face bits: 1 = back, 2 = edge.
template <bool COUNT>
__device__ inline bool tri_test(const DScene& sc, uint32_t tri, const RayK& r, float& t_o, uint32_t& face_o,
                                unsigned long long* cnt) {
    const float4 p0 = sc.tplane[2 * tri], p1 = sc.tplane[2 * tri + 1];
    float ax = p0.x - r.ox, ay = p0.y - r.oy, az = p0.z - r.oz;
    float num = (((0.f + p1.x * ax) + p1.y * ay) + p1.z * az) + r.qn;
    float den = (((0.f + p1.x * r.dx) + p1.y * r.dy) + p1.z * r.dz) + r.qd;
    float t = num / den;
    if (COUNT) cnt[1]++;
    if (t < 0.f) return false;
    float px = r.dx * t + r.ox, py = r.dy * t + r.oy, pz = r.dz * t + r.oz, pw = r.dw * t + r.ow;
    float ix = px - p0.x, iy = py - p0.y, iz = pz - p0.z;
    float l2 = ((ix * ix + iy * iy) + iz * iz) + pw * pw;
    if (l2 > p0.w) return false;
    if (COUNT) cnt[2]++;
    const float4 e0 = sc.tedge[4 * tri], e1 = sc.tedge[4 * tri + 1], e2 = sc.tedge[4 * tri + 2], e3 = sc.tedge[4 * tri + 3];
    float z = pw * 0.f;  // lane-3 product ip.w * side.w (side.w is +-0)
    float d0 = ((ix * e0.x + iy * e0.y) + iz * e0.z) + z;
    float d1 = ((ix * e1.x + iy * e1.y) + iz * e1.z) + z;
    float d2 = ((ix * e2.x + iy * e2.y) + iz * e2.z) + z;
    if (d0 > e0.w) return false;
    if (d1 > e1.w) return false;
    if (d2 > e2.w) return false;
    bool edge = (d0 > e3.x) | (d1 > e3.y) | (d2 > e3.z);
    t_o = t;
    face_o = (den > 0.f ? 1u : 0u) | (edge ? 2u : 0u);
    return true;
}

// ---------------------------------------------------------------- generic traversal (v1)
// Iterative form of BoundingBox::get_object_intersection_for_ray
// (raytrace.rs:909-1010).  One lane = one ray.  A frame is one inner box whose
// colliding children wait in sorted order:
//   w0 = index of its first child
//   w1 = order word: 3-bit child slots, next child in bits 0-2 | count << 24 |
//        has_hit << 28 | any_tmin_is_MAX << 29
//   w2 = t of the frame's best hit, w3 = triangle | face << 30
// Ancestors of the current frame live in LDS ([level][word][thread], so a
// lane always hits its own bank).  The children's tmin are not kept: the
// reference's skip rule `tmin < best_t` (raytrace.rs:965) is re-evaluated by
// running collides() again on the one child about to be visited, which
// returns the same float; since children are visited in ascending tmin and the
// best t never grows, the first skipped child ends the frame.
struct Frame { uint32_t first, w1; float t; uint32_t tri; };

#define F_COUNT(w) (((w) >> 24) & 15u)
#define F_HAS 0x10000000u
#define F_ANYMAX 0x20000000u

template <bool COUNT>
__device__ inline Frame expand(const DScene& sc, uint32_t first, uint32_t count, const RayK& r, unsigned long long* cnt) {
    float tm[8];
    bool anymax = false;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        tm[j] = INFINITY;
        if (j < (int)count) {
            const float4 g = *reinterpret_cast<const float4*>(&sc.nodes[first + j]);
            float tmin;
            if (COUNT) cnt[0]++;
            if (collides(g.x, g.y, g.z, g.w, r, tmin)) {
                tm[j] = tmin;
                anymax |= (tmin == FLT_MAX);
            }
        }
    }
    // stable ascending rank (insertion sort of raytrace.rs:941-947): child i
    // goes after every j < i with tm[j] <= tm[i] and every j > i with tm[j] < tm[i].
    uint32_t order = 0, nh = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t rank = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if (j < i) rank += (tm[j] <= tm[i]) ? 1u : 0u;
            if (j > i) rank += (tm[j] < tm[i]) ? 1u : 0u;
        }
        if (tm[i] != INFINITY) { order |= (uint32_t)i << (3u * rank); nh++; }
    }
    Frame f;
    f.first = first;
    f.w1 = order | (nh << 24) | (anymax ? F_ANYMAX : 0u);
    f.t = 0.f;
    f.tri = 0;
    return f;
}

// get_box_min_time_intersection (raytrace.rs:1012-1050)
template <bool COUNT>
__device__ inline bool leaf_scan(const DScene& sc, uint32_t first, uint32_t count, const RayK& r, float& t_o,
                                 uint32_t& tf_o, unsigned long long* cnt) {
    bool have = false;
    float bt = 0.f;
    uint32_t btf = 0;
    if (COUNT) cnt[4]++;
    for (uint32_t i = 0; i < count; i++) {
        uint32_t tri = sc.refs[first + i];
        float t; uint32_t face;
        if (tri_test<COUNT>(sc, tri, r, t, face, cnt)) {
            if (!have || t < bt) { bt = t; btf = tri | (face << 30); }
            have = true;
        }
    }
    t_o = bt; tf_o = btf;
    return have;
}

__device__ inline void merge(Frame& f, bool have, float t, uint32_t tf) {
    // fold step of raytrace.rs:949-1007: first hit is taken, later ones replace iff strictly closer
    if (have) {
        if (!(f.w1 & F_HAS) || t < f.t) { f.t = t; f.tri = tf; }
        f.w1 |= F_HAS;
    }
}

template <bool COUNT>
__device__ inline bool traverse(const DScene& sc, const RayK& r, uint32_t* lds, int nthreads, int tid, float& t_o,
                                uint32_t& tf_o, unsigned long long* cnt) {
    const DNode root = sc.nodes[0];
    if (root.is_leaf) return leaf_scan<COUNT>(sc, root.first, root.count, r, t_o, tf_o, cnt);
    if (COUNT) cnt[3]++;
    Frame cur = expand<COUNT>(sc, root.first, root.count, r, cnt);
    int sp = 0;
    for (;;) {
        if (F_COUNT(cur.w1) == 0) {
            if (sp == 0) break;
            const bool have = (cur.w1 & F_HAS) != 0;
            const float ct = cur.t;
            const uint32_t ctf = cur.tri;
            sp--;
            uint32_t* fr = lds + (size_t)sp * 4 * nthreads + tid;
            cur.first = fr[0];
            cur.w1 = fr[nthreads];
            cur.t = __uint_as_float(fr[2 * nthreads]);
            cur.tri = fr[3 * nthreads];
            merge(cur, have, ct, ctf);
            continue;
        }
        const uint32_t c = cur.w1 & 7u;
        cur.w1 = ((cur.w1 & 0x00FFFFFFu) >> 3) | ((cur.w1 & 0xFF000000u) - (1u << 24));
        const DNode ch = sc.nodes[cur.first + c];
        if (cur.w1 & F_HAS) {
            float tmin;
            collides(ch.cx, ch.cy, ch.cz, ch.half, r, tmin);
            if (!(tmin < cur.t)) { cur.w1 &= ~(15u << 24); continue; }  // raytrace.rs:965
        } else if (cur.w1 & F_ANYMAX) {
            float tmin;
            collides(ch.cx, ch.cy, ch.cz, ch.half, r, tmin);
            if (tmin == FLT_MAX) continue;  // raytrace.rs:986
        }
        if (ch.is_leaf) {
            float t; uint32_t tf;
            bool have = leaf_scan<COUNT>(sc, ch.first, ch.count, r, t, tf, cnt);
            merge(cur, have, t, tf);
        } else {
            uint32_t* fr = lds + (size_t)sp * 4 * nthreads + tid;
            fr[0] = cur.first;
            fr[nthreads] = cur.w1;
            fr[2 * nthreads] = __float_as_uint(cur.t);
            fr[3 * nthreads] = cur.tri;
            sp++;
            if (COUNT) cnt[3]++;
            cur = expand<COUNT>(sc, ch.first, ch.count, r, cnt);
        }
    }
    t_o = cur.t; tf_o = cur.tri;
    return (cur.w1 & F_HAS) != 0;
}

// Persistent closest-hit kernel: each wave pulls 64 queued rays at a time.
template <bool COUNT>
__global__ void __launch_bounds__(256) k_trace(DScene sc, const float4* __restrict__ qo, const float4* __restrict__ qd,
                                               DCtrl* __restrict__ ctrl, int pass, uint32_t* __restrict__ hit_tf,
                                               float* __restrict__ hit_t) {
    extern __shared__ uint32_t lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t count = ctrl->count[pass];
    uint32_t* head = &ctrl->head[pass];
    if (blockIdx.x == 0 && tid == 0) atomicAdd(&ctrl->rays, (unsigned long long)count);
    unsigned long long cnt[5] = {0, 0, 0, 0, 0};
    for (;;) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(head, 64u);
        base = __builtin_amdgcn_readfirstlane(base);
        if (base >= count) break;
        const uint32_t i = base + lane;
        if (i < count) {
            const RayK r = make_rayk(qo[i], qd[i]);
            float t = 0.f; uint32_t tf = 0;
            bool have = traverse<COUNT>(sc, r, lds, blockDim.x, tid, t, tf, cnt);
            hit_tf[i] = have ? tf : 0u;
            hit_t[i] = have ? t : 0.f;
        }
    }
    if (COUNT) {
#pragma unroll
        for (int k = 0; k < 5; k++)
            if (cnt[k]) atomicAdd(&ctrl->counters[k], cnt[k]);
    }
}

}  // namespace rtmi
#include "make_triangle.hpp"
#include "build_octree.hpp"
#include "shade.hpp"
#include "trace_oct.hpp"
#include "trace_pool.hpp"
#include "bvh_fast.hpp"
namespace rtmi {


// ---------------------------------------------------------------- linear-list closest hit (root box is a leaf)
// build_trivial_bounding_box (raytrace.rs:847-856): every ray scans the same list, so the list is
// streamed ONCE per 256 rays through LDS in chunks of RTMI_LIN_CHUNK triangles (coalesced float4 gathers
// by the whole block, double-buffered against the tests) and every lane reads the records at a
// wave-uniform LDS address (broadcast).  Fold and test exactly as get_box_min_time_intersection /
// Triangle::intersects (raytrace.rs:1012-1050, 400-439).
#define RTMI_LIN_CHUNK 64
template <bool COUNT>
__global__ void __launch_bounds__(256) k_trace_linear(DScene sc, const float4* __restrict__ qo, const float4* __restrict__ qd,
                                                      DCtrl* __restrict__ ctrl, int pass, uint32_t* __restrict__ hit_tf,
                                                      float* __restrict__ hit_t) {
    constexpr int C = RTMI_LIN_CHUNK;
    __shared__ float4 rec[2][6][C];   // [buffer][plane0, plane1, edge0..3][triangle]
    __shared__ uint32_t rid[2][C];
    __shared__ uint32_t s_base;
    const int tid = threadIdx.x;
    const uint32_t count = ctrl->count[pass];
    if (blockIdx.x == 0 && tid == 0) atomicAdd(&ctrl->rays, (unsigned long long)count);
    const DNode root = sc.nodes[0];
    const uint32_t first = root.first, ntri = root.count;
    const uint32_t nchunks = (ntri + C - 1) / C;
    unsigned long long cnt[5] = {0, 0, 0, 0, 0};
    // staging: slot s of a chunk = float4 number (s / C) of triangle (s % C); 6*C slots, 256 threads
    auto stage_load = [&](uint32_t chunk, float4 (&v)[2], uint32_t (&id)[2]) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint32_t s = tid + k * 256;
            v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            id[k] = 0;
            if (s < 6 * C) {
                const uint32_t j = chunk * C + (s % C), c = s / C;
                if (j < ntri) {
                    const uint32_t t = sc.refs[first + j];
                    id[k] = t;
                    v[k] = c < 2 ? sc.tplane[2 * t + c] : sc.tedge[4 * t + (c - 2)];
                }
            }
        }
    };
    auto stage_store = [&](int buf, const float4 (&v)[2], const uint32_t (&id)[2]) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const uint32_t s = tid + k * 256;
            if (s < 6 * C) {
                rec[buf][s / C][s % C] = v[k];
                if (s < C) rid[buf][s] = id[k];
            }
        }
    };
    for (;;) {
        __syncthreads();
        if (tid == 0) s_base = atomicAdd(&ctrl->head[pass], 256u);
        __syncthreads();
        const uint32_t base = s_base;
        if (base >= count) break;
        const uint32_t i = base + tid;
        const bool active = i < count;
        const RayK r = active ? make_rayk(qo[i], qd[i]) : make_rayk(make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 1.f, 0.f));
        bool lhave = false;
        float lt = 0.f;
        uint32_t ltf = 0;
        float4 v[2];
        uint32_t id[2];
        if (nchunks) { stage_load(0, v, id); stage_store(0, v, id); }
        __syncthreads();
        for (uint32_t ch = 0; ch < nchunks; ch++) {
            const int buf = ch & 1;
            const bool next = ch + 1 < nchunks;
            if (next) stage_load(ch + 1, v, id);  // global gathers in flight while this chunk is tested
            const uint32_t n = min((uint32_t)C, ntri - ch * C);
            for (uint32_t j = 0; j < n; j++) {
                const float4 p0 = rec[buf][0][j], p1 = rec[buf][1][j];
                const float ax = p0.x - r.ox, ay = p0.y - r.oy, az = p0.z - r.oz;
                const float num = (((0.f + p1.x * ax) + p1.y * ay) + p1.z * az) + r.qn;
                const float den = (((0.f + p1.x * r.dx) + p1.y * r.dy) + p1.z * r.dz) + r.qd;
                const float t = num / den;
                if (!(t < 0.f)) {
                    const float px = r.dx * t + r.ox, py = r.dy * t + r.oy, pz = r.dz * t + r.oz, pw = r.dw * t + r.ow;
                    const float ix = px - p0.x, iy = py - p0.y, iz = pz - p0.z;
                    const float l2 = ((ix * ix + iy * iy) + iz * iz) + pw * pw;
                    if (!(l2 > p0.w)) {
                        if (COUNT && active) cnt[2]++;
                        const float4 e0 = rec[buf][2][j], e1 = rec[buf][3][j], e2 = rec[buf][4][j], e3 = rec[buf][5][j];
                        const float z = pw * 0.f;
                        const float d0 = ((ix * e0.x + iy * e0.y) + iz * e0.z) + z;
                        const float d1 = ((ix * e1.x + iy * e1.y) + iz * e1.z) + z;
                        const float d2 = ((ix * e2.x + iy * e2.y) + iz * e2.z) + z;
                        const bool inside = !(d0 > e0.w) & !(d1 > e1.w) & !(d2 > e2.w);
                        const bool edge = (d0 > e3.x) | (d1 > e3.y) | (d2 > e3.z);
                        const uint32_t face = (den > 0.f ? 1u : 0u) | (edge ? 2u : 0u);
                        const bool take = inside & (!lhave | (t < lt));
                        lt = take ? t : lt;
                        ltf = take ? (rid[buf][j] | (face << 30)) : ltf;
                        lhave = lhave | inside;
                    }
                }
            }
            if (COUNT && active) cnt[1] += n;
            if (next) stage_store(buf ^ 1, v, id);
            __syncthreads();
        }
        if (active) {
            if (COUNT) cnt[4]++;
            hit_tf[i] = lhave ? ltf : 0u;
            hit_t[i] = lhave ? lt : 0.f;
        }
    }
    if (COUNT) {
#pragma unroll
        for (int k = 0; k < 5; k++)
            if (cnt[k]) atomicAdd(&ctrl->counters[k], cnt[k]);
    }
}

// ---------------------------------------------------------------- generation / shading (per-pass pipeline)
// pixel_ray / color_ray themselves are in shade.hpp (shared with the fused path kernels of trace_oct.hpp).
__global__ void __launch_bounds__(256) k_gen(DView v, uint64_t seed, uint32_t pix0, uint32_t npaths,
                                             float4* __restrict__ qo, float4* __restrict__ qd,
                                             uint32_t* __restrict__ qpath, DCtrl* __restrict__ ctrl) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t path = blockIdx.x * blockDim.x + threadIdx.x; path < npaths; path += stride) {
        uint32_t row, col, sample;
        path_pixel(v, pix0, path, row, col, sample);
        const uint32_t pixel = row * v.width + col;
        RayV r = pixel_ray(v, row, col, seed, pixel, sample);
        qo[path] = make_float4(r.orig.x, r.orig.y, r.orig.z, r.orig.w);
        qd[path] = make_float4(r.dir.x, r.dir.y, r.dir.z, r.dir.w);
        qpath[path] = path;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) ctrl->count[0] = npaths;
}

// color_ray + the tail of project_ray for every ray of pass `pass`.
__global__ void __launch_bounds__(256) k_shade(DScene sc, DView v, uint64_t seed, uint32_t pix0, uint32_t npaths, int pass,
                                               const float4* __restrict__ qo, const float4* __restrict__ qd,
                                               const uint32_t* __restrict__ qpath, const uint32_t* __restrict__ hit_tf,
                                               const float* __restrict__ hit_t, float4* __restrict__ qo_n,
                                               float4* __restrict__ qd_n, uint32_t* __restrict__ qpath_n,
                                               uint16_t* __restrict__ mstack, float4* __restrict__ scol,
                                               DCtrl* __restrict__ ctrl, SlowQ slow) {
    __shared__ uint32_t s_cnt[4], s_base;
    const uint32_t count = ctrl->count[pass];
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    // round the loop bound up so that whole BLOCKS stay converged for the ballot and the block-wide queue reservation
    const uint32_t bound = (count + 255u) & ~255u;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < bound; i += stride) {
        bool push = false;
        RayV nr;
        uint32_t path = 0;
        if (i < count) {
            // a miss (sky) or an edge face ends the path without looking at the ray: 2/3 of the primary rays never touch
            // their 36 bytes of origin / direction / hit time.  What a hit needs is requested together: the loads do not
            // depend on each other.
            path = qpath[i];
            const uint32_t tf = hit_tf[i];
            float t = 0.f;
            float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f), d4 = make_float4(0.f, 0.f, 1.f, 0.f);
            if ((tf & 0x3FFFFFFFu) != 0u && !((tf >> 30) & 2u)) { t = hit_t[i]; o4 = qo[i]; d4 = qd[i]; }
            uint32_t prow, pcol, sample;
            path_pixel(v, pix0, path, prow, pcol, sample);
            push = shade_hit(sc, v.maxdepth, seed, npaths, path, prow * v.width + pcol, sample, (uint32_t)pass, tf, t,
                             V4{o4.x, o4.y, o4.z, o4.w}, V4{d4.x, d4.y, d4.z, d4.w}, mstack, scol, nr);
            // a bounce ray with an exactly-zero direction component goes to the slow path (SlowQ), not to the next pass
            if (push && slow.cap && has_zero_component(nr.dir.x, nr.dir.y, nr.dir.z) &&
                slow_push(slow, ctrl, make_float4(nr.orig.x, nr.orig.y, nr.orig.z, nr.orig.w), make_float4(nr.dir.x, nr.dir.y, nr.dir.z, nr.dir.w),
                          path, (uint32_t)pass + 1u))
                push = false;
        }
        // compact surviving rays into the next queue: ballot + prefix sum per wave, ONE atomic per block of four waves.  (One
        // per wave made the queue counter the kernel's bottleneck: 1.7 M same-address atomics per frame at ~4.8 ns each were
        // half of its 16.7 ms.)
        const unsigned long long mask = __ballot(push);
        if (lane == 0) s_cnt[wv] = (uint32_t)__popcll(mask);
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
            s_base = total ? atomicAdd(&ctrl->count[pass + 1], total) : 0u;
        }
        __syncthreads();
        {
            uint32_t base = s_base;
            for (uint32_t k = 0; k < wv; k++) base += s_cnt[k];
            if (push) {
                const uint32_t slot = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                store_stream(&qo_n[slot], make_float4(nr.orig.x, nr.orig.y, nr.orig.z, nr.orig.w));
                store_stream(&qd_n[slot], make_float4(nr.dir.x, nr.dir.y, nr.dir.z, nr.dir.w));
                store_stream(&qpath_n[slot], path);
            }
        }
        __syncthreads();  // s_cnt / s_base are rewritten by the next iteration
    }
}

// walk_ray_set's per-pixel accumulation (raytrace.rs:1414-1426): acc = 0; acc += sample_i in sample order; * (1/spp).
// The sample colours of a pixel are consecutive in `scol` ([pixel][sample]), so one thread per pixel would read 16 B at a
// stride of spp * 16 B (round 2: 5.9 x the bytes fetched, 128-B lines evicted between a thread's iterations).  Instead a
// block takes RTMI_ACC_PIX consecutive pixels, streams their samples through LDS in chunks with fully coalesced float4
// loads, and thread (pixel j, channel c) adds its pixel's samples of the chunk in order, keeping the running sum in a
// register across chunks -- the same sequence of f32 additions per channel as the reference's Vec3 adds.  One float4 of
// padding per 64 keeps the 16 pixels that are summed at a time (spp = 64) on different LDS banks.
// `out` is the caller's tile buffer; the sub-tile `sub` of `nsub` holds every nsub-th ROW of that tile, so local row lr of
// the sub-tile is row lr * nsub + sub of the buffer.
#define RTMI_ACC_PIX 64
#define RTMI_ACC_CHUNK 1024
__global__ void __launch_bounds__(256) k_accum(uint32_t npixels, uint32_t spp, const float4* __restrict__ scol,
                                               float* __restrict__ out, uint32_t pix0, uint32_t W, uint32_t nsub, uint32_t sub,
                                               FastDiv dW) {
    __shared__ float4 stage[RTMI_ACC_CHUNK + RTMI_ACC_CHUNK / 64 + 1];
    const float* stage_f = reinterpret_cast<const float*>(stage);
    const uint32_t tid = threadIdx.x, j = tid >> 2, c = tid & 3u;
    const float inv = 1.f / (float)spp;
    for (uint32_t pb = blockIdx.x * RTMI_ACC_PIX; pb < npixels; pb += gridDim.x * RTMI_ACC_PIX) {
        const uint32_t npb = min((uint32_t)RTMI_ACC_PIX, npixels - pb);
        const uint32_t f0 = pb * spp, f1 = f0 + npb * spp;       // the block's samples: scol[f0 .. f1)
        const uint32_t my0 = f0 + j * spp, my1 = my0 + spp;      // this thread's pixel (when j < npb)
        float acc = 0.f;
        for (uint32_t ch = f0; ch < f1; ch += RTMI_ACC_CHUNK) {
            const uint32_t n = min((uint32_t)RTMI_ACC_CHUNK, f1 - ch);
            __syncthreads();  // the previous chunk has been consumed
            for (uint32_t k = tid; k < n; k += 256u) stage[k + (k >> 6)] = scol[ch + k];
            __syncthreads();
            if (j < npb) {
                const uint32_t lo = max(my0, ch), hi = min(my1, ch + n);
                for (uint32_t s = lo; s < hi; s++) {
                    const uint32_t k = s - ch;
                    acc = acc + stage_f[(k + (k >> 6)) * 4u + c];
                }
            }
        }
        if (j < npb) {
            const uint32_t lp = pix0 + pb + j, lr = fdiv(lp, dW), col = lp - lr * W;
            const size_t orow = (size_t)lr * nsub + sub;
            out[(orow * W + col) * 4u + c] = acc * inv;
        }
    }
}

// write_png's quantisation (raytrace.rs:1468-1473): `as u8` truncates and saturates, NaN -> 0
__global__ void __launch_bounds__(256) k_quantize(uint64_t npixels, const float4* __restrict__ in, uint8_t* __restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npixels; p += stride) {
        const float4 c = in[p];
        const float ch[3] = {c.x * 255.f, c.y * 255.f, c.z * 255.f};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float x = ch[k];
            out[p * 3 + k] = (x != x) ? 0 : (x <= 0.f ? 0 : (x >= 255.f ? 255 : (uint8_t)x));
        }
    }
}

// rtmi_render_frame_multi: bands of the n scenes, each padded to `mr` rows, lie one after the other in `stage`;
// image row `row` is local row (row / (S*n)) * S + row % S of band (row / S) % n.  `px` = bytes per pixel (16 or 3).
__global__ void __launch_bounds__(256) k_deinterleave(const uint8_t* __restrict__ stage, uint8_t* __restrict__ out, uint32_t W,
                                                      uint32_t H, uint32_t S, uint32_t n, uint32_t mr, uint32_t px) {
    const uint64_t npix = (uint64_t)W * H, stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += stride) {
        const uint32_t row = (uint32_t)(p / W), col = (uint32_t)(p - (uint64_t)row * W);
        const uint32_t band = (row / S) % n, lr = (row / (S * n)) * S + row % S;
        const uint64_t src = (((uint64_t)band * mr + lr) * W + col) * px;
        if (px == 16) *reinterpret_cast<float4*>(out + p * 16) = *reinterpret_cast<const float4*>(stage + src);
        else { out[p * 3] = stage[src]; out[p * 3 + 1] = stage[src + 1]; out[p * 3 + 2] = stage[src + 2]; }
    }
}

// Analytic spheres (rtmi_sphere_t; a build-defined extension, see include/rtmi.h): every ray of the pass against the
// scene's flat sphere list, after the tree's closest hit.  4-lane Vec3 arithmetic in the operation order DESIGN.md 4.6 states.
__global__ void __launch_bounds__(256) k_trace_spheres(DScene sc, const float4* __restrict__ qo, const float4* __restrict__ qd,
                                                       const DCtrl* __restrict__ ctrl, int pass, uint32_t* __restrict__ hit_tf,
                                                       float* __restrict__ hit_t) {
    const uint32_t count = ctrl->count[pass];
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const float4 o4 = qo[i], d4 = qd[i];
        const V4 ro{o4.x, o4.y, o4.z, o4.w}, rd{d4.x, d4.y, d4.z, d4.w};
        uint32_t tf = hit_tf[i];
        float bt = hit_t[i];
        bool have = tf != 0u;
        for (uint32_t k = 0; k < sc.nspheres; k++) {
            const float4 s0 = sc.spheres[2 * k];
            const V4 oc = vsub(ro, mk(s0.x, s0.y, s0.z));
            const float b = vdot(oc, rd);
            const float c = vlen2(oc) - s0.w * s0.w;
            const float disc = b * b - c;
            if (!(disc >= 0.f)) continue;
            const float sq = sqrtf(disc);
            const float t0 = (0.f - b) - sq, t1 = (0.f - b) + sq;
            float t; uint32_t face;
            if (t0 >= 0.f) { t = t0; face = 0u; }
            else if (t1 >= 0.f) { t = t1; face = 1u; }
            else continue;
            if (!have || t < bt) { bt = t; tf = (sc.ntris + k) | (face << 30); have = true; }
        }
        hit_tf[i] = tf;
        hit_t[i] = bt;
    }
}

// Explicit-ray entry (rtmi_trace): queue = the caller's rays
__global__ void k_set_count(DCtrl* ctrl, uint32_t n) { ctrl->count[0] = n; }

// ---------------------------------------------------------------- host side of the ABI
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }

// HIP status -> ABI status: out of memory, "no device visible" and every other runtime failure are told apart
static int hip_code(hipError_t e) {
    // the failure is reported through the ABI's own channel (status + rtmi_last_error); what the runtime keeps pending for
    // the thread's next hipGetLastError() is dropped, so that the CALLER's next HIP call (PyTorch polls after every
    // operation) does not inherit this library's error
    (void)hipGetLastError();
    if (e == hipErrorOutOfMemory) return RTMI_ERR_OOM;
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return RTMI_ERR_NO_DEVICE;
    return RTMI_ERR_DEVICE;
}

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(hip_code(e_), std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// No C++ exception crosses the ABI: bodies that allocate host memory run inside this guard.
#define RTMI_GUARD_BEGIN try {
#define RTMI_GUARD_END                                                                                 \
    } catch (const std::bad_alloc&) { return fail(RTMI_ERR_OOM, "host allocation failed");             \
    } catch (const std::exception& ex_) { return fail(RTMI_ERR_INVALID, std::string("internal error: ") + ex_.what()); \
    } catch (...) { return fail(RTMI_ERR_INVALID, "internal error"); }

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t ensure(size_t want) {
        if (want <= n) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
        hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
        if (e == hipSuccess) n = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

}  // namespace rtmi

using namespace rtmi;

// Per-stream workspace of one batch (queues, hit records, surface stacks, sample colours, control block).
struct Work {
    size_t cap = 0;
    uint32_t cap_depth = 0;
    bool full = false;  // both queues + hit records allocated (per-pass pipeline, rtmi_trace)
    DevBuf<float4> qo[2], qd[2], scol;
    DevBuf<uint32_t> qpath[2], hit_tf;
    DevBuf<float> hit_t;
    DevBuf<uint16_t> mstack;
    DevBuf<DCtrl> ctrl;
    // slow path (SlowQ): its queue, the side stream its consumer launches run on, "producer k done" / "slow path done" events
    DevBuf<float4> sqo, sqd;
    DevBuf<uint32_t> sqpath, sqbounce;
    hipStream_t sstream = nullptr;
    std::vector<hipEvent_t> sev;
    hipEvent_t sdone = nullptr, sgo = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> pass_ev;  // start/stop of the trace kernel of every pass
    void release() {
        for (int k = 0; k < 2; k++) { qo[k].release(); qd[k].release(); qpath[k].release(); }
        scol.release(); hit_tf.release(); hit_t.release(); mstack.release(); ctrl.release();
        for (int k = 0; k < 2; k++) if (ev[k]) { (void)hipEventDestroy(ev[k]); ev[k] = nullptr; }
        for (hipEvent_t e : pass_ev) (void)hipEventDestroy(e);
        pass_ev.clear();
        sqo.release(); sqd.release(); sqpath.release(); sqbounce.release();
        for (hipEvent_t e : sev) (void)hipEventDestroy(e);
        sev.clear();
        if (sdone) { (void)hipEventDestroy(sdone); sdone = nullptr; }
        if (sgo) { (void)hipEventDestroy(sgo); sgo = nullptr; }
        if (sstream) { (void)hipStreamDestroy(sstream); sstream = nullptr; }
        cap = 0; cap_depth = 0; full = false;
    }
};

struct rtmi_scene {
    int device = 0;
    uint32_t options = 0;
    DScene d{};
    DevBuf<DNode> nodes;
    DevBuf<uint32_t> refs;
    DevBuf<float4> tplane, tedge, mats, spheres;
    std::vector<float4> hmats;  // the triangles' surface table (host copy): sphere surfaces are appended to it
    std::vector<rtmi_triangle_t> htris;  // the records (host copy): the fast-mode BVH is rebuilt from them when corners arrive
    DevBuf<uint4> fnodes, oblocks;
    DevBuf<uint32_t> wlinks;
    // RTMI_OPT_BVH: SAH BVH over the triangles' bounding spheres (bvh_fast.hpp)
    DevBuf<float4> bnodes;
    DevBuf<uint4> bleaves;
    uint32_t bvh_root = 0;
    size_t bvh_lds = 0;
    int bvh_blocks_per_cu = 8;
    bool bvh_ok = false;
    bool root_is_leaf = false; // build_trivial_bounding_box: one list for every ray -> k_trace_linear
    bool octree = false;       // the tree passed the exact-octree check
    std::string why_generic;   // reason when it did not
    int oct_blocks_per_cu = 8;   // what fits (occupancy query)
    uint32_t active_streams = 1; // sub-tiles of the render call in flight: their persistent kernels share the CUs
    size_t oct_lds = 0;
    // k_trace_pool: rays per wave, slot stride (words), LDS bytes per wave, waves per CU; pool_P == 0: not available
    uint32_t pool_P = 0, pool_stride = 0;
    size_t pool_lds = 0;
    int pool_blocks_per_cu = 0;
    // A tile is rendered as up to two interleaved sub-tiles, each with its own workspace on its own internal
    // stream, so that the small deep bounce passes of one overlap the bulk of the other.
    Work w[RTMI_MAX_STREAMS];
    hipStream_t istream[RTMI_MAX_STREAMS] = {};
    hipEvent_t fork_ev = nullptr, end_ev = nullptr, join_ev[RTMI_MAX_STREAMS] = {};
    DevBuf<float4> tile;
    DevBuf<uint8_t> qbytes;
    DevBuf<uint8_t> mstage, mframe;  // rtmi_render_frame_multi, root scene: received bands / the frame
    hipStream_t mstream = nullptr;   // rtmi_render_frame_multi: this scene's band stream
    std::vector<ncclComm_t> comms;   // root scene, RTMI_FRAME_RCCL: one communicator per scene of the last device list
    std::vector<int> comm_devices;
    int peer_root = -1;              // root device the peer-access state below refers to (-1: not asked yet)
    int peer_ok = 0;                 // 1 = this device reaches peer_root directly (enabled once), 0 = the runtime refused
    std::string peer_msg;            // the runtime's reason when it refused
    int num_cu = 256;
    int trace_block = 256;
    size_t trace_lds = 0;
    // tuning knobs, read from the environment ONCE at scene creation (getenv is not free and not thread-safe
    // against setenv): waves per CU of the octree kernel, refill thresholds, XCD mode, streams, batch size
    rtmi_tuning_t tune{};
    bool verbose = false;
    int vote[4] = {3, 2, 3, 2};  // SELECT : LEAF vote weights of the walk, primary rays / bounce rays (experiments: RTMI_VOTE="a,b,c,d")
    unsigned long long vprev[RTMI_MAX_STREAMS][13] = {};  // verbose per-pass deltas (per handle: no shared statics)
};

// RCCL for RTMI_FRAME_RCCL, loaded on first use: no link-time dependency, and in a process that already holds PyTorch's
// RCCL (same SONAME librccl.so.1) dlopen hands back that copy instead of mapping a second one.
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGather) Gather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};
static Rccl* rccl_api() {
    static Rccl api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.lib) break;
        }
        if (!api.lib) return;
        api.CommInitAll = (decltype(api.CommInitAll))dlsym(api.lib, "ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
        api.GroupStart = (decltype(api.GroupStart))dlsym(api.lib, "ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))dlsym(api.lib, "ncclGroupEnd");
        api.Gather = (decltype(api.Gather))dlsym(api.lib, "ncclGather");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
        api.ok = api.CommInitAll && api.CommDestroy && api.GroupStart && api.GroupEnd && api.Gather && api.GetErrorString;
    });
    return api.ok ? &api : nullptr;
}

// (Re)build the fast-mode BVH (bvh_fast.hpp) from the scene's triangle records and, when given, their corners.
static int upload_bvh(rtmi_scene* s, const float* corners9) {
    s->bvh_ok = false;
    BvhBuild bb;
    if (s->htris.size() >= (1u << 26) || !bvh_build(s->htris.data(), s->htris.size(), bb, corners9)) return RTMI_OK;  // no BVH: RTMI_OPT_BVH is then ignored
    if (bb.wide.empty()) bb.wide.resize(8, make_float4(0.f, 0.f, 0.f, 0.f));
    auto up = [&](auto& buf, const auto& host) -> hipError_t {
        hipError_t e = buf.ensure(std::max<size_t>(host.size(), 1));
        if (e != hipSuccess || host.empty()) return e;
        return hipMemcpy(buf.p, host.data(), host.size() * sizeof(host[0]), hipMemcpyHostToDevice);
    };
    hipError_t be = up(s->bnodes, bb.wide);
    if (be == hipSuccess) be = up(s->bleaves, bb.leaves);
    if (be != hipSuccess) return fail(hip_code(be), std::string("scene upload (BVH): ") + hipGetErrorString(be));
    s->bvh_root = bb.root_link;
    s->bvh_lds = (size_t)(3 * bb.depth + 2) * 64 * 4;  // an INNER step leaves at most 3 links waiting per level
    int nb = 0;
    if (s->bvh_lds <= 64 * 1024 &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_bvh<false>, 64, s->bvh_lds) == hipSuccess && nb > 0) {
        s->bvh_blocks_per_cu = nb;
        s->bvh_ok = true;
    } else (void)hipGetLastError();
    return RTMI_OK;
}

static size_t env_size(const char* name, size_t dflt) {
    const char* s = getenv(name);
    if (!s || !*s) return dflt;
    char* end = nullptr;
    unsigned long long v = strtoull(s, &end, 10);
    return (end && *end == '\0' && (v > 0 || dflt == 0)) ? (size_t)v : dflt;
}

extern "C" {

const char* rtmi_last_error(void) { return g_err.c_str(); }

int rtmi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int rtmi_scene_create(const rtmi_triangle_t* tris, uint64_t ntris, const rtmi_box_t* boxes, uint64_t nboxes,
                      const uint32_t* tri_refs, uint64_t nrefs, int device, rtmi_scene_t** out) {
    if (!out) return fail(RTMI_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!tris || ntris < 1) return fail(RTMI_ERR_INVALID, "need at least the sentinel triangle (index 0)");
    if (!boxes || nboxes < 1) return fail(RTMI_ERR_INVALID, "need at least the root box");
    if (nrefs > 0 && !tri_refs) return fail(RTMI_ERR_INVALID, "tri_refs is NULL");
    if (ntris >= (1ull << 30)) return fail(RTMI_ERR_UNSUPPORTED, "more than 2^30 triangles");
    if (nboxes >= (1ull << 32) || nrefs >= (1ull << 32)) return fail(RTMI_ERR_UNSUPPORTED, "tree too large for 32-bit indices");

    RTMI_GUARD_BEGIN
    // ---- validate the tree, compute inner depth (levels of the LDS stack)
    std::vector<uint32_t> depth(nboxes, 0xFFFFFFFFu);
    depth[0] = 0;
    uint32_t max_inner_depth = 0;
    for (uint64_t i = 0; i < nboxes; i++) {
        const rtmi_box_t& b = boxes[i];
        if (depth[i] == 0xFFFFFFFFu) return fail(RTMI_ERR_INVALID, "box " + std::to_string(i) + " is not reachable from an earlier box");
        if (b.is_leaf) {
            if ((uint64_t)b.first + b.count > nrefs) return fail(RTMI_ERR_INVALID, "leaf triangle range out of bounds");
            for (uint32_t k = 0; k < b.count; k++)
                if (tri_refs[b.first + k] >= ntris) return fail(RTMI_ERR_INVALID, "triangle index out of range");
        } else {
            // the reference's boxmap has 8 slots (raytrace.rs:929): more children would panic there
            if (b.count < 1 || b.count > 8) return fail(RTMI_ERR_INVALID, "inner box must have 1..8 children");
            if (b.first <= i || (uint64_t)b.first + b.count > nboxes) return fail(RTMI_ERR_INVALID, "child range must follow its parent");
            for (uint32_t k = 0; k < b.count; k++) {
                if (depth[b.first + k] != 0xFFFFFFFFu) return fail(RTMI_ERR_INVALID, "box has two parents");
                depth[b.first + k] = depth[i] + 1;
            }
            max_inner_depth = std::max(max_inner_depth, depth[i]);
        }
    }
    const uint32_t levels = std::max<uint32_t>(1u, max_inner_depth);
    int block = 256;
    if ((size_t)levels * 16 * 256 > 40 * 1024) block = 64;
    if ((size_t)levels * 16 * block > 64 * 1024) return fail(RTMI_ERR_UNSUPPORTED, "octree deeper than the LDS stack allows");

    // ---- device records
    std::vector<DNode> hn(nboxes);
    for (uint64_t i = 0; i < nboxes; i++) {
        const rtmi_box_t& b = boxes[i];
        hn[i] = DNode{b.orig[0], b.orig[1], b.orig[2], b.len2, b.first, b.count, b.is_leaf ? 1u : 0u, 0u};
    }
    // ---- exact-octree form: every child must be the builder's octant of its parent, bit for bit
    //      (orig + (+-newlen2), newlen2 = len2 / 2, raytrace.rs:816-824), stored in octant order.
    //      Inner boxes get a 64-B record (centre, child mask, 8 child links); leaves only their reference blocks.
    std::vector<uint4> hfn;
    std::vector<uint4> hob;
    std::vector<uint32_t> hwl;  // explicit block indices of FN_WIDE boxes
    std::string why;
    {
        auto fb = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
        bool ok = !boxes[0].is_leaf;  // a one-leaf tree is the linear list (k_trace_linear)
        if (!ok) why = "the root box is a leaf";
        // record index of every inner box (in box order, root = 0) / first reference block of every leaf
        std::vector<uint32_t> slot(nboxes, 0);
        uint64_t ninner = 0;
        const float root_len2 = boxes[0].len2;
        for (uint64_t i = 0; i < nboxes && ok; i++) {
            const rtmi_box_t& b = boxes[i];
            if (fb(b.len2) != fb(ldexpf(root_len2, -(int)depth[i])) || !(b.len2 > 0.f) || !std::isfinite(b.len2)) { ok = false; why = "box half-length is not root/2^depth"; break; }
            if (b.is_leaf) {
                for (uint32_t k = 0; k < b.count; k++)
                    if (tri_refs[b.first + k] == 0) { ok = false; why = "a leaf lists the sentinel triangle 0"; }
                if (!ok) break;
                if (hob.size() >= (1ull << 31)) { ok = false; why = "more than 2^31 reference blocks"; break; }
                slot[i] = (uint32_t)hob.size() | 0x80000000u;
                // blocks of 4 indices; the list ends at the first 0, or after a block whose 4th index carries
                // bit 31 (a full last block: no extra all-zero block, triangle indices are < 2^30)
                for (uint32_t k = 0; k < std::max<uint32_t>(b.count, 1u); k += 4) {
                    uint32_t v[4] = {0, 0, 0, 0};
                    for (uint32_t j = 0; j < 4; j++)
                        if (k + j < b.count) v[j] = tri_refs[b.first + k + j];
                    if (k + 4 == b.count) v[3] |= 0x80000000u;
                    hob.push_back(make_uint4(v[0], v[1], v[2], v[3]));
                }
            } else {
                slot[i] = (uint32_t)ninner++;
            }
        }
        if (ok) {
            hfn.assign(2 * ninner, make_uint4(0, 0, 0, 0));
            for (uint64_t i = 0; i < nboxes && ok; i++) {
                const rtmi_box_t& b = boxes[i];
                if (b.is_leaf) continue;
                const float h = b.len2 / 2.f;
                uint32_t mask = 0, leafmask = 0, first_block[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                uint32_t base_inner = 0, base_block = 0;
                bool have_inner = false, have_leaf = false;
                int prev = -1;
                for (uint32_t k = 0; k < b.count && ok; k++) {
                    const rtmi_box_t& c = boxes[b.first + k];
                    int oct = 0;
                    for (int a = 0; a < 3; a++) {
                        const float lo = b.orig[a] + (-1.f * h), hi = b.orig[a] + h;
                        if (fb(lo) == fb(hi)) { ok = false; why = "degenerate box"; break; }
                        if (fb(c.orig[a]) == fb(hi)) oct |= 1 << a;
                        else if (fb(c.orig[a]) != fb(lo)) { ok = false; why = "child centre is not an octant centre of its parent"; break; }
                    }
                    if (ok && oct <= prev) { ok = false; why = "children are not in octant order"; }
                    prev = oct;
                    mask |= 1u << oct;
                    // children follow each other in box order, so the inner ones have consecutive records and the leaf
                    // ones consecutive runs of blocks: one base each + (leaves) a byte offset per octant
                    if (c.is_leaf) {
                        leafmask |= 1u << oct;
                        first_block[oct & 7] = slot[b.first + k] & 0x7FFFFFFFu;
                        if (!have_leaf) { base_block = first_block[oct & 7]; have_leaf = true; }
                    } else if (!have_inner) { base_inner = slot[b.first + k]; have_inner = true; }
                }
                uint32_t w = mask | (leafmask << 8), offlo = 0, offhi = 0, q1y = base_block;
                bool wide = false;
                for (int o = 0; o < 8; o++)
                    if ((leafmask >> o) & 1u) wide |= first_block[o] - base_block > 255u;
                if (wide) {
                    w |= RTMI_FN_WIDE;
                    q1y = (uint32_t)(hwl.size() / 8);
                    hwl.insert(hwl.end(), first_block, first_block + 8);
                } else {
                    for (int o = 0; o < 8; o++) {
                        const uint32_t off = ((leafmask >> o) & 1u) ? first_block[o] - base_block : 0u;
                        if (o < 4) offlo |= off << (8 * o); else offhi |= off << (8 * (o - 4));
                    }
                }
                uint4* rec = &hfn[2 * (size_t)slot[i]];
                rec[0] = make_uint4(fb(b.orig[0]), fb(b.orig[1]), fb(b.orig[2]), w);
                rec[1] = make_uint4(base_inner, q1y, offlo, offhi);
            }
        }
        // k_trace_oct addresses its records with 32-bit byte offsets (ld_off32): 32 B per inner box and per triangle plane
        // record, 64 B per triangle edge record, 16 B per reference block; its stack keeps an inner box's record index in 22 bits
        if (ok && (ninner >= (1ull << 22) || ntris >= (1ull << 26) || hob.size() >= (1ull << 28))) { ok = false; why = "more than 2^22 inner boxes, or an array of the octree form would exceed 4 GiB"; }
        if (!ok) { hfn.clear(); hob.clear(); hwl.clear(); }
    }

    std::map<std::tuple<uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t>, uint32_t> matmap;
    std::vector<float4> hm, hp(2 * ntris), he(4 * ntris);
    auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
    auto fbits = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
    for (uint64_t i = 0; i < ntris; i++) {
        const rtmi_triangle_t& t = tris[i];
        if (t.surface_kind > RTMI_REFLECTIVE) return fail(RTMI_ERR_INVALID, "unknown surface kind");
        // fields a kind does not carry do not distinguish materials
        const float alpha = t.surface_kind == RTMI_SOLID ? 0.f : t.alpha;
        const float scat = t.surface_kind == RTMI_REFLECTIVE ? t.scattering : 0.f;
        auto key = std::make_tuple(t.surface_kind, bits(t.color[0]), bits(t.color[1]), bits(t.color[2]), bits(alpha), bits(scat));
        auto it = matmap.find(key);
        uint32_t mid;
        if (it == matmap.end()) {
            mid = (uint32_t)matmap.size();
            matmap.emplace(key, mid);
            hm.push_back(make_float4(t.color[0], t.color[1], t.color[2], alpha));
            hm.push_back(make_float4(scat, fbits(t.surface_kind), 0.f, 0.f));
        } else mid = it->second;
        hp[2 * i] = make_float4(t.incenter[0], t.incenter[1], t.incenter[2], t.bounding_r2);
        hp[2 * i + 1] = make_float4(t.norm[0], t.norm[1], t.norm[2], fbits(mid));
        const float om = 1.f - t.edge_thickness;  // raytrace.rs:419
        for (int k = 0; k < 3; k++) he[4 * i + k] = make_float4(t.sides[k][0], t.sides[k][1], t.sides[k][2], t.side_lens[k]);
        he[4 * i + 3] = make_float4(t.side_lens[0] * om, t.side_lens[1] * om, t.side_lens[2] * om, 0.f);
    }
    if (matmap.size() > 65535) return fail(RTMI_ERR_UNSUPPORTED, "more than 65535 distinct surfaces");

    int ndev = rtmi_device_count();
    if (ndev <= 0) return fail(RTMI_ERR_NO_DEVICE, "no HIP device visible: the MI355X kernels cannot run (there is no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(RTMI_ERR_INVALID, "device index out of range");
    HIPCHK(hipSetDevice(device));

    std::vector<uint32_t> hrefs(tri_refs, tri_refs + nrefs);
    struct Owner {  // destroys a half-built scene on any early exit, exceptions included
        rtmi_scene* s;
        ~Owner() { if (s) rtmi_scene_destroy(s); }
    } own{new rtmi_scene()};
    rtmi_scene* s = own.s;
    s->device = device;
    s->tune.batch_paths = env_size("RTMI_BATCH_PATHS", (size_t)256 << 20);
    s->tune.streams = (uint32_t)std::min<size_t>(env_size("RTMI_STREAMS", 0), RTMI_MAX_STREAMS);
    s->tune.subtile_min_paths = (uint32_t)std::min<size_t>(env_size("RTMI_SUBTILE_MIN_PATHS", 32768), 0xFFFFFFFFu);
    s->tune.oct_waves_per_cu = (uint32_t)std::min<size_t>(env_size("RTMI_OCT_WAVES_PER_CU", 0), 32);
    s->tune.refill_min0 = (uint32_t)std::min<size_t>(env_size("RTMI_REFILL_MIN0", 64), 64);
    s->tune.refill_min = (uint32_t)std::min<size_t>(env_size("RTMI_REFILL_MIN", 16), 64);
    s->tune.xcd_aware = (uint32_t)(env_size("RTMI_XCD_AWARE", 0) % 3);
    s->tune.kernel = (uint32_t)std::min<size_t>(env_size("RTMI_KERNEL", 0), 2);
    s->tune.pipeline = (uint32_t)std::min<size_t>(env_size("RTMI_PIPELINE", 0), 3);
    s->tune.slow_path_off = (uint32_t)std::min<size_t>(env_size("RTMI_SLOW_PATH_OFF", 0), 1);
    s->verbose = getenv("RTMI_VERBOSE") != nullptr;
    if (const char* v = getenv("RTMI_VOTE")) {
        int q[4];
        if (sscanf(v, "%d,%d,%d,%d", &q[0], &q[1], &q[2], &q[3]) == 4 && q[0] > 0 && q[1] > 0 && q[2] > 0 && q[3] > 0) memcpy(s->vote, q, sizeof q);
    }
    s->trace_block = block;
    s->trace_lds = (size_t)levels * 16 * block;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) s->num_cu = prop.multiProcessorCount;
    auto up = [&](auto& buf, const auto& host) -> hipError_t {
        hipError_t e = buf.ensure(std::max<size_t>(host.size(), 1));
        if (e != hipSuccess) return e;
        if (host.empty()) return hipSuccess;
        return hipMemcpy(buf.p, host.data(), host.size() * sizeof(host[0]), hipMemcpyHostToDevice);
    };
    hipError_t e = up(s->nodes, hn);
    if (e == hipSuccess) e = up(s->refs, hrefs);
    if (e == hipSuccess) e = up(s->tplane, hp);
    if (e == hipSuccess) e = up(s->tedge, he);
    if (e == hipSuccess) e = up(s->mats, hm);
    s->octree = !hfn.empty();
    s->root_is_leaf = boxes[0].is_leaf != 0;
    s->why_generic = why;
    if (e == hipSuccess && s->octree) e = up(s->fnodes, hfn);
    if (e == hipSuccess && s->octree) e = up(s->oblocks, hob);
    if (e == hipSuccess && s->octree) e = up(s->wlinks, hwl);
    for (int k = 0; k < RTMI_MAX_STREAMS && e == hipSuccess; k++) {
        e = s->w[k].ctrl.ensure(1);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->istream[k], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreate(&s->w[k].ev[0]);
        if (e == hipSuccess) e = hipEventCreate(&s->w[k].ev[1]);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&s->join_ev[k], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipEventCreate(&s->fork_ev);
    if (e == hipSuccess) e = hipEventCreate(&s->end_ev);
    if (e != hipSuccess) return fail(hip_code(e), std::string("scene upload: ") + hipGetErrorString(e));
    s->d = DScene{s->nodes.p, s->refs.p, s->tplane.p, s->tedge.p, s->mats.p,
                  (uint32_t)nboxes, (uint32_t)ntris, (uint32_t)matmap.size(), levels,
                  s->octree ? s->fnodes.p : nullptr, s->octree ? s->oblocks.p : nullptr, s->octree ? s->wlinks.p : nullptr, nullptr, 0u, boxes[0].len2, max_inner_depth + 1, (uint32_t)hob.size()};
    s->hmats = hm;
    if (s->octree) {
        s->oct_lds = (size_t)std::max<uint32_t>(1u, max_inner_depth) * 8 * 64;  // 2 words per level per lane
        if (s->oct_lds > 64 * 1024) { s->octree = false; s->why_generic = "octree deeper than the LDS stack allows"; }
        else {
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_oct<false, false>, 64, s->oct_lds) == hipSuccess && nb > 0)
                s->oct_blocks_per_cu = nb;
            // ray-pool form: 24 state words + 2 per stack level, an odd number of 16-B quads per slot (conflict-free
            // ds_read_b128 of neighbouring slots); as many rays as fit 1/8 of the CU's 160 KB (8 waves per CU)
            uint32_t quads = (24u + 2u * std::max<uint32_t>(1u, max_inner_depth) + 3u) / 4u;
            if (!(quads & 1u)) quads++;
            const uint32_t stride = quads * 4u;
            const size_t budget = 160u * 1024u / 8u - 2u * RTMI_POOL_MAX;
            const uint32_t P = (uint32_t)std::min<size_t>(RTMI_POOL_MAX, budget / (stride * 4u));
            if (P >= 96u && hfn.size() / 2 < (1u << 22)) {
                s->pool_P = P; s->pool_stride = stride;
                s->pool_lds = (size_t)P * stride * 4u + 2u * RTMI_POOL_MAX;
                nb = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace_pool<false, false>, 64, s->pool_lds) == hipSuccess && nb > 0)
                    s->pool_blocks_per_cu = nb;
                else s->pool_P = 0;
            }
        }
    }
    s->htris.assign(tris, tris + ntris);  // rtmi_scene_set_corners rebuilds the fast-mode BVH from them
    {   // fast-mode BVH (cheap: binned SAH over the triangles' disc boxes); absent when a record is not finite
        const int rc = upload_bvh(s, nullptr);
        if (rc != RTMI_OK) return rc;
    }
    own.s = nullptr;
    *out = s;
    return RTMI_OK;
    RTMI_GUARD_END
}

int rtmi_scene_destroy(rtmi_scene_t* s) {
    if (!s) return RTMI_OK;
    (void)hipSetDevice(s->device);
    s->nodes.release(); s->refs.release(); s->tplane.release(); s->tedge.release(); s->mats.release();
    s->fnodes.release(); s->oblocks.release(); s->wlinks.release(); s->bnodes.release(); s->bleaves.release(); s->spheres.release();
    for (int k = 0; k < RTMI_MAX_STREAMS; k++) {
        s->w[k].release();
        if (s->istream[k]) (void)hipStreamDestroy(s->istream[k]);
        if (s->join_ev[k]) (void)hipEventDestroy(s->join_ev[k]);
    }
    if (s->fork_ev) (void)hipEventDestroy(s->fork_ev);
    if (s->end_ev) (void)hipEventDestroy(s->end_ev);
    s->tile.release(); s->qbytes.release(); s->mstage.release(); s->mframe.release();
    if (s->mstream) (void)hipStreamDestroy(s->mstream);
    if (!s->comms.empty()) { if (Rccl* r = rccl_api()) for (ncclComm_t c : s->comms) (void)r->CommDestroy(c); }
    delete s;
    return RTMI_OK;
}

int rtmi_scene_set_options(rtmi_scene_t* s, uint32_t options) {
    if (!s) return fail(RTMI_ERR_INVALID, "scene is NULL");
    s->options = options;
    return RTMI_OK;
}

int rtmi_scene_set_spheres(rtmi_scene_t* s, const rtmi_sphere_t* sp, uint64_t n) {
    if (!s) return fail(RTMI_ERR_INVALID, "scene is NULL");
    if (n && !sp) return fail(RTMI_ERR_INVALID, "spheres is NULL");
    if (n > 4096) return fail(RTMI_ERR_UNSUPPORTED, "more than 4096 analytic spheres (they are a flat list, not in the tree)");
    if ((uint64_t)s->d.ntris + n >= (1ull << 30)) return fail(RTMI_ERR_UNSUPPORTED, "hit index space exhausted");
    RTMI_GUARD_BEGIN
    auto fbits = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
    std::vector<float4> mats = s->hmats, rec(2 * n);
    for (uint64_t i = 0; i < n; i++) {
        if (sp[i].surface_kind > RTMI_REFLECTIVE) return fail(RTMI_ERR_INVALID, "unknown surface kind");
        const float alpha = sp[i].surface_kind == RTMI_SOLID ? 0.f : sp[i].alpha;
        const float scat = sp[i].surface_kind == RTMI_REFLECTIVE ? sp[i].scattering : 0.f;
        const uint32_t mid = (uint32_t)(mats.size() / 2);
        mats.push_back(make_float4(sp[i].color[0], sp[i].color[1], sp[i].color[2], alpha));
        mats.push_back(make_float4(scat, fbits(sp[i].surface_kind), 0.f, 0.f));
        rec[2 * i] = make_float4(sp[i].center[0], sp[i].center[1], sp[i].center[2], sp[i].radius);
        rec[2 * i + 1] = make_float4(fbits(mid), 0.f, 0.f, 0.f);
    }
    if (mats.size() / 2 > 65535) return fail(RTMI_ERR_UNSUPPORTED, "more than 65535 distinct surfaces");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipDeviceSynchronize());  // no render of this scene is in flight (one call at a time per handle)
    HIPCHK(s->mats.ensure(std::max<size_t>(mats.size(), 1)));
    HIPCHK(hipMemcpy(s->mats.p, mats.data(), mats.size() * sizeof(float4), hipMemcpyHostToDevice));
    HIPCHK(s->spheres.ensure(std::max<size_t>(rec.size(), 1)));
    if (n) HIPCHK(hipMemcpy(s->spheres.p, rec.data(), rec.size() * sizeof(float4), hipMemcpyHostToDevice));
    s->d.mats = s->mats.p;
    s->d.nmats = (uint32_t)(mats.size() / 2);
    s->d.spheres = s->spheres.p;
    s->d.nspheres = (uint32_t)n;
    return RTMI_OK;
    RTMI_GUARD_END
}

int rtmi_scene_set_corners(rtmi_scene_t* s, const float* corners9, uint64_t n) {
    if (!s) return fail(RTMI_ERR_INVALID, "scene is NULL");
    if (!corners9) return fail(RTMI_ERR_INVALID, "corners is NULL");
    if (n != s->htris.size()) return fail(RTMI_ERR_INVALID, "one corner triple per triangle of the scene (the sentinel included)");
    RTMI_GUARD_BEGIN
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipDeviceSynchronize());  // no render of this scene is in flight (one call at a time per handle)
    return upload_bvh(s, corners9);
    RTMI_GUARD_END
}

int rtmi_scene_get_tuning(rtmi_scene_t* s, rtmi_tuning_t* out) {
    if (!s || !out) return fail(RTMI_ERR_INVALID, "NULL argument");
    *out = s->tune;
    return RTMI_OK;
}

int rtmi_scene_set_tuning(rtmi_scene_t* s, const rtmi_tuning_t* in) {
    if (!s || !in) return fail(RTMI_ERR_INVALID, "NULL argument");
    if (in->batch_paths == 0 || in->streams > RTMI_MAX_STREAMS || in->oct_waves_per_cu > 32 || in->refill_min0 < 1 ||
        in->refill_min0 > 64 || in->refill_min < 1 || in->refill_min > 64 || in->xcd_aware > 2 || in->kernel > 2 || in->pipeline > 3 || in->slow_path_off > 1)
        return fail(RTMI_ERR_INVALID, "tuning value out of range");
    s->tune = *in;
    return RTMI_OK;
}

// fused = the path kernels only: ONE ray queue (the bounce rays of the primary pass), no hit records -- 62 B per path at
// depth 5 instead of 106
static int ensure_workspace(Work& w, size_t cap, uint32_t maxdepth, bool fused) {
    if (cap <= w.cap && maxdepth <= w.cap_depth && (fused || w.full)) return RTMI_OK;
    cap = std::max(cap, w.cap);
    maxdepth = std::max(maxdepth, w.cap_depth);
    const bool full = w.full || !fused;
    for (int k = 0; k < (full ? 2 : 1); k++) {
        HIPCHK(w.qo[k].ensure(cap));
        HIPCHK(w.qd[k].ensure(cap));
        HIPCHK(w.qpath[k].ensure(cap));
    }
    HIPCHK(w.scol.ensure(cap));
    if (full) {
        HIPCHK(w.hit_tf.ensure(cap));
        HIPCHK(w.hit_t.ensure(cap));
    }
    w.full = full;
    HIPCHK(w.mstack.ensure(cap * (size_t)maxdepth));
    HIPCHK(w.sqo.ensure(RTMI_SLOW_CAP)); HIPCHK(w.sqd.ensure(RTMI_SLOW_CAP));
    HIPCHK(w.sqpath.ensure(RTMI_SLOW_CAP)); HIPCHK(w.sqbounce.ensure(RTMI_SLOW_CAP));
    if (!w.sstream) {
        // highest priority: the persistent kernels of the ordinary passes never yield their wave slots, so the slow path's
        // blocks must be first in line whenever slots come free (the end of the producer launch), not behind the queued
        // blocks of the other streams' launches
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { (void)hipGetLastError(); lo = hi = 0; }
        HIPCHK(hipStreamCreateWithPriority(&w.sstream, hipStreamNonBlocking, hi));
    }
    if (!w.sdone) HIPCHK(hipEventCreateWithFlags(&w.sdone, hipEventDisableTiming));
    if (!w.sgo) HIPCHK(hipEventCreateWithFlags(&w.sgo, hipEventDisableTiming));
    while (w.sev.size() < (size_t)std::max<uint32_t>(maxdepth, 2u)) {
        hipEvent_t e;
        HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        w.sev.push_back(e);
    }
    while (w.pass_ev.size() < 2 * (size_t)std::max<uint32_t>(maxdepth, 2u)) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        w.pass_ev.push_back(e);
    }
    w.cap = cap;
    w.cap_depth = maxdepth;
    return RTMI_OK;
}

extern "C++" {
template <bool COUNT>
static void launch_trace(rtmi_scene* s, Work& w, hipStream_t st, const float4* qo, const float4* qd, int pass, hipEvent_t stop) {
    // `stop` is recorded right after the closest-hit kernel, so that the event pair of the caller times exactly
    // the kernel rocprofv3 lists as k_trace_oct / k_trace_linear / k_trace
    if ((s->options & RTMI_OPT_BVH) && s->bvh_ok) {
        const dim3 grid((unsigned)(s->num_cu * s->bvh_blocks_per_cu)), block(64);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_trace_bvh<COUNT>), grid, block, s->bvh_lds, st, s->d, s->bnodes.p, s->bleaves.p, s->bvh_root,
                           qo, qd, w.ctrl.p, pass, w.hit_tf.p, w.hit_t.p, (int)(pass == 0 ? s->tune.refill_min0 : s->tune.refill_min));
    } else if (s->root_is_leaf && !(s->options & RTMI_OPT_GENERIC)) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_trace_linear<COUNT>), dim3((unsigned)(s->num_cu * 8)), dim3(256), 0, st, s->d, qo, qd,
                           w.ctrl.p, pass, w.hit_tf.p, w.hit_t.p);
    } else if (s->octree && !(s->options & RTMI_OPT_GENERIC)) {
        // About 24 resident waves per CU in all is the optimum of this VALU-issue-bound kernel (more only adds cache
        // pressure): one stream launches what fits, two or more share the CUs with 16 each (two streams: 12 / 14 / 16 / 17 /
        // 18 / 20 / 24 waves per launch = 929 / 962 / 980 / 985 / 971 / 965 / 955 Mrays/s; one stream: 16 / 20 / 24 = 808 /
        // 882 / 931).
        const int per_cu = s->tune.oct_waves_per_cu ? (int)s->tune.oct_waves_per_cu
                                                     : s->active_streams > 1 ? std::min(s->oct_blocks_per_cu, 16) : s->oct_blocks_per_cu;
        const dim3 grid((unsigned)(s->num_cu * per_cu)), block(64);
        const int refill = (int)(pass == 0 ? s->tune.refill_min0 : s->tune.refill_min);
        const int xcd = (int)(s->tune.xcd_aware % 3u);  // 1 = ranges by XCC_ID, 2 = by blockIdx % 8, 0 = one range
        const bool pool = s->pool_P != 0 && s->tune.kernel != 1u && (s->tune.kernel == 2u || RTMI_DEFAULT_POOL);
        if (pool) {
            const int ppc = s->tune.oct_waves_per_cu ? std::min<int>((int)s->tune.oct_waves_per_cu, s->pool_blocks_per_cu) : s->pool_blocks_per_cu;
            const dim3 pgrid((unsigned)(s->num_cu * ppc));
            if (s->options & RTMI_OPT_FAST)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_trace_pool<COUNT, true>), pgrid, block, s->pool_lds, st, s->d, qo, qd, w.ctrl.p, pass,
                                   w.hit_tf.p, w.hit_t.p, refill, xcd, s->pool_P, s->pool_stride);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_trace_pool<COUNT, false>), pgrid, block, s->pool_lds, st, s->d, qo, qd, w.ctrl.p, pass,
                                   w.hit_tf.p, w.hit_t.p, refill, xcd, s->pool_P, s->pool_stride);
        } else {
            OctArgs a{};
            a.qo = qo; a.qd = qd; a.hit_tf = w.hit_tf.p; a.hit_t = w.hit_t.p; a.pass = pass;
            a.vote_s = pass == 0 ? s->vote[0] : s->vote[2]; a.vote_l = pass == 0 ? s->vote[1] : s->vote[3];
            if (s->options & RTMI_OPT_FAST)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_trace_oct<COUNT, true>), grid, block, s->oct_lds, st, s->d, a, w.ctrl.p, refill, xcd);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(k_trace_oct<COUNT, false>), grid, block, s->oct_lds, st, s->d, a, w.ctrl.p, refill, xcd);
        }
    } else {
        // persistent grid: enough blocks to fill every CU at the occupancy LDS allows
        const int per_cu = s->trace_block == 256 ? 4 : 16;
        const dim3 grid((unsigned)(s->num_cu * per_cu)), block((unsigned)s->trace_block);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_trace<COUNT>), grid, block, s->trace_lds, st, s->d, qo, qd, w.ctrl.p, pass,
                           w.hit_tf.p, w.hit_t.p);
    }
    (void)hipEventRecord(stop, st);
    if (s->d.nspheres)  // analytic spheres: a flat list against every ray, after the tree (not part of the timed trace kernel)
        hipLaunchKernelGGL(k_trace_spheres, dim3((unsigned)(s->num_cu * 8)), dim3(256), 0, st, s->d, qo, qd, w.ctrl.p, pass, w.hit_tf.p, w.hit_t.p);
}
// where a stream's zero-component rays go; cap 0 (tuning slow_path = 0... or the queue not allocated) keeps them in place
static SlowQ slow_queue(rtmi_scene* s, Work& w) {
    const bool on = s->tune.slow_path_off == 0u && w.sqo.p && w.sstream;
    return SlowQ{w.sqo.p, w.sqd.p, w.sqpath.p, w.sqbounce.p, on ? RTMI_SLOW_CAP : 0u};
}
// The fused path kernels (trace_oct.hpp): which = W_PRIMARY, W_BOUNCE or W_SLOW.  `stop` is recorded right after the kernel.
template <bool COUNT>
static void launch_path(rtmi_scene* s, Work& w, hipStream_t st, int which, const DView& dv, uint64_t seed, uint32_t pix0, uint32_t npaths,
                        hipEvent_t stop, int queue = 0) {
    const int per_cu = s->tune.oct_waves_per_cu ? (int)s->tune.oct_waves_per_cu
                                                 : s->active_streams > 1 ? std::min(s->oct_blocks_per_cu, 16) : s->oct_blocks_per_cu;
    const dim3 grid((unsigned)(s->num_cu * per_cu)), block(64);
    const int refill = (int)(which == W_PRIMARY ? s->tune.refill_min0 : s->tune.refill_min);
    const int xcd = (int)(s->tune.xcd_aware % 3u);
    OctArgs a{};
    a.v = dv; a.seed = seed; a.pix0 = pix0; a.npaths = npaths;
    const int bq = which == W_SLOW ? 0 : queue;  // (for W_SLOW `queue` is the consumer launch number)
    a.bqo = w.qo[bq].p; a.bqd = w.qd[bq].p; a.bqpath = w.qpath[bq].p;
    a.mstack = w.mstack.p; a.scol = w.scol.p;
    a.slow = slow_queue(s, w);
    a.vote_s = which == W_PRIMARY ? s->vote[0] : s->vote[2]; a.vote_l = which == W_PRIMARY ? s->vote[1] : s->vote[3];
    const bool fast = (s->options & RTMI_OPT_FAST) != 0;
    if (which == W_SLOW) {
        // consumer launch `queue` of the slow path, on the side stream: after the producer whose event is sev[queue]
        a.slow_k = (uint32_t)queue;
        (void)hipStreamWaitEvent(w.sstream, w.sev[queue], 0);
        hipLaunchKernelGGL(k_slow_snapshot, dim3(1), dim3(1), 0, w.sstream, w.ctrl.p, (uint32_t)queue, a.slow.cap);
        // The ordinary stream goes on only after the snapshot: its next persistent launch and the slow-path launch then become
        // ready together and the high-priority one gets its few wave slots first.  (Without this the next launch, already
        // queued in order, took every slot while the side stream was still resolving the event, and the slow paths started
        // a whole pass late.)
        (void)hipEventRecord(w.sgo, w.sstream);
        (void)hipStreamWaitEvent(st, w.sgo, 0);
        const dim3 sgrid((unsigned)std::max(s->num_cu / 2, 1));  // one path per wave at a time; a frame has ~100 such paths, a wave takes one after the other
        if (fast) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_path_slow<COUNT, true>), sgrid, block, s->oct_lds, w.sstream, s->d, a, w.ctrl.p, 1, 0);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_path_slow<COUNT, false>), sgrid, block, s->oct_lds, w.sstream, s->d, a, w.ctrl.p, 1, 0);
        return;
    }
    if (which == W_PRIMARY) {
        if (fast) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_path_primary<COUNT, true>), grid, block, s->oct_lds, st, s->d, a, w.ctrl.p, refill, xcd);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_path_primary<COUNT, false>), grid, block, s->oct_lds, st, s->d, a, w.ctrl.p, refill, xcd);
    } else {
        if (fast) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_path_bounce<COUNT, true>), grid, block, s->oct_lds, st, s->d, a, w.ctrl.p, refill, xcd);
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_path_bounce<COUNT, false>), grid, block, s->oct_lds, st, s->d, a, w.ctrl.p, refill, xcd);
    }
    (void)hipEventRecord(stop, st);
}
}  // extern "C++"

static int read_stats(Work& w, hipStream_t st, rtmi_stats_t* stats, float kernel_ms, float trace_ms, uint32_t launches) {
    DCtrl h;
    HIPCHK(hipMemcpyAsync(&h, w.ctrl.p, sizeof(DCtrl), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (stats) {
        stats->rays += h.rays;
        stats->box_tests += h.counters[0]; stats->tri_tests += h.counters[1]; stats->full_tests += h.counters[2];
        stats->nodes += h.counters[3]; stats->leaves += h.counters[4];
        stats->kernel_ms += kernel_ms; stats->trace_ms += trace_ms; stats->trace_launches += launches;
        stats->slow_paths += std::min<uint32_t>(h.scount, RTMI_SLOW_CAP);
    }
    return RTMI_OK;
}

int rtmi_render_device(rtmi_scene_t* s, const rtmi_viewport_t* vp, uint64_t seed, uint32_t row0, uint32_t nrows,
                       void* out_device, void* hip_stream, rtmi_stats_t* stats) {
    const rtmi_tile_t tile{row0, nrows, nrows ? nrows : 1u, 0u};
    return rtmi_render_tile_device(s, vp, seed, &tile, out_device, hip_stream, stats);
}

// One sub-tile = every nsub-th stripe of the caller's tile, rendered on its own stream with its own workspace.
struct SubTile {
    DView dv;          // row mapping of the sub-tile's local rows (tile_pixel)
    uint64_t npix = 0; // pixels of the sub-tile
    uint32_t index = 0;
};

int rtmi_render_tile_device(rtmi_scene_t* s, const rtmi_viewport_t* vp, uint64_t seed, const rtmi_tile_t* tile,
                            void* out_device, void* hip_stream, rtmi_stats_t* stats) {
    if (!s || !vp || !tile) return fail(RTMI_ERR_INVALID, "NULL argument");
    const uint32_t row0 = tile->row0, nrows = tile->nrows;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (nrows == 0) return RTMI_OK;
    RTMI_GUARD_BEGIN
    // leftovers of the caller's own HIP calls on this thread (or of failures this library tolerated, e.g. an occupancy
    // query) must not make a launch below look refused: hipGetLastError() reports the last error of ANY runtime call
    (void)hipGetLastError();
    if (!out_device) return fail(RTMI_ERR_INVALID, "NULL argument");
    if (vp->width == 0 || vp->height == 0) return fail(RTMI_ERR_INVALID, "empty viewport");
    if (tile->stripe_rows == 0) return fail(RTMI_ERR_INVALID, "stripe_rows must be >= 1");
    {
        const uint64_t nstripes = ((uint64_t)nrows + tile->stripe_rows - 1) / tile->stripe_rows;
        const uint64_t last_row = (uint64_t)row0 + (nstripes - 1) * tile->stripe_step + ((uint64_t)nrows - 1 - (nstripes - 1) * tile->stripe_rows);
        if (last_row >= vp->height) return fail(RTMI_ERR_INVALID, "row range outside the viewport");
        if (nstripes > 1 && tile->stripe_step < tile->stripe_rows) return fail(RTMI_ERR_INVALID, "stripes overlap");
    }
    if (vp->samples_per_pixel == 0) return fail(RTMI_ERR_INVALID, "samples_per_pixel must be >= 1");  // reference: 1/0 -> NaN image
    if (vp->maxdepth > RTMI_MAX_PASSES) return fail(RTMI_ERR_UNSUPPORTED, "maxdepth above 32");
    if ((uint64_t)vp->width * vp->height >= (1ull << 32)) return fail(RTMI_ERR_UNSUPPORTED, "more than 2^32 pixels");
    HIPCHK(hipSetDevice(s->device));
    hipStream_t ust = (hipStream_t)hip_stream;
    const uint32_t W = vp->width, spp = vp->samples_per_pixel, maxdepth = vp->maxdepth;
    const uint64_t npix = (uint64_t)nrows * W;
    float4* out = (float4*)out_device;
    if (maxdepth == 0) {  // project_ray returns black immediately (raytrace.rs:1261-1263); acc*(1/spp) of zeros
        HIPCHK(hipMemsetAsync(out, 0, npix * sizeof(float4), ust));
        HIPCHK(hipStreamSynchronize(ust));
        return RTMI_OK;
    }

    // ---- the tile's rows dealt out to the streams one row at a time (sub-tile t = rows t, t + nsub, ... of the tile): equal
    //      shares whatever the tile's own striping is.  (Round 2 dealt out whole stripes: with the 16-row stripes of an
    //      8-rank tiling a stream's stripes repeat every 384 image rows, the teapot covers two such periods and the slowest
    //      stream of a rank carried up to 12 % more rays than the others.)
    // path kernels (pipelines 2 and 3): exact-octree scenes traced by k_trace_oct's walk; everything else (linear list, generic
    // tree, BVH mode, ray-pool kernel, analytic spheres) runs one launch per bounce pass
    const bool pool_kernel = s->pool_P != 0 && s->tune.kernel != 1u && (s->tune.kernel == 2u || RTMI_DEFAULT_POOL);
    // 0 = automatic = 3: k_path_primary, then one launch per bounce pass (measured fastest on MI355X at every tile size,
    // DESIGN.md 4.1c)
    const bool hybrid_req = s->tune.pipeline == 3u || s->tune.pipeline == 0u;
    const bool fused = s->tune.pipeline != 1u && s->octree && !s->root_is_leaf && !(s->options & (RTMI_OPT_GENERIC | RTMI_OPT_BVH)) &&
                       !pool_kernel && s->d.nspheres == 0;
    // streams = 0 (automatic): one stream for path-kernel tiles of 2^26 paths and more, three otherwise (the per-pass
    // pipelines -- BVH mode: 29.6 ms on three streams, 35.2 on one -- have elementwise kernels to hide).  Since k_shade stopped being
    // atomic-bound (round 3) there is little left for a second stream to hide: the full config-3 frame takes 366.9 ms on one
    // stream and 371.8 on three (the sub-tiles' persistent launches compete for the same wave slots); a 1/8 tile 52.1 vs 51.6.
    const uint32_t auto_streams = (fused && npix * spp >= (1ull << 26)) ? 1u : 3u;
    uint32_t nsub = std::min<uint32_t>(s->tune.streams ? s->tune.streams : auto_streams, (uint32_t)RTMI_MAX_STREAMS);
    nsub = std::min<uint32_t>(nsub, nrows);
    if (npix * spp < s->tune.subtile_min_paths) nsub = 1;
    s->active_streams = nsub;
    SubTile sub[RTMI_MAX_STREAMS];
    for (uint32_t t = 0; t < nsub; t++) {
        DView& dv = sub[t].dv;
        dv.orig = mk(vp->orig[0], vp->orig[1], vp->orig[2]);
        dv.cam = mk(vp->cam[0], vp->cam[1], vp->cam[2]);
        dv.vu = mk(vp->vu[0], vp->vu[1], vp->vu[2]);
        dv.vv = mk(vp->vv[0], vp->vv[1], vp->vv[2]);
        dv.width = W; dv.height = vp->height; dv.maxdepth = maxdepth; dv.spp = spp;
        dv.row0 = row0; dv.stripe_rows = tile->stripe_rows; dv.stripe_step = tile->stripe_step; dv.pad = 0;
        dv.sub_mul = nsub; dv.sub_off = t;
        view_set_divisors(dv);
        sub[t].npix = (uint64_t)((nrows - t + nsub - 1) / nsub) * W;
        sub[t].index = t;
    }

    // batch = whole pixels with all their samples
    const size_t want_paths = (size_t)std::max<uint64_t>(s->tune.batch_paths, 1) / nsub;
    uint64_t pix_per_batch = std::max<uint64_t>(1, want_paths / spp);
    uint64_t max_sub_npix = 0;
    for (uint32_t t = 0; t < nsub; t++) max_sub_npix = std::max(max_sub_npix, sub[t].npix);
    // equal batches: a sub-tile a little larger than the budget (thirds of a frame whose stripes do not divide evenly)
    // would otherwise get a full batch and a sliver with five tiny passes of its own
    if (pix_per_batch < max_sub_npix) {
        const uint64_t nb = (max_sub_npix + pix_per_batch - 1) / pix_per_batch;
        pix_per_batch = max_sub_npix * 8 <= pix_per_batch * 9 ? max_sub_npix : (max_sub_npix + nb - 1) / nb;
    }
    pix_per_batch = std::min<uint64_t>(pix_per_batch, max_sub_npix);
    if (pix_per_batch * spp >= (1ull << 31)) return fail(RTMI_ERR_UNSUPPORTED, "batch above 2^31 paths");
    for (uint32_t t = 0; t < nsub; t++) {
        int rc = ensure_workspace(s->w[t], (size_t)(std::min<uint64_t>(pix_per_batch, sub[t].npix) * spp), maxdepth, fused && !hybrid_req);
        if (rc != RTMI_OK) return rc;
    }

    const bool counting = (s->options & RTMI_OPT_COUNTERS) != 0;
    const bool verbose = s->verbose;
    const unsigned ew_blocks = (unsigned)(s->num_cu * 8);
    float trace_ms = 0.f, primary_ms = 0.f, bounce_ms = 0.f;
    uint32_t launches = 0;
    // internal streams start after whatever the caller queued on its stream
    HIPCHK(hipEventRecord(s->fork_ev, ust));
    for (uint32_t t = 0; t < nsub; t++) HIPCHK(hipStreamWaitEvent(s->istream[t], s->fork_ev, 0));

    const uint64_t max_npix = max_sub_npix;
    for (uint64_t p0 = 0; p0 < max_npix; p0 += pix_per_batch) {
        // enqueue this batch of every sub-tile (no host dependency inside a batch: queue sizes live on the device)
        for (uint32_t t = 0; t < nsub; t++) {
            if (p0 >= sub[t].npix) continue;
            Work& w = s->w[t];
            hipStream_t st = s->istream[t];
            const DView& dv = sub[t].dv;
            const uint32_t np = (uint32_t)std::min<uint64_t>(pix_per_batch, sub[t].npix - p0);
            const uint32_t npaths = np * spp;
            const uint32_t pix0 = (uint32_t)p0;  // local pixel index inside the sub-tile
            HIPCHK(hipMemsetAsync(w.ctrl.p, 0, sizeof(DCtrl), st));
            HIPCHK(hipEventRecord(w.ev[0], st));
            if (fused && hybrid_req) {
                // primary rays generated, traced and shaded in one kernel (its bounce rays go to the queue pass 1 reads),
                // then one closest-hit + one shading launch per bounce pass
                HIPCHK(hipEventRecord(w.pass_ev[0], st));
                if (counting) launch_path<true>(s, w, st, W_PRIMARY, dv, seed, pix0, npaths, w.pass_ev[1], 1);
                else launch_path<false>(s, w, st, W_PRIMARY, dv, seed, pix0, npaths, w.pass_ev[1], 1);
                HIPCHK(hipGetLastError());
                launches++;
                // the slow path (zero-component rays the producer set aside) runs beside the following passes
                const bool slow_on = slow_queue(s, w).cap != 0u;
                auto slow_after = [&](uint32_t k) {
                    if (!slow_on) return;
                    (void)hipEventRecord(w.sev[k], st);
                    if (counting) launch_path<true>(s, w, st, W_SLOW, dv, seed, pix0, npaths, nullptr, (int)k);
                    else launch_path<false>(s, w, st, W_SLOW, dv, seed, pix0, npaths, nullptr, (int)k);
                };
                slow_after(0);
                HIPCHK(hipGetLastError());
                for (uint32_t pass = 1; pass < maxdepth; pass++) {
                    const int a = pass & 1, b = a ^ 1;
                    HIPCHK(hipEventRecord(w.pass_ev[2 * pass], st));
                    if (counting) launch_trace<true>(s, w, st, w.qo[a].p, w.qd[a].p, (int)pass, w.pass_ev[2 * pass + 1]);
                    else launch_trace<false>(s, w, st, w.qo[a].p, w.qd[a].p, (int)pass, w.pass_ev[2 * pass + 1]);
                    HIPCHK(hipGetLastError());
                    hipLaunchKernelGGL(k_shade, dim3(ew_blocks), dim3(256), 0, st, s->d, dv, seed, pix0, npaths, (int)pass,
                                       w.qo[a].p, w.qd[a].p, w.qpath[a].p, w.hit_tf.p, w.hit_t.p, w.qo[b].p, w.qd[b].p,
                                       w.qpath[b].p, w.mstack.p, w.scol.p, w.ctrl.p, slow_queue(s, w));
                    HIPCHK(hipGetLastError());
                    if (pass + 1 < maxdepth) slow_after(pass);  // the last pass's shading emits no rays
                    HIPCHK(hipGetLastError());
                    launches++;
                }
                if (slow_on) {  // the sample colours of the slow paths must be there before k_accum
                    HIPCHK(hipEventRecord(w.sdone, w.sstream));
                    HIPCHK(hipStreamWaitEvent(st, w.sdone, 0));
                }
            } else if (fused) {
                // primary rays generated, traced and shaded in one kernel; every bounce of every path in one more
                HIPCHK(hipEventRecord(w.pass_ev[0], st));
                if (counting) launch_path<true>(s, w, st, W_PRIMARY, dv, seed, pix0, npaths, w.pass_ev[1]);
                else launch_path<false>(s, w, st, W_PRIMARY, dv, seed, pix0, npaths, w.pass_ev[1]);
                HIPCHK(hipGetLastError());  // a refused launch is reported where it happens, not at the end of the batch
                launches++;
                const bool slow_on = slow_queue(s, w).cap != 0u;
                if (slow_on) {  // zero-component primary rays (and bounce rays of the primary hits): beside the bounce kernel
                    (void)hipEventRecord(w.sev[0], st);
                    if (counting) launch_path<true>(s, w, st, W_SLOW, dv, seed, pix0, npaths, nullptr, 0);
                    else launch_path<false>(s, w, st, W_SLOW, dv, seed, pix0, npaths, nullptr, 0);
                    HIPCHK(hipGetLastError());
                    HIPCHK(hipEventRecord(w.sdone, w.sstream));
                }
                if (maxdepth > 1) {
                    HIPCHK(hipEventRecord(w.pass_ev[2], st));
                    if (counting) launch_path<true>(s, w, st, W_BOUNCE, dv, seed, pix0, npaths, w.pass_ev[3]);
                    else launch_path<false>(s, w, st, W_BOUNCE, dv, seed, pix0, npaths, w.pass_ev[3]);
                    HIPCHK(hipGetLastError());
                    launches++;
                }
                if (slow_on) HIPCHK(hipStreamWaitEvent(st, w.sdone, 0));
            } else {
            hipLaunchKernelGGL(k_gen, dim3(ew_blocks), dim3(256), 0, st, dv, seed, pix0, npaths, w.qo[0].p, w.qd[0].p, w.qpath[0].p, w.ctrl.p);
            HIPCHK(hipGetLastError());  // a refused launch is reported where it happens, not at the end of the batch
            for (uint32_t pass = 0; pass < maxdepth; pass++) {
                const int a = pass & 1, b = a ^ 1;
                HIPCHK(hipEventRecord(w.pass_ev[2 * pass], st));
                if (counting) launch_trace<true>(s, w, st, w.qo[a].p, w.qd[a].p, (int)pass, w.pass_ev[2 * pass + 1]);
                else launch_trace<false>(s, w, st, w.qo[a].p, w.qd[a].p, (int)pass, w.pass_ev[2 * pass + 1]);
                HIPCHK(hipGetLastError());
                if (counting && verbose) {
                    DCtrl hc2;
                    HIPCHK(hipMemcpyAsync(&hc2, w.ctrl.p, sizeof(DCtrl), hipMemcpyDeviceToHost, st));
                    HIPCHK(hipStreamSynchronize(st));
                    unsigned long long (*prev)[13] = s->vprev;
                    if (pass == 0) memset(prev[t], 0, sizeof(s->vprev[t]));
                    unsigned long long cur[13];
                    for (int k = 0; k < 5; k++) cur[k] = hc2.counters[k];
                    for (int k = 0; k < 8; k++) cur[5 + k] = hc2.dbg[k];
                    const double n = hc2.count[pass] ? (double)hc2.count[pass] : 1.0;
                    fprintf(stderr, "[rtmi]   stream %u pass %u per ray: box %.1f tri %.1f full %.2f nodes %.1f leaves %.1f | S-steps %.1f (util %.2f) L-steps %.1f (util %.2f)\n",
                            t, pass, (cur[0] - prev[t][0]) / n, (cur[1] - prev[t][1]) / n, (cur[2] - prev[t][2]) / n, (cur[3] - prev[t][3]) / n, (cur[4] - prev[t][4]) / n,
                            (cur[6] - prev[t][6]) / n, (double)(cur[6] - prev[t][6]) / (64.0 * (cur[5] - prev[t][5] ? cur[5] - prev[t][5] : 1)),
                            (cur[8] - prev[t][8]) / n, (double)(cur[8] - prev[t][8]) / (64.0 * (cur[7] - prev[t][7] ? cur[7] - prev[t][7] : 1)));
                    memcpy(prev[t], cur, sizeof(s->vprev[t]));
                }
                hipLaunchKernelGGL(k_shade, dim3(ew_blocks), dim3(256), 0, st, s->d, dv, seed, pix0, npaths, (int)pass,
                                   w.qo[a].p, w.qd[a].p, w.qpath[a].p, w.hit_tf.p, w.hit_t.p, w.qo[b].p, w.qd[b].p,
                                   w.qpath[b].p, w.mstack.p, w.scol.p, w.ctrl.p, SlowQ{nullptr, nullptr, nullptr, nullptr, 0u});
                HIPCHK(hipGetLastError());
                launches++;
            }
            }
            hipLaunchKernelGGL(k_accum, dim3(ew_blocks), dim3(256), 0, st, np, spp, w.scol.p, (float*)out, pix0, W, nsub, t, make_fastdiv(W));
            HIPCHK(hipEventRecord(w.ev[1], st));
            HIPCHK(hipGetLastError());
        }
        // collect: counters and per-launch trace times of every sub-tile's batch
        for (uint32_t t = 0; t < nsub; t++) {
            if (p0 >= sub[t].npix) continue;
            Work& w = s->w[t];
            rtmi_stats_t bs; memset(&bs, 0, sizeof(bs));
            int rc = read_stats(w, s->istream[t], &bs, 0.f, 0.f, 0);
            if (rc != RTMI_OK) return rc;
            DCtrl hc;
            if (verbose) HIPCHK(hipMemcpy(&hc, w.ctrl.p, sizeof(DCtrl), hipMemcpyDeviceToHost));
            const uint32_t ntimed = (fused && !hybrid_req) ? (maxdepth > 1 ? 2u : 1u) : maxdepth;  // fused: primary kernel, bounce kernel
            for (uint32_t pass = 0; pass < ntimed; pass++) {
                float pm = 0.f;
                HIPCHK(hipEventElapsedTime(&pm, w.pass_ev[2 * pass], w.pass_ev[2 * pass + 1]));
                trace_ms += pm;
                if (fused) { if (pass == 0) primary_ms += pm; else bounce_ms += pm; }
                if (verbose) fprintf(stderr, "[rtmi] stream %u batch@%llu pass %u: %u rays, trace %.3f ms, %.1f Mrays/s\n", t, (unsigned long long)p0, pass, hc.count[pass], pm, hc.count[pass] / (pm * 1e3));
            }
            if (stats) {
                stats->rays += bs.rays; stats->box_tests += bs.box_tests; stats->tri_tests += bs.tri_tests;
                stats->full_tests += bs.full_tests; stats->nodes += bs.nodes; stats->leaves += bs.leaves;
                stats->slow_paths += bs.slow_paths;
            }
        }
    }
    // the caller's stream continues after both internal streams
    for (uint32_t t = 0; t < nsub; t++) {
        HIPCHK(hipEventRecord(s->join_ev[t], s->istream[t]));
        HIPCHK(hipStreamWaitEvent(ust, s->join_ev[t], 0));
    }
    HIPCHK(hipEventRecord(s->end_ev, ust));
    HIPCHK(hipEventSynchronize(s->end_ev));
    float kernel_ms = 0.f;
    HIPCHK(hipEventElapsedTime(&kernel_ms, s->fork_ev, s->end_ev));
    if (stats) {
        stats->kernel_ms = kernel_ms; stats->trace_ms = trace_ms; stats->trace_launches = launches; stats->streams = nsub;
        stats->primary_ms = primary_ms; stats->bounce_ms = bounce_ms; stats->pipeline = fused ? (hybrid_req ? 3u : 2u) : 1u;
    }
    return RTMI_OK;
    RTMI_GUARD_END
}

int rtmi_render(rtmi_scene_t* s, const rtmi_viewport_t* vp, uint64_t seed, uint32_t row0, uint32_t nrows,
                float* out_host, rtmi_stats_t* stats) {
    if (!s || !vp || !out_host) return fail(RTMI_ERR_INVALID, "NULL argument");
    HIPCHK(hipSetDevice(s->device));
    const uint64_t npix = (uint64_t)nrows * vp->width;
    if (npix == 0) { if (stats) memset(stats, 0, sizeof(*stats)); return RTMI_OK; }
    if ((uint64_t)row0 + nrows > vp->height) return fail(RTMI_ERR_INVALID, "row range outside the viewport");
    HIPCHK(s->tile.ensure(npix));
    int rc = rtmi_render_device(s, vp, seed, row0, nrows, s->tile.p, nullptr, stats);
    if (rc != RTMI_OK) return rc;
    HIPCHK(hipMemcpy(out_host, s->tile.p, npix * sizeof(float4), hipMemcpyDeviceToHost));
    return RTMI_OK;
}

// Direct (xGMI) access from the scene's device to `root`, asked for ONCE per (device, root) pair and remembered on the
// handle: hipDeviceEnablePeerAccess answers hipErrorPeerAccessAlreadyEnabled from the second call on and leaves that
// error pending for the thread's next hipGetLastError() poll (the launch checks of rtmi_render_tile_device).  Whatever the
// answer, the pending error is cleared here; a refusal is recorded (peer_ok = 0, peer_msg) -- hipMemcpyPeerAsync still
// works then, staged by the runtime, and rtmi_render_frame_multi reports it instead of running slowly in silence.
static hipError_t ensure_peer_access(rtmi_scene* sc, int root) {
    if (sc->peer_root == root) return hipSuccess;
    sc->peer_root = root;
    sc->peer_msg.clear();
    if (sc->device == root) { sc->peer_ok = 1; return hipSuccess; }
    const hipError_t pe = hipDeviceEnablePeerAccess(root, 0);
    (void)hipGetLastError();
    sc->peer_ok = (pe == hipSuccess || pe == hipErrorPeerAccessAlreadyEnabled) ? 1 : 0;
    if (!sc->peer_ok) sc->peer_msg = hipGetErrorString(pe);
    return hipSuccess;
}

// Development/test aid (not in rtmi.h): run the peer-access step for `scene` against `root_device` again, as a second
// frame would, and report what is pending afterwards (0 = nothing).  forget != 0 drops the cached answer first.
int rtmi_debug_peer_access(rtmi_scene_t* s, int root_device, int forget, int* peer_ok, int* pending_error) {
    if (!s) return fail(RTMI_ERR_INVALID, "scene is NULL");
    HIPCHK(hipSetDevice(s->device));
    if (forget) s->peer_root = -1;
    HIPCHK(ensure_peer_access(s, root_device));
    if (peer_ok) *peer_ok = s->peer_ok;
    if (pending_error) *pending_error = (int)hipGetLastError();
    return RTMI_OK;
}

int rtmi_render_frame_multi(rtmi_scene_t* const* scenes, uint32_t nscenes, const rtmi_viewport_t* vp, uint64_t seed,
                            uint32_t stripe_rows, uint32_t flags, void* out_host, void* out_device, rtmi_stats_t* stats) {
    if (!scenes || nscenes == 0 || !vp) return fail(RTMI_ERR_INVALID, "NULL argument");
    if (nscenes > 64) return fail(RTMI_ERR_UNSUPPORTED, "more than 64 scene handles");
    for (uint32_t i = 0; i < nscenes; i++) {
        if (!scenes[i]) return fail(RTMI_ERR_INVALID, "scene handle is NULL");
        for (uint32_t j = 0; j < i; j++)
            if (scenes[j] == scenes[i]) return fail(RTMI_ERR_INVALID, "the same scene handle is listed twice (one render in flight per handle)");
    }
    if (!out_host && !out_device) return fail(RTMI_ERR_INVALID, "out_host and out_device are both NULL");
    if (vp->width == 0 || vp->height == 0) return fail(RTMI_ERR_INVALID, "empty viewport");
    if (flags & ~(uint32_t)(RTMI_FRAME_RGB8 | RTMI_FRAME_RCCL)) return fail(RTMI_ERR_INVALID, "unknown flag");
    RTMI_GUARD_BEGIN
    const bool use_rccl = (flags & RTMI_FRAME_RCCL) != 0;
    Rccl* rccl = nullptr;
    if (use_rccl) {
        // one communicator rank per scene handle: RCCL wants every rank on a device of its own
        for (uint32_t i = 0; i < nscenes; i++)
            for (uint32_t j = 0; j < i; j++)
                if (scenes[j]->device == scenes[i]->device)
                    return fail(RTMI_ERR_UNSUPPORTED, "RTMI_FRAME_RCCL needs every scene handle on a device of its own (RCCL refuses two ranks on one GPU)");
        rccl = rccl_api();
        if (!rccl) return fail(RTMI_ERR_UNSUPPORTED, "RTMI_FRAME_RCCL: librccl.so.1 could not be loaded");
    }
    const uint32_t W = vp->width, H = vp->height, n = nscenes;
    const uint32_t S = stripe_rows ? stripe_rows : 16u;
    const bool rgb8 = (flags & RTMI_FRAME_RGB8) != 0;
    const uint32_t px = rgb8 ? 3u : 16u;
    // rows of scene i: stripes i, i+n, i+2n, ... of S rows (the last stripe of the image may be short)
    std::vector<uint32_t> rows(n, 0);
    for (uint32_t k = 0; (uint64_t)k * S < H; k++) rows[k % n] += std::min<uint32_t>(S, H - k * S);
    const uint32_t mr = *std::max_element(rows.begin(), rows.end());
    rtmi_scene* root = scenes[0];
    HIPCHK(hipSetDevice(root->device));
    HIPCHK(root->mstage.ensure((size_t)n * mr * W * px));
    if (!out_device) { HIPCHK(root->mframe.ensure((size_t)H * W * px)); }
    uint8_t* frame = out_device ? (uint8_t*)out_device : root->mframe.p;

    // ---- fan-out: one host thread per scene handle renders its tile and sends the band to the root device
    std::vector<int> rcs(n, RTMI_OK);
    std::vector<std::string> errs(n);
    std::vector<rtmi_stats_t> sts(n);
    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };
    auto work_body = [&](uint32_t i) {
        rtmi_scene* sc = scenes[i];
        rtmi_stats_t& st = sts[i];
        auto bail = [&](int rc, const std::string& msg) { rcs[i] = rc; errs[i] = msg; };
        hipError_t e = hipSetDevice(sc->device);
        if (e == hipSuccess) e = ensure_peer_access(sc, root->device);  // once per (device, root) pair, cached on the handle
        const int peer = sc->peer_ok;
        if (rows[i] == 0) { st.peer_access = peer; return; }
        if (e == hipSuccess && !sc->mstream) e = hipStreamCreateWithFlags(&sc->mstream, hipStreamNonBlocking);
        // (a gather sends the same count from every rank: the band buffers then hold the padded band of `mr` rows)
        if (e == hipSuccess) e = sc->tile.ensure((size_t)(use_rccl ? mr : rows[i]) * W);
        if (e == hipSuccess && rgb8) e = sc->qbytes.ensure((size_t)(use_rccl ? mr : rows[i]) * W * 3);
        if (e != hipSuccess) return bail(hip_code(e), std::string("rtmi_render_frame_multi: ") + hipGetErrorString(e));
        const rtmi_tile_t tile{i * S, rows[i], S, n * S};
        const clk::time_point t0 = clk::now();
        int rc = rtmi_render_tile_device(sc, vp, seed, &tile, sc->tile.p, sc->mstream, &st);
        if (rc != RTMI_OK) return bail(rc, g_err);
        st.render_ms = ms_since(t0);
        st.peer_access = peer;
        const void* band = sc->tile.p;
        const clk::time_point t1 = clk::now();
        if (rgb8) {
            hipLaunchKernelGGL(k_quantize, dim3((unsigned)(sc->num_cu * 8)), dim3(256), 0, sc->mstream, (uint64_t)rows[i] * W,
                               (const float4*)sc->tile.p, sc->qbytes.p);
            band = sc->qbytes.p;
        }
        if (use_rccl) {  // the bands cross together below, in ONE ncclGather; here only the quantisation is awaited
            e = hipStreamSynchronize(sc->mstream);
            if (e != hipSuccess) return bail(hip_code(e), std::string("rtmi_render_frame_multi: ") + hipGetErrorString(e));
            return;
        }
        // the single crossing of this band: its own link to the root (a plain device copy when both are one device)
        e = hipMemcpyPeerAsync(root->mstage.p + (size_t)i * mr * W * px, root->device, band, sc->device, (size_t)rows[i] * W * px, sc->mstream);
        if (e == hipSuccess) e = hipStreamSynchronize(sc->mstream);
        if (e != hipSuccess) return bail(hip_code(e), std::string("rtmi_render_frame_multi: band copy: ") + hipGetErrorString(e));
        st.band_copy_ms = ms_since(t1);
    };
    // no C++ exception leaves a worker thread (std::terminate) or crosses the ABI
    auto work = [&](uint32_t i) {
        memset(&sts[i], 0, sizeof(rtmi_stats_t));
        try { work_body(i); }
        catch (const std::bad_alloc&) { rcs[i] = RTMI_ERR_OOM; try { errs[i] = "host allocation failed"; } catch (...) {} }
        catch (...) { rcs[i] = RTMI_ERR_INVALID; try { errs[i] = "internal error"; } catch (...) {} }
    };
    if (n == 1) work(0);
    else {
        struct Joiner {  // joins what was started, also when starting a later thread throws
            std::vector<std::thread> th;
            ~Joiner() { for (auto& t : th) if (t.joinable()) t.join(); }
        } pool;
        pool.th.reserve(n);
        for (uint32_t i = 0; i < n; i++) pool.th.emplace_back(work, i);
    }
    std::string warn;
    for (uint32_t i = 0; i < n; i++) {
        if (rcs[i] != RTMI_OK) return fail(rcs[i], "scene " + std::to_string(i) + ": " + errs[i]);
        if (!scenes[i]->peer_ok && warn.empty())
            warn = "warning: device " + std::to_string(scenes[i]->device) + " has no peer access to root device " + std::to_string(root->device) +
                   " (" + scenes[i]->peer_msg + "): its band is staged by the runtime, see rtmi_stats_t.peer_access / band_copy_ms";
    }
    if (use_rccl) {
        // ---- ONE ncclGather over the scenes' devices (rccl.h ncclGather): every rank sends its padded band, the
        //      root receives them rank-major into the staging buffer the de-interleave kernel reads.  Communicators are
        //      made once per device list (ncclCommInitAll) and kept on the root handle.
        std::vector<int> devs(n);
        for (uint32_t i = 0; i < n; i++) devs[i] = scenes[i]->device;
        if (root->comm_devices != devs) {
            for (ncclComm_t c : root->comms) (void)rccl->CommDestroy(c);
            root->comms.assign(n, nullptr);
            root->comm_devices.clear();
            const ncclResult_t r = rccl->CommInitAll(root->comms.data(), (int)n, devs.data());
            if (r != ncclSuccess) { root->comms.clear(); return fail(RTMI_ERR_DEVICE, std::string("ncclCommInitAll: ") + rccl->GetErrorString(r)); }
            root->comm_devices = devs;
        }
        const clk::time_point t1 = clk::now();
        const size_t count = (size_t)mr * W * px;  // bytes per rank
        ncclResult_t r = rccl->GroupStart();
        for (uint32_t i = 0; i < n && r == ncclSuccess; i++) {
            rtmi_scene* sc = scenes[i];
            HIPCHK(hipSetDevice(sc->device));
            if (!sc->mstream) HIPCHK(hipStreamCreateWithFlags(&sc->mstream, hipStreamNonBlocking));
            if (rows[i] == 0) { HIPCHK(sc->tile.ensure((size_t)mr * W)); if (rgb8) HIPCHK(sc->qbytes.ensure((size_t)mr * W * 3)); }
            const void* band = rgb8 ? (const void*)sc->qbytes.p : (const void*)sc->tile.p;
            r = rccl->Gather(band, i == 0 ? (void*)root->mstage.p : nullptr, count, ncclUint8, 0, root->comms[i], sc->mstream);
        }
        const ncclResult_t r2 = rccl->GroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) return fail(RTMI_ERR_DEVICE, std::string("ncclGather: ") + rccl->GetErrorString(r));
        for (uint32_t i = 0; i < n; i++) {
            HIPCHK(hipSetDevice(scenes[i]->device));
            HIPCHK(hipStreamSynchronize(scenes[i]->mstream));
            sts[i].band_copy_ms = ms_since(t1);
        }
    }
    // ---- root: de-interleave the stripes into the frame
    HIPCHK(hipSetDevice(root->device));
    if (!root->mstream) HIPCHK(hipStreamCreateWithFlags(&root->mstream, hipStreamNonBlocking));
    const clk::time_point t2 = clk::now();
    hipLaunchKernelGGL(k_deinterleave, dim3((unsigned)(root->num_cu * 8)), dim3(256), 0, root->mstream, root->mstage.p, frame, W, H, S, n, mr, px);
    HIPCHK(hipGetLastError());
    if (out_host) HIPCHK(hipMemcpyAsync(out_host, frame, (size_t)H * W * px, hipMemcpyDeviceToHost, root->mstream));
    HIPCHK(hipStreamSynchronize(root->mstream));
    sts[0].deinterleave_ms = ms_since(t2);
    if (stats) memcpy(stats, sts.data(), sizeof(rtmi_stats_t) * n);
    g_err = warn;  // RTMI_OK with a non-empty rtmi_last_error(): the frame is right, a link is slow
    return RTMI_OK;
    RTMI_GUARD_END
}

int rtmi_trace(rtmi_scene_t* s, uint64_t n, const float* orig4, const float* dir4, uint32_t* tri, float* t,
               uint32_t* face, rtmi_stats_t* stats) {
    if (!s) return fail(RTMI_ERR_INVALID, "scene is NULL");
    if (stats) memset(stats, 0, sizeof(*stats));
    if (n == 0) return RTMI_OK;
    if (!orig4 || !dir4 || !tri || !t || !face) return fail(RTMI_ERR_INVALID, "NULL argument");
    if (n >= (1ull << 31)) return fail(RTMI_ERR_UNSUPPORTED, "more than 2^31 rays per call");
    RTMI_GUARD_BEGIN
    HIPCHK(hipSetDevice(s->device));
    Work& w = s->w[0];
    int rc = ensure_workspace(w, (size_t)n, 1, false);
    if (rc != RTMI_OK) return rc;
    hipStream_t st = s->istream[0];
    HIPCHK(hipMemcpyAsync(w.qo[0].p, orig4, n * 16, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(w.qd[0].p, dir4, n * 16, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemsetAsync(w.ctrl.p, 0, sizeof(DCtrl), st));
    hipLaunchKernelGGL(k_set_count, dim3(1), dim3(1), 0, st, w.ctrl.p, (uint32_t)n);
    HIPCHK(hipEventRecord(w.ev[0], st));
    s->active_streams = 1;
    if (s->options & RTMI_OPT_COUNTERS) launch_trace<true>(s, w, st, w.qo[0].p, w.qd[0].p, 0, w.ev[1]);
    else launch_trace<false>(s, w, st, w.qo[0].p, w.qd[0].p, 0, w.ev[1]);
    HIPCHK(hipGetLastError());
    std::vector<uint32_t> tf(n);
    HIPCHK(hipMemcpyAsync(tf.data(), w.hit_tf.p, n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(t, w.hit_t.p, n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (uint64_t i = 0; i < n; i++) { tri[i] = tf[i] & 0x3FFFFFFFu; face[i] = tf[i] >> 30; }
    // face encoding of the ABI: 0 front 1 back 2 edge-front 3 edge-back (bit0 = back, bit1 = edge)
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, w.ev[0], w.ev[1]));
    rc = read_stats(w, st, stats, ms, ms, 1);
    if (stats) stats->streams = 1;
    return rc;
    RTMI_GUARD_END
}

// Test hook (not in rtmi.h): n / d as the kernels compute it for launch-constant divisors (FastDiv, shade.hpp), evaluated on
// the host with the same inline functions; needs no device.
uint32_t rtmi_debug_fastdiv(uint32_t n, uint32_t d) { return fdiv(n, make_fastdiv(d)); }

// Test hook (not in rtmi.h): the ABI status and message a HIP runtime failure `hip_error` is reported as.
int rtmi_debug_status_of(int hip_error) {
    HIPCHK((hipError_t)hip_error);
    return RTMI_OK;
}

// Development aid (not in rtmi.h): step statistics of the last counting render/trace.
int rtmi_debug_counters(rtmi_scene_t* s, unsigned long long* out16) {
    if (!s || !out16) return fail(RTMI_ERR_INVALID, "NULL argument");
    HIPCHK(hipSetDevice(s->device));
    memset(out16, 0, 16 * sizeof(unsigned long long));
    for (int k = 0; k < RTMI_MAX_STREAMS; k++) {
        DCtrl h;
        HIPCHK(hipMemcpy(&h, s->w[k].ctrl.p, sizeof(DCtrl), hipMemcpyDeviceToHost));
        for (int j = 0; j < 16; j++) out16[j] += h.dbg[j];
    }
    return RTMI_OK;
}

int rtmi_make_triangles(int device, const float* corners9_host, uint64_t n, const rtmi_triangle_t* proto, rtmi_triangle_t* out_host) {
    if (n == 0) return RTMI_OK;
    if (!corners9_host || !proto || !out_host) return fail(RTMI_ERR_INVALID, "NULL argument");
    if (n >= (1ull << 30)) return fail(RTMI_ERR_UNSUPPORTED, "more than 2^30 triangles");
    const int ndev = rtmi_device_count();
    if (ndev <= 0) return fail(RTMI_ERR_NO_DEVICE, "no HIP device visible: the MI355X kernels cannot run (there is no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(RTMI_ERR_INVALID, "device index out of range");
    RTMI_GUARD_BEGIN
    HIPCHK(hipSetDevice(device));
    DevBuf<float> dpts, dout;
    DevBuf<uint32_t> dok;
    auto cleanup = [&]() { dpts.release(); dout.release(); dok.release(); };
    hipError_t e = dpts.ensure(n * 9);
    if (e == hipSuccess) e = dout.ensure(n * 20);
    if (e == hipSuccess) e = dok.ensure(n);
    if (e == hipSuccess) e = hipMemcpy(dpts.p, corners9_host, n * 9 * sizeof(float), hipMemcpyHostToDevice);
    std::vector<float> rec(n * 20);
    std::vector<uint32_t> ok(n);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_make_triangles, dim3((unsigned)std::min<uint64_t>((n + 255) / 256, 2048)), dim3(256), 0, nullptr, (uint32_t)n,
                           dpts.p, dout.p, dok.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(rec.data(), dout.p, n * 20 * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(ok.data(), dok.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost);
    cleanup();
    if (e != hipSuccess) return fail(hip_code(e), std::string("rtmi_make_triangles: ") + hipGetErrorString(e));
    for (uint64_t i = 0; i < n; i++) {
        if (!ok[i]) return fail(RTMI_ERR_INVALID, "make_triangle: degenerate triangle " + std::to_string(i) + " (the reference panics at raytrace.rs:357)");
        rtmi_triangle_t t = *proto;
        const float* o = rec.data() + i * 20;
        memcpy(t.incenter, o, 12); memcpy(t.norm, o + 3, 12); t.bounding_r2 = o[6];
        memcpy(t.sides, o + 7, 36); memcpy(t.side_lens, o + 16, 12);
        out_host[i] = t;
    }
    return RTMI_OK;
    RTMI_GUARD_END
}

struct rtmi_builder {
    int device = 0;
    uint64_t ntris = 0;
    int num_cu = 256;
    DevBuf<float> tris;
    DevBuf<float4> geo;
    DevBuf<uint4> rng;
    DevBuf<uint2> items;
    DevBuf<uint32_t> cand;
    DevBuf<uint8_t> keep;
};

int rtmi_builder_create(int device, const float* tris15, uint64_t ntris, rtmi_builder_t** out) {
    if (!out) return fail(RTMI_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!tris15 || ntris == 0) return fail(RTMI_ERR_INVALID, "no triangles");
    if (ntris >= (1ull << 30)) return fail(RTMI_ERR_UNSUPPORTED, "more than 2^30 triangles");
    const int ndev = rtmi_device_count();
    if (ndev <= 0) return fail(RTMI_ERR_NO_DEVICE, "no HIP device visible: the MI355X kernels cannot run (there is no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(RTMI_ERR_INVALID, "device index out of range");
    RTMI_GUARD_BEGIN
    HIPCHK(hipSetDevice(device));
    rtmi_builder* b = new rtmi_builder();
    b->device = device; b->ntris = ntris;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) b->num_cu = prop.multiProcessorCount;
    hipError_t e = b->tris.ensure(ntris * 15);
    if (e == hipSuccess) e = hipMemcpy(b->tris.p, tris15, ntris * 15 * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { rtmi_builder_destroy(b); return fail(hip_code(e), std::string("rtmi_builder_create: ") + hipGetErrorString(e)); }
    *out = b;
    return RTMI_OK;
    RTMI_GUARD_END
}

int rtmi_builder_destroy(rtmi_builder_t* b) {
    if (!b) return RTMI_OK;
    (void)hipSetDevice(b->device);
    b->tris.release(); b->geo.release(); b->rng.release(); b->items.release(); b->cand.release(); b->keep.release();
    delete b;
    return RTMI_OK;
}

int rtmi_builder_filter(rtmi_builder_t* b, const rtmi_build_box_t* boxes, uint64_t nboxes, const uint32_t* cand, uint64_t ncand,
                        uint8_t* keep, uint64_t nkeep) {
    if (!b) return fail(RTMI_ERR_INVALID, "builder is NULL");
    if (nboxes == 0 || nkeep == 0) return RTMI_OK;
    if (!boxes || !cand || !keep) return fail(RTMI_ERR_INVALID, "NULL argument");
    if (nboxes >= (1ull << 32) || ncand >= (1ull << 32)) return fail(RTMI_ERR_UNSUPPORTED, "level too large for 32-bit indices");
    RTMI_GUARD_BEGIN
    // validate on the host: every range inside `cand` / `keep`, every candidate a triangle of the handle
    for (uint64_t k = 0; k < ncand; k++)
        if (cand[k] >= b->ntris) return fail(RTMI_ERR_INVALID, "candidate triangle index out of range");
    std::vector<float4> geo(nboxes);
    std::vector<uint4> rng(nboxes);
    std::vector<uint2> items;
    for (uint64_t i = 0; i < nboxes; i++) {
        const rtmi_build_box_t& bx = boxes[i];
        if ((uint64_t)bx.cand_first + bx.cand_count > ncand || bx.keep_first + bx.cand_count > nkeep)
            return fail(RTMI_ERR_INVALID, "box candidate range out of bounds");
        geo[i] = make_float4(bx.orig[0], bx.orig[1], bx.orig[2], bx.len2);
        rng[i] = make_uint4(bx.cand_first, bx.cand_count, (uint32_t)bx.keep_first, (uint32_t)(bx.keep_first >> 32));
        for (uint32_t lo = 0; lo < bx.cand_count; lo += 256) items.push_back(make_uint2((uint32_t)i, lo));
    }
    if (items.empty()) return RTMI_OK;
    if (items.size() >= (1ull << 32)) return fail(RTMI_ERR_UNSUPPORTED, "level too large");
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(b->geo.ensure(nboxes)); HIPCHK(b->rng.ensure(nboxes)); HIPCHK(b->items.ensure(items.size()));
    HIPCHK(b->cand.ensure(ncand)); HIPCHK(b->keep.ensure(nkeep));
    HIPCHK(hipMemcpy(b->geo.p, geo.data(), nboxes * sizeof(float4), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->rng.p, rng.data(), nboxes * sizeof(uint4), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->items.p, items.data(), items.size() * sizeof(uint2), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->cand.p, cand, ncand * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(b->keep.p, 0, nkeep));
    const unsigned grid = (unsigned)std::min<uint64_t>(items.size(), (uint64_t)b->num_cu * 64);
    hipLaunchKernelGGL(k_box_contains, dim3(grid), dim3(256), 0, nullptr, b->tris.p, b->geo.p, b->rng.p, b->items.p, (uint32_t)items.size(),
                       b->cand.p, b->keep.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(keep, b->keep.p, nkeep, hipMemcpyDeviceToHost));
    return RTMI_OK;
    RTMI_GUARD_END
}

int rtmi_quantize_device(rtmi_scene_t* s, const void* rgba_device, uint64_t npixels, void* rgb_device, void* hip_stream) {
    if (!s) return fail(RTMI_ERR_INVALID, "scene is NULL");
    if (npixels == 0) return RTMI_OK;
    if (!rgba_device || !rgb_device) return fail(RTMI_ERR_INVALID, "NULL argument");
    HIPCHK(hipSetDevice(s->device));
    hipLaunchKernelGGL(k_quantize, dim3((unsigned)(s->num_cu * 8)), dim3(256), 0, (hipStream_t)hip_stream, (uint64_t)npixels,
                       (const float4*)rgba_device, (uint8_t*)rgb_device);
    HIPCHK(hipGetLastError());
    return RTMI_OK;
}

int rtmi_quantize(rtmi_scene_t* s, const float* rgba_host, uint64_t npixels, uint8_t* rgb_host) {
    if (!s) return fail(RTMI_ERR_INVALID, "scene is NULL");
    if (npixels == 0) return RTMI_OK;
    if (!rgba_host || !rgb_host) return fail(RTMI_ERR_INVALID, "NULL argument");
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(s->tile.ensure(npixels));
    HIPCHK(s->qbytes.ensure(npixels * 3));
    HIPCHK(hipMemcpy(s->tile.p, rgba_host, npixels * 16, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_quantize, dim3((unsigned)(s->num_cu * 8)), dim3(256), 0, nullptr, (uint64_t)npixels, s->tile.p, s->qbytes.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(rgb_host, s->qbytes.p, npixels * 3, hipMemcpyDeviceToHost));
    return RTMI_OK;
}

}  // extern "C"
