// trace_oct.hpp — the closest-hit kernel for trees that are exact octrees
// (what build_bounding_box produces, raytrace_lib/src/raytrace.rs:795-845;
// checked box by box in rtmi_scene_create, otherwise the generic kernel of
// rtmi_device.hip runs).  Included by rtmi_device.hip.
//
// Same traversal order and skip rule as get_object_intersection_for_ray
// (raytrace.rs:909-1010), restructured for a 64-lane wavefront:
//
//  * one lane = one ray, lanes are PERSISTENT: a lane that finishes pulls the
//    next queued ray (wave ballot + prefix sum over the idle lanes, one atomic
//    per refill), so a wave stays full until the queue is empty;
//  * a lane is in one of two working states, SELECT (at a box: pop finished
//    frames, pick the next child in sorted order, apply the skip rule, expand
//    it or enter it as a leaf) or LEAF (scan one 16-B block of triangle
//    references).  Each iteration the wave runs the step that the majority of
//    its lanes wait for, instead of serialising nested per-lane loops;
//  * child boxes are implicit: the 8 children of a box share 2 candidate
//    planes per axis (centre +- half/2), so one expansion is 6 slab-plane pairs
//    + 8 max3/min3, with no child records loaded.  The child centres are
//    recomputed with the builder's own expression (orig + (+-newlen2),
//    raytrace.rs:816-824) and were verified bitwise at scene creation;
//  * the stack of frames lives in LDS, [level][word][lane]: a lane only ever
//    touches its own bank.
//
// Records (HBM, served from L2 / Infinity Cache):
//   onodes  : float4 per box  (cx, cy, cz, link)
//             link = low 24 bits | child_mask << 24.  Inner: low = index of the
//             first child, mask = octants present (children stored in octant
//             order).  Leaf: mask = 0, low = index of its first reference block.
//   oblocks : uint4 blocks of triangle indices of a leaf, in list order.  The list ends at the first index 0
//             (the sentinel triangle is never in a tree, raytrace.rs:791) or after a full block whose 4th
//             index has bit 31 set.
#pragma once

namespace rtmi {

enum : uint32_t { M_IDLE = 0, M_SELECT = 1, M_LEAF = 2 };

// frame words: w0 = first_child | mask << 24 ; w1 = order(24) | count << 24 | F_HAS | F_ANYMAX ; t = best hit
// time inside this frame's subtree (what the sibling-local skip rule compares against).
// Which triangle that was is NOT kept per frame: the ray keeps one running best (t, tri) over its leaf results in
// visiting order.  Merging with strict `<` is associative as long as no leaf result has a NaN time, so the running
// best equals the reference's nested per-box results; a ray that meets a NaN leaf result (a 0/0 plane test that comes
// first in a leaf) is re-traced by the generic kernel, which merges box by box.  3 words per level instead of 4.
struct OFrame { uint32_t w0, w1; float t; };

#define RTMI_REFILL_MIN 16

// v_max3_f32 / v_min3_f32: max(max(a,b),c) with fmaxf's NaN rule (a NaN operand is ignored), one instruction
// instead of the two v_max + canonicalising moves hipcc emits for nested fmaxf.
__device__ inline float max3f(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ inline float min3f(float a, float b, float c) {
    float d;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

template <bool COUNT>
__device__ inline OFrame expand_oct(float cx, float cy, float cz, uint32_t link, float hc, const RayK& r,
                                    unsigned long long* cnt, const bool fast) {
    const uint32_t mask = link >> 24;
    if (COUNT) { cnt[0] += __popc(mask); cnt[3]++; }
    // candidate planes per axis (child centres: builder's orig.add(off_vec))
    const float xl = cx + (-hc), xh = cx + hc, yl = cy + (-hc), yh = cy + hc, zl = cz + (-hc), zh = cz + hc;
    // t1s = tmp1 - tmp2, t2s = tmp1 + tmp2 with tmp2 = inv_dir * len2 (raytrace.rs:866-870); near/far swap when
    // inv_dir <= 0.  With b = (inv_dir > 0 ? inv_dir : -inv_dir) * len2 the pair is (a - b, a + b) in both cases,
    // bit for bit: negation is exact and x + y == x - (-y).
    const bool px = r.ix > 0.f, py = r.iy > 0.f, pz = r.iz > 0.f;
    const float bx = (px ? r.ix : -r.ix) * hc, by = (py ? r.iy : -r.iy) * hc, bz = (pz ? r.iz : -r.iz) * hc;
    const float axl = (xl - r.ox) * r.ix, axh = (xh - r.ox) * r.ix;
    const float ayl = (yl - r.oy) * r.iy, ayh = (yh - r.oy) * r.iy;
    const float azl = (zl - r.oz) * r.iz, azh = (zh - r.oz) * r.iz;
    float nx[2] = {axl - bx, axh - bx}, fx[2] = {axl + bx, axh + bx};
    float ny[2] = {ayl - by, ayh - by}, fy[2] = {ayl + by, ayh + by};
    float nz[2] = {azl - bz, azh - bz}, fz[2] = {azl + bz, azh + bz};
    // a zero direction component skips its slab (raytrace.rs:872, :882, :892): axis 0 then leaves the
    // initial (-MAX, MAX); for axes 1, 2 a NaN operand makes max3/min3 return the running value.  Rare:
    // whole waves skip this block.
    if (!(r.dx != 0.f) | !(r.dy != 0.f) | !(r.dz != 0.f)) {
        if (!(r.dx != 0.f)) { nx[0] = nx[1] = -FLT_MAX; fx[0] = fx[1] = FLT_MAX; }
        if (!(r.dy != 0.f)) { ny[0] = ny[1] = fy[0] = fy[1] = __uint_as_float(0x7FC00000u); }
        if (!(r.dz != 0.f)) { nz[0] = nz[1] = fz[0] = fz[1] = __uint_as_float(0x7FC00000u); }
    }
    // Conservative any-tmin-is-MAX flag (raytrace.rs:986 only matters when a colliding tmin == f32::MAX): a
    // child's tmin is the max of three of these six values, so it can only be MAX if one of them is >= MAX.  The
    // flag merely enables the exact re-check at selection time, so over-approximating is harmless.
    const bool anymax = max3f(max3f(nx[0], nx[1], ny[0]), max3f(ny[1], nz[0], nz[1]), -FLT_MAX) >= FLT_MAX;
    float tm[8];
    uint32_t nh = 0;
    const uint32_t inf_bits = 0x7F800000u;
#pragma unroll
    for (int o = 0; o < 8; o++) {
        const float tmin = max3f(nx[o & 1], ny[(o >> 1) & 1], nz[o >> 2]);
        const float tmax = min3f(fx[o & 1], fy[(o >> 1) & 1], fz[o >> 2]);
        // RTMI_OPT_FAST (off by default, NOT the reference's traversal): ignore boxes that lie entirely behind the
        // ray origin.  The reference visits them (collides() has no `tmax > 0` test, raytrace.rs:902) although only
        // triangles that stick out of such a box towards the front can be hit through it.
        const uint32_t sel = ((tmin < tmax) & (!fast | !(tmax < 0.f))) ? __float_as_uint(tmin) : inf_bits;
        // keep it only when the octant exists: exists_o is bit o of the child mask, spread to a full-width mask
        const uint32_t ex = (uint32_t)__builtin_amdgcn_sbfe((int)mask, o, 1);
        tm[o] = __uint_as_float((sel & ex) | (inf_bits & ~ex));  // a colliding tmin is never NaN and never +inf
        nh += (tm[o] < INFINITY) ? 1u : 0u;
    }
    // Stable ascending order of the colliding children (insertion sort of raytrace.rs:941-947) as a
    // rank: child i goes after every j < i with tm[j] <= tm[i] and every k > i with tm[k] < tm[i].
    // One compare per pair (no NaN among tm): c = tm[j] <= tm[k] puts j before k, !c puts k before j.
    uint32_t A[8] = {0, 0, 0, 0, 0, 0, 0, 0}, B[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 8; j++) {
#pragma unroll
        for (int k = j + 1; k < 8; k++) {
            const uint32_t c = (tm[j] <= tm[k]) ? 1u : 0u;
            A[k] += c;
            B[j] += c;
        }
    }
    // non-colliding children (tm = +inf) rank last; their slots lie beyond `nh` and are never read
    uint32_t order = 0;
#pragma unroll
    for (int i = 1; i < 8; i++) {
        const uint32_t rank3 = (A[i] - B[i]) * 3u + (uint32_t)(3 * (7 - i));
        order |= (uint32_t)i << rank3;
    }
    OFrame f;
    f.w0 = link;
    f.w1 = order | (nh << 24) | (anymax ? F_ANYMAX : 0u);
    f.t = 0.f;
    return f;
}

__device__ inline void omerge(OFrame& f, bool have, float t) {
    if (have) {
        if (!(f.w1 & F_HAS) || t < f.t) f.t = t;
        f.w1 |= F_HAS;
    }
}

template <bool COUNT>
__global__ void __launch_bounds__(64, 6) k_trace_oct(DScene sc, const float4* __restrict__ qo, const float4* __restrict__ qd,
                                                  DCtrl* __restrict__ ctrl, int pass, uint32_t* __restrict__ hit_tf,
                                                  float* __restrict__ hit_t, uint32_t* __restrict__ redo, int refill_min, int xcd_aware, int fast) {
    extern __shared__ uint32_t lds[];
    const int lane = threadIdx.x;  // one wave per block
    constexpr int NT = 64;
    const uint32_t count = ctrl->count[pass];
    if (blockIdx.x == 0 && lane == 0) atomicAdd(&ctrl->rays, (unsigned long long)count);
    unsigned long long cnt[5] = {0, 0, 0, 0, 0};
    unsigned long long dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // COUNT only: S steps, S lanes, L steps, L lanes, refills, refill lanes, edge blocks, edge lanes
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const float root_half = sc.root_half;

    uint32_t mode = M_IDLE;
    bool exhausted = false;  // wave-uniform
    // HW_REG_XCC_ID (id 20, bits 3:0): which XCD this wave runs on; only a locality hint
    const uint32_t nranges = xcd_aware ? 8u : 1u;
    const uint32_t home = xcd_aware == 1 ? (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u) : (blockIdx.x & 7u);
    uint32_t tries = 0;      // ranges this wave has seen exhausted (wave-uniform)
    RayK r = make_rayk(make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 1.f, 0.f));
    uint32_t ridx = 0;
    OFrame cur{0, 0, 0.f};
    int lvl = -1;  // depth of the current frame's box; -1 = the virtual frame above the root
    bool ghave = false, gnan = false;  // running best over the ray's leaf results
    float gt = 0.f;
    uint32_t gtf = 0;
    uint4 blk = make_uint4(0, 0, 0, 0);  // current reference block of the leaf being scanned
    uint32_t lblock = 0;
    bool lhave = false;
    float lt = 0.f;
    uint32_t ltf = 0;

    for (;;) {
        const unsigned long long m_idle = __ballot(mode == M_IDLE);
        if (m_idle == ~0ull && exhausted) break;
        if (!exhausted && (__popcll(m_idle) >= refill_min || m_idle == ~0ull)) {
            // ---- refill: idle lanes take consecutive queued rays
            const uint32_t n = (uint32_t)__popcll(m_idle);
            if (COUNT && lane == 0) { dbg[4]++; dbg[5] += n; }
            // XCD-aware work fetch: the queue is cut into 8 contiguous ranges, one per XCD (each XCD has its own
            // L2, so the waves of an XCD walk one image region and share its boxes/triangles there).  A wave pulls
            // from the range of the XCD it runs on and moves to the next range when that one is exhausted, so the
            // ranges only steer locality, never correctness or balance.
            uint32_t base = 0, hi = 0;
            for (;;) {
                const uint32_t x = (home + tries) % nranges;
                hi = (uint32_t)(((unsigned long long)count * (x + 1)) / nranges);
                const uint32_t lo = (uint32_t)(((unsigned long long)count * x) / nranges);
                if (lane == 0) base = lo + atomicAdd(&ctrl->xhead[pass][x], n);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base < hi) break;
                if (++tries == nranges) { exhausted = true; break; }
            }
            if (exhausted) continue;
            if (mode == M_IDLE) {
                const uint32_t i = base + (uint32_t)__popcll(m_idle & lt_mask);
                if (i < hi) {
                    ridx = i;
                    r = make_rayk(qo[i], qd[i]);
                    // virtual frame whose only child is the root box (index 0): the root itself is never
                    // slab-tested (raytrace.rs:1272 calls get_object_intersection_for_ray on it directly)
                    cur.w0 = 0u | (1u << 24);
                    cur.w1 = 0u | (1u << 24);
                    cur.t = 0.f;
                    lvl = -1;
                    ghave = false; gnan = false; gt = 0.f; gtf = 0;
                    mode = M_SELECT;
                }
            }
            continue;
        }
        const int nS = __popcll(__ballot(mode == M_SELECT));
        const int nL = __popcll(__ballot(mode == M_LEAF));
        if (COUNT && lane == 0) { if (nS >= nL) { dbg[0]++; dbg[1] += nS; } else { dbg[2]++; dbg[3] += nL; } }
        if (nS >= nL) {  // majority vote; hysteresis (stay in a phase until its lanes fall below 1/2..1/8 of the other's) measured 1-7 % slower
            // ================================================= SELECT step
            if (mode == M_SELECT) {
                // pop finished frames; the frames of depth 0 .. lvl-1 are in LDS levels 0 .. lvl-1
                while (F_COUNT(cur.w1) == 0) {
                    if (lvl <= 0) {
                        hit_tf[ridx] = ghave ? gtf : 0u;
                        hit_t[ridx] = ghave ? gt : 0.f;
                        if (gnan) redo[atomicAdd(&ctrl->redo[pass], 1u)] = ridx;  // rare: exact box-by-box merge needed
                        mode = M_IDLE;
                        break;
                    }
                    const bool have = (cur.w1 & F_HAS) != 0;
                    const float ct = cur.t;
                    lvl--;
                    const uint32_t* fr = lds + lvl * 3 * NT + lane;
                    cur.w0 = fr[0];
                    cur.w1 = fr[NT];
                    cur.t = __uint_as_float(fr[2 * NT]);
                    omerge(cur, have, ct);
                }
                if (mode == M_SELECT) {
                    const uint32_t o = cur.w1 & 7u;
                    cur.w1 = ((cur.w1 & 0x00FFFFFFu) >> 3) | ((cur.w1 & 0xFF000000u) - (1u << 24));
                    const uint32_t cidx = (cur.w0 & 0x00FFFFFFu) + (uint32_t)__popc((cur.w0 >> 24) & ((1u << o) - 1u));
                    const float4 rec = sc.onodes[cidx];
                    const float hc = ldexpf(root_half, -(lvl + 1));  // half edge of the child (depth lvl + 1)
                    bool go = true;
                    if (cur.w1 & (F_HAS | F_ANYMAX)) {
                        float tmin;
                        collides(rec.x, rec.y, rec.z, hc, r, tmin);
                        if (cur.w1 & F_HAS) {
                            if (!(tmin < cur.t)) { cur.w1 &= ~(15u << 24); go = false; }  // raytrace.rs:965; later children are farther
                        } else if (tmin == FLT_MAX) go = false;                           // raytrace.rs:986
                    }
                    if (go) {
                        const uint32_t link = __float_as_uint(rec.w);
                        if ((link >> 24) == 0u) {
                            lblock = link;
                            blk = sc.oblocks[lblock];
                            lhave = false; lt = 0.f; ltf = 0;
                            if (COUNT) cnt[4]++;
                            mode = M_LEAF;
                        } else {
                            if (lvl >= 0) {  // the virtual frame needs no slot: it has nothing left to do
                                uint32_t* fr = lds + lvl * 3 * NT + lane;
                                fr[0] = cur.w0;
                                fr[NT] = cur.w1;
                                fr[2 * NT] = __float_as_uint(cur.t);
                            }
                            lvl++;
                            cur = expand_oct<COUNT>(rec.x, rec.y, rec.z, link, ldexpf(root_half, -(lvl + 1)), r, cnt, fast != 0);
                        }
                    }
                }
            }
        } else {
            // ================================================= LEAF step: one block of <= 4 references
            if (mode == M_LEAF) {
                const uint32_t ids[4] = {blk.x, blk.y, blk.z, blk.w & 0x7FFFFFFFu};
                float4 p0[4], p1[4];
#pragma unroll
                for (int k = 0; k < 4; k++) { p0[k] = sc.tplane[2 * ids[k]]; p1[k] = sc.tplane[2 * ids[k] + 1]; }
                const bool more = blk.w != 0u && !(blk.w >> 31);  // bit 31 of the 4th index: this full block is the last
                if (more) { lblock++; blk = sc.oblocks[lblock]; }  // prefetch the next block
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (ids[k] != 0u) {
                        // Triangle::intersects (raytrace.rs:400-439), see tri_test() for the lane-3 terms
                        const float ax = p0[k].x - r.ox, ay = p0[k].y - r.oy, az = p0[k].z - r.oz;
                        const float num = (((0.f + p1[k].x * ax) + p1[k].y * ay) + p1[k].z * az) + r.qn;
                        const float den = (((0.f + p1[k].x * r.dx) + p1[k].y * r.dy) + p1[k].z * r.dz) + r.qd;
                        const float t = num / den;
                        if (COUNT) cnt[1]++;
                        if (!(t < 0.f)) {
                            const float px = r.dx * t + r.ox, py = r.dy * t + r.oy, pz = r.dz * t + r.oz, pw = r.dw * t + r.ow;
                            const float ix = px - p0[k].x, iy = py - p0[k].y, iz = pz - p0[k].z;
                            const float l2 = ((ix * ix + iy * iy) + iz * iz) + pw * pw;
                            if (!(l2 > p0[k].w)) {
                                if (COUNT) { cnt[2]++; const unsigned long long em = __ballot(true); if (lane == __ffsll((long long)em) - 1) { dbg[6]++; dbg[7] += __popcll(em); } }
                                const uint32_t tri = ids[k];
                                // all four edge records are requested together and every comparison is
                                // evaluated (no short-circuit): one memory round trip instead of three
                                const float4 e0 = sc.tedge[4 * tri], e1 = sc.tedge[4 * tri + 1], e2 = sc.tedge[4 * tri + 2], e3 = sc.tedge[4 * tri + 3];
                                const float z = pw * 0.f;
                                const float d0 = ((ix * e0.x + iy * e0.y) + iz * e0.z) + z;
                                const float d1 = ((ix * e1.x + iy * e1.y) + iz * e1.z) + z;
                                const float d2 = ((ix * e2.x + iy * e2.y) + iz * e2.z) + z;
                                const bool inside = !(d0 > e0.w) & !(d1 > e1.w) & !(d2 > e2.w);
                                const bool edge = (d0 > e3.x) | (d1 > e3.y) | (d2 > e3.z);
                                const uint32_t face = (den > 0.f ? 1u : 0u) | (edge ? 2u : 0u);
                                const bool take = inside & (!lhave | (t < lt));  // raytrace.rs:1028-1038
                                lt = take ? t : lt;
                                ltf = take ? (tri | (face << 30)) : ltf;
                                lhave = lhave | inside;
                            }
                        }
                    }
                }
                if (!more) {
                    omerge(cur, lhave, lt);
                    if (lhave) {
                        if (!ghave || lt < gt) { gt = lt; gtf = ltf; }
                        ghave = true;
                        gnan |= (lt != lt);
                    }
                    mode = M_SELECT;
                }
            }
        }
    }
    if (COUNT) {
#pragma unroll
        for (int k = 0; k < 5; k++)
            if (cnt[k]) atomicAdd(&ctrl->counters[k], cnt[k]);
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (dbg[k]) atomicAdd(&ctrl->dbg[k], dbg[k]);
    }
}

}  // namespace rtmi
