// trace_pool.hpp — the octree closest-hit kernel with a per-wave RAY POOL in LDS.
//
// Same records, same traversal, same two step kinds (SELECT / LEAF) and the same arithmetic as k_trace_oct
// (trace_oct.hpp; reference: get_object_intersection_for_ray raytrace.rs:909-1010, Triangle::intersects :400-439).
// What changes is who runs a step.  In k_trace_oct a lane OWNS a ray, so when the wave votes for a SELECT step the
// lanes whose ray is in a leaf sit idle (and the other way round): ~57 % of the lanes work in a step.  Here a wave
// keeps P > 64 rays (P = 112 for a depth-10 octree) with their complete state in LDS -- ray, current frame, running
// best, leaf cursor, and the frame stack -- and every step
//   1. counts the pool's rays per kind (one mode byte per ray),
//   2. takes the kind with more rays and COMPACTS up to 64 of them onto the lanes (ballot + mbcnt ranks, the slot
//      list goes through 64 bytes of LDS),
//   3. each lane loads its ray's state from LDS (ds_read_b128), runs the step, writes back what changed.
// Because S + L rays = P >= 2 x 64 x 7/8, the larger kind nearly always fills the wave.  The moves are LDS traffic,
// which has its own issue port; the VALU and vector-memory instructions, which bound the kernel, are spent on full
// waves.  Cost: ~20 KB of LDS per wave, i.e. 8 waves per CU -- the occupancy sweep of k_trace_oct shows that kernel
// saturating at 12-16 waves per CU, 84 % at 8.
//
// Slot layout (32-bit words; stride = 24 + 2 x levels rounded up to a multiple of 4, 44 for a depth-10 octree, so
// that consecutive slots start 4-bank groups apart and ds_read_b128 of consecutive slots is conflict-free):
//    0 ox  1 oy  2 oz  3 ray index | 4 dx 5 dy 6 dz 7 depth of the current frame | 8 1/dx 9 1/dy 10 1/dz 11 frame
//   12 frame best t  13 running best t  14 running best tri|face<<30 (0 = none)  15 next block index
//   16-19 current reference block | 20 leaf best t  21 leaf best tri|face<<30 (0 = none)  22 orig lane 3  23 dir lane 3
//   24.. frame stack: (frame word, best t) per level
// frame word = visited octants (bits 0-7) | O_DONE | O_HAS | record index << 10.
#pragma once

namespace rtmi {

enum : uint32_t { PM_IDLE = 0, PM_SELECT = 1, PM_LEAF = 2, PM_NONE = 3 };

#define RTMI_POOL_MAX 128

template <bool COUNT, bool FAST>
__global__ void __launch_bounds__(64, 2) k_trace_pool(DScene sc, const float4* __restrict__ qo, const float4* __restrict__ qd,
                                                   DCtrl* __restrict__ ctrl, int pass, uint32_t* __restrict__ hit_tf,
                                                   float* __restrict__ hit_t, int refill_min, int xcd_aware, uint32_t P,
                                                   uint32_t stride) {
    extern __shared__ uint32_t lds[];
    const uint32_t lane = threadIdx.x;  // one wave per block
    uint8_t* const modes = reinterpret_cast<uint8_t*>(lds + P * stride);  // RTMI_POOL_MAX bytes
    uint8_t* const list = modes + RTMI_POOL_MAX;                          // 128 bytes
    const uint32_t count = ctrl->count[pass];
    if (blockIdx.x == 0 && lane == 0) atomicAdd(&ctrl->rays, (unsigned long long)count);
    unsigned long long cnt[5] = {0, 0, 0, 0, 0};
    unsigned long long dbg[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // COUNT only: S steps, S lanes, L steps, L lanes, refills, refill lanes, edge steps, edge lanes
    const float root_half = sc.root_half;
    const uint32_t inf_bits = 0x7F800000u;
    const bool two = lane + 64u < P;  // this lane also looks after slot lane + 64

    modes[lane] = PM_IDLE;
    if (two) modes[lane + 64u] = PM_IDLE;
    bool exhausted = false;  // wave-uniform
    const uint32_t nranges = xcd_aware ? 8u : 1u;
    const uint32_t home = xcd_aware == 1 ? (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u) : (blockIdx.x & 7u);
    uint32_t tries = 0;

    for (;;) {
        const uint32_t m0 = modes[lane];
        const uint32_t m1 = two ? (uint32_t)modes[lane + 64u] : (uint32_t)PM_NONE;
        const unsigned long long i0 = __ballot(m0 == PM_IDLE), i1 = __ballot(m1 == PM_IDLE);
        const unsigned long long s0 = __ballot(m0 == PM_SELECT), s1 = __ballot(m1 == PM_SELECT);
        const unsigned long long l0 = __ballot(m0 == PM_LEAF), l1 = __ballot(m1 == PM_LEAF);
        const uint32_t nI = (uint32_t)(__popcll(i0) + __popcll(i1));
        const uint32_t nS = (uint32_t)(__popcll(s0) + __popcll(s1)), nL = (uint32_t)(__popcll(l0) + __popcll(l1));
        if (nS + nL == 0u && exhausted) break;
        if (!exhausted && ((int)nI >= refill_min || nS + nL == 0u)) {
            // ---- refill: idle slots take consecutive queued rays (slot order = queue order, so the samples of a
            //      pixel stay neighbours in the pool)
            if (COUNT && lane == 0) { dbg[4]++; dbg[5] += nI; }
            uint32_t base = 0, hi = 0;
            for (;;) {
                const uint32_t x = (home + tries) % nranges;
                hi = (uint32_t)(((unsigned long long)count * (x + 1)) / nranges);
                const uint32_t lo = (uint32_t)(((unsigned long long)count * x) / nranges);
                if (lane == 0) base = lo + atomicAdd(&ctrl->xhead[pass][x], nI);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base < hi) break;
                if (++tries == nranges) { exhausted = true; break; }
            }
            if (exhausted) continue;
            const uint32_t r0 = __builtin_amdgcn_mbcnt_hi((uint32_t)(i0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)i0, 0u));
            const uint32_t r1 = (uint32_t)__popcll(i0) + __builtin_amdgcn_mbcnt_hi((uint32_t)(i1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)i1, 0u));
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const bool idle = h == 0 ? (m0 == PM_IDLE) : (m1 == PM_IDLE);
                const uint32_t i = base + (h == 0 ? r0 : r1);
                if (idle && i < hi) {
                    const uint32_t slot = lane + 64u * (uint32_t)h;
                    const float4 o = qo[i], d = qd[i];
                    uint4* q = reinterpret_cast<uint4*>(lds + slot * stride);
                    q[0] = make_uint4(__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), i);
                    q[1] = make_uint4(__float_as_uint(d.x), __float_as_uint(d.y), __float_as_uint(d.z), 0u);  // depth 0: the root's frame
                    // raytrace.rs:206-208; the root box itself is never slab-tested (raytrace.rs:1272)
                    q[2] = make_uint4(__float_as_uint(1.f / d.x), __float_as_uint(1.f / d.y), __float_as_uint(1.f / d.z), 0u);
                    q[3] = make_uint4(0u, 0u, 0u, 0u);
                    q[5] = make_uint4(0u, 0u, __float_as_uint(o.w), __float_as_uint(d.w));
                    modes[slot] = PM_SELECT;
                }
            }
            continue;
        }
        // ---- pick the step kind and compact its rays onto the lanes
        const bool doS = nS >= nL;
        const unsigned long long b0 = doS ? s0 : l0, b1 = doS ? s1 : l1;
        const uint32_t n0 = (uint32_t)__popcll(b0), tot = n0 + (uint32_t)__popcll(b1);
        const uint32_t want = doS ? PM_SELECT : PM_LEAF;
        if (m0 == want) list[__builtin_amdgcn_mbcnt_hi((uint32_t)(b0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b0, 0u))] = (uint8_t)lane;
        if (m1 == want) list[n0 + __builtin_amdgcn_mbcnt_hi((uint32_t)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b1, 0u))] = (uint8_t)(lane + 64u);
        const uint32_t nact = tot < 64u ? tot : 64u;
        if (COUNT && lane == 0) { if (doS) { dbg[0]++; dbg[1] += nact; } else { dbg[2]++; dbg[3] += nact; } }
        if (lane < nact) {
            const uint32_t slot = list[lane];
            uint32_t* const st = lds + slot * stride;
            uint4* const q = reinterpret_cast<uint4*>(st);
            const uint4 q0 = q[0], q1 = q[1];
            RayK r;
            r.ox = __uint_as_float(q0.x); r.oy = __uint_as_float(q0.y); r.oz = __uint_as_float(q0.z);
            r.dx = __uint_as_float(q1.x); r.dy = __uint_as_float(q1.y); r.dz = __uint_as_float(q1.z);
            if (doS) {
                // ================================================= SELECT step
                const uint4 q2 = q[2];
                r.ix = __uint_as_float(q2.x); r.iy = __uint_as_float(q2.y); r.iz = __uint_as_float(q2.z);
                uint32_t fwn = q2.w;              // frame word
                float ft = __uint_as_float(st[12]);
                uint32_t lvl = q1.w;
                bool finished = false;
                // pop finished frames; the frames of depth 0 .. lvl-1 are stack entries 0 .. lvl-1
                while (fwn & O_DONE) {
                    if (lvl == 0u) {
                        const uint32_t gtf = st[14];
                        hit_tf[q0.w] = gtf;
                        hit_t[q0.w] = gtf ? __uint_as_float(st[13]) : 0.f;
                        modes[slot] = PM_IDLE;
                        finished = true;
                        break;
                    }
                    const bool have = (fwn & O_HAS) != 0u;
                    const float ct = ft;
                    lvl--;
                    const uint2 fr = *reinterpret_cast<const uint2*>(st + 24u + 2u * lvl);
                    fwn = fr.x;
                    ft = __uint_as_float(fr.y);
                    if (have) {  // fold step of raytrace.rs:949-1007: first hit is taken, later ones replace iff strictly closer
                        if (!(fwn & O_HAS) || ct < ft) ft = ct;
                        fwn |= O_HAS;
                    }
                }
                if (!finished) {
                    const uint4* fp = sc.fnodes + 2 * (size_t)(fwn >> 10);
                    const uint4 n0q = fp[0], n1q = fp[1];
                    const float cx = __uint_as_float(n0q.x), cy = __uint_as_float(n0q.y), cz = __uint_as_float(n0q.z);
                    const float hc = ldexpf(root_half, -(int)(lvl + 1u));  // half edge of the children (depth lvl + 1)
                    if (COUNT && (fwn & 0xFFu) == 0u) { cnt[0] += __popc(n0q.w & 0xFFu); cnt[3]++; }  // first visit: collides() on every child
                    const float xl = cx + (-hc), xh = cx + hc, yl = cy + (-hc), yh = cy + hc, zl = cz + (-hc), zh = cz + hc;
                    // see trace_oct.hpp for the derivation of this form of BoundingBox::collides (raytrace.rs:860-907)
                    const float bx = fabsf(r.ix) * hc, by = fabsf(r.iy) * hc, bz = fabsf(r.iz) * hc;
                    const float axl = (xl - r.ox) * r.ix, axh = (xh - r.ox) * r.ix;
                    const float ayl = (yl - r.oy) * r.iy, ayh = (yh - r.oy) * r.iy;
                    const float azl = (zl - r.oz) * r.iz, azh = (zh - r.oz) * r.iz;
                    float nx[2] = {axl - bx, axh - bx}, fx[2] = {axl + bx, axh + bx};
                    float ny[2] = {ayl - by, ayh - by}, fy[2] = {ayl + by, ayh + by};
                    float nz[2] = {azl - bz, azh - bz}, fz[2] = {azl + bz, azh + bz};
                    if (!(r.dx != 0.f) | !(r.dy != 0.f) | !(r.dz != 0.f)) {
                        if (!(r.dx != 0.f)) { nx[0] = nx[1] = -FLT_MAX; fx[0] = fx[1] = FLT_MAX; }
                        if (!(r.dy != 0.f)) { ny[0] = ny[1] = fy[0] = fy[1] = __uint_as_float(0x7FC00000u); }
                        if (!(r.dz != 0.f)) { nz[0] = nz[1] = fz[0] = fz[1] = __uint_as_float(0x7FC00000u); }
                    }
                    float tmv[8];
                    uint32_t hits = 0;
#pragma unroll
                    for (int o = 0; o < 8; o++) {
                        tmv[o] = max3f(nx[o & 1], ny[(o >> 1) & 1], nz[o >> 2]);
                        const float tmax = min3f(fx[o & 1], fy[(o >> 1) & 1], fz[o >> 2]);
                        const bool c = FAST ? ((tmv[o] < tmax) & !(tmax < 0.f)) : (tmv[o] < tmax);
                        hits |= c ? (1u << o) : 0u;
                    }
                    const uint32_t cand = hits & n0q.w & ~fwn & 0xFFu;  // colliding, present, not visited yet
                    const uint32_t nh = (uint32_t)__popc(cand);
                    float tm[8];
#pragma unroll
                    for (int o = 0; o < 8; o++) {
                        const uint32_t ex = (uint32_t)__builtin_amdgcn_sbfe((int)cand, o, 1);
                        tm[o] = __uint_as_float((__float_as_uint(tmv[o]) & ex) | (inf_bits & ~ex));
                    }
                    const float m1v = min3f(min3f(tm[0], tm[1], tm[2]), min3f(tm[3], tm[4], tm[5]), fminf(tm[6], tm[7]));
                    bool ok = nh != 0u;
                    if (fwn & O_HAS) ok = ok & (m1v < ft);   // raytrace.rs:965
                    else ok = ok & (m1v != FLT_MAX);         // raytrace.rs:986
                    uint32_t bit = 0u;
#pragma unroll
                    for (int o = 7; o >= 0; o--) bit = (tm[o] == m1v) ? (1u << o) : bit;
                    if (!ok) {
                        fwn |= O_DONE;
                    } else {
                        fwn |= bit | (nh == 1u ? O_DONE : 0u);
                        const uint32_t leafmask = (n0q.w >> 8) & 0xFFu;
                        if (leafmask & bit) {
                            const uint32_t oct = (uint32_t)__ffs((int)bit) - 1u;
                            uint32_t lblock;
                            if (n0q.w & FN_WIDE) lblock = sc.wlinks[(size_t)n1q.y * 8u + oct];
                            else lblock = n1q.y + __builtin_amdgcn_ubfe(oct < 4u ? n1q.z : n1q.w, (oct & 3u) * 8u, 8u);
                            const uint4 blk = sc.oblocks[lblock];
                            st[15] = lblock;
                            q[4] = blk;
                            *reinterpret_cast<uint2*>(st + 20) = make_uint2(0u, 0u);
                            if (COUNT) cnt[4]++;
                            modes[slot] = PM_LEAF;
                        } else {
                            *reinterpret_cast<uint2*>(st + 24u + 2u * lvl) = make_uint2(fwn, __float_as_uint(ft));
                            lvl++;
                            fwn = (n1q.x + (uint32_t)__popc((n0q.w & ~leafmask & 0xFFu) & (bit - 1u))) << 10;
                            ft = 0.f;
                        }
                    }
                    st[7] = lvl;
                    st[11] = fwn;
                    st[12] = __float_as_uint(ft);
                }
            } else {
                // ================================================= LEAF step: one block of <= 4 references
                const uint4 q5 = q[5];
                r.ow = __uint_as_float(q5.z); r.dw = __uint_as_float(q5.w);
                r.qn = 0.f * (0.f - r.ow);
                r.qd = 0.f * r.dw;
                float lt = __uint_as_float(q5.x);
                uint32_t ltf = q5.y;
                bool lhave = ltf != 0u;
                const uint4 blk = q[4];
                uint32_t lblock = st[15];
                const uint32_t ids[4] = {blk.x, blk.y, blk.z, blk.w & 0x7FFFFFFFu};
                float4 p0[4], p1[4];
#pragma unroll
                for (int k = 0; k < 4; k++) { p0[k] = sc.tplane[2 * ids[k]]; p1[k] = sc.tplane[2 * ids[k] + 1]; }
                const bool more = blk.w != 0u && !(blk.w >> 31);
                uint4 nblk = make_uint4(0u, 0u, 0u, 0u);
                if (more) { lblock++; nblk = sc.oblocks[lblock]; }  // the next block, parked in the slot
                uint32_t ptri = 0u, pback = 0u;
                float pt = 0.f, pix = 0.f, piy = 0.f, piz = 0.f, pz = 0.f;
                auto resolve = [&]() {
                    if (COUNT) { cnt[2]++; const unsigned long long em = __ballot(true); if (lane == (uint32_t)(__ffsll((long long)em) - 1)) { dbg[6]++; dbg[7] += __popcll(em); } }
                    const float4 e0 = sc.tedge[4 * ptri], e1 = sc.tedge[4 * ptri + 1], e2 = sc.tedge[4 * ptri + 2], e3 = sc.tedge[4 * ptri + 3];
                    const float d0 = ((pix * e0.x + piy * e0.y) + piz * e0.z) + pz;
                    const float d1 = ((pix * e1.x + piy * e1.y) + piz * e1.z) + pz;
                    const float d2 = ((pix * e2.x + piy * e2.y) + piz * e2.z) + pz;
                    const bool inside = !(d0 > e0.w) & !(d1 > e1.w) & !(d2 > e2.w);
                    const bool edge = (d0 > e3.x) | (d1 > e3.y) | (d2 > e3.z);
                    const uint32_t face = pback | (edge ? 2u : 0u);
                    const bool take = inside & (!lhave | (pt < lt));  // raytrace.rs:1028-1038
                    lt = take ? pt : lt;
                    ltf = take ? (ptri | (face << 30)) : ltf;
                    lhave = lhave | inside;
                };
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float ax = p0[k].x - r.ox, ay = p0[k].y - r.oy, az = p0[k].z - r.oz;
                    const float num = (((0.f + p1[k].x * ax) + p1[k].y * ay) + p1[k].z * az) + r.qn;
                    const float den = (((0.f + p1[k].x * r.dx) + p1[k].y * r.dy) + p1[k].z * r.dz) + r.qd;
                    const float t = num / den;
                    const float px = r.dx * t + r.ox, py = r.dy * t + r.oy, pz_ = r.dz * t + r.oz, pw = r.dw * t + r.ow;
                    const float ix = px - p0[k].x, iy = py - p0[k].y, iz = pz_ - p0[k].z;
                    const float l2 = ((ix * ix + iy * iy) + iz * iz) + pw * pw;
                    const bool real = ids[k] != 0u;
                    if (COUNT) cnt[1] += real ? 1u : 0u;
                    const bool c = real & !(t < 0.f) & !(l2 > p0[k].w);
                    if (c & (ptri != 0u)) resolve();  // second candidate of this ray in one block: rare
                    ptri = c ? ids[k] : ptri;
                    pt = c ? t : pt;
                    pix = c ? ix : pix; piy = c ? iy : piy; piz = c ? iz : piz;
                    pz = c ? pw * 0.f : pz;
                    pback = c ? (den > 0.f ? 1u : 0u) : pback;
                }
                if (ptri != 0u) resolve();
                if (more) {
                    st[15] = lblock;
                    q[4] = nblk;
                    *reinterpret_cast<uint2*>(st + 20) = make_uint2(__float_as_uint(lt), ltf);
                } else {
                    if (lhave) {
                        uint32_t fwn = st[11];
                        const uint4 q3 = q[3];  // ft, gt, gtf, -
                        float ft = __uint_as_float(q3.x), gt = __uint_as_float(q3.y);
                        uint32_t gtf = q3.z;
                        if (!(fwn & O_HAS) || lt < ft) ft = lt;
                        fwn |= O_HAS;
                        if (gtf == 0u || lt < gt) { gt = lt; gtf = ltf; }
                        st[11] = fwn;
                        st[12] = __float_as_uint(ft);
                        st[13] = __float_as_uint(gt);
                        st[14] = gtf;
                    }
                    modes[slot] = PM_SELECT;
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
