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
//  * a lane is in one of two working states, SELECT (at an inner box: pop
//    finished frames, pick the next child) or LEAF (scan one 16-B block of
//    triangle references).  Each iteration the wave runs the step that the
//    majority of its lanes wait for (weighted 3 : 2 towards SELECT, whose
//    step is the cheaper one), instead of serialising nested loops;
//  * LAZY CHILD SELECTION instead of a sort.  The reference sorts the <= 8
//    colliding children by tmin (stable insertion sort, raytrace.rs:941-947)
//    and folds over them.  Visiting "the smallest tmin not visited yet, lowest
//    index on ties" one child at a time is the same order, and a ray enters
//    only ~1.4 children of a box on average, so the kernel never sorts: every
//    SELECT step recomputes the 8 implicit child slabs of the current box
//    (6 plane pairs + 8 max3/min3, nothing loaded but the box's own 32-B
//    record), masks out absent and already visited children, takes the
//    minimum and applies the skip rule to it.  Half the VALU work of the
//    previous sort-then-recheck form (rank sort 105 + recheck 25 per child);
//  * child boxes are implicit: the 8 children of a box share 2 candidate
//    planes per axis (centre +- half/2).  The child centres are recomputed
//    with the builder's own expression (orig + (+-newlen2), raytrace.rs:816-824)
//    and were verified bitwise at scene creation;
//  * the Triangle::intersects edge part (3 half-plane dots, 4 more records)
//    is needed by ~4 % of the plane tests.  It is DEFERRED: the plane tests of
//    a block run branch-free for all lanes, a lane remembers its candidate and
//    the edge part runs once per step for all lanes that have one;
//  * the stack of frames lives in LDS, [level][word][lane], 2 words per frame
//    (record index | flags << 22, best t): a lane only ever touches its own
//    bank;
//  * nothing but the ray, the current frame, the running best and the leaf
//    cursor is carried from step to step, and the loop has ONE back edge:
//    hipcc copies every loop-carried register that is written inside nested
//    divergent branches into a temporary and back ("the
//    instruction diet").
//
// Records (HBM, served from L2 / Infinity Cache):
//   fnodes  : 2 x uint4 (32 B) per INNER box: (cx, cy, cz, present mask | leaf mask << 8 | FN_WIDE)
//             (first inner child's record, first block of the first leaf child, 8 x u8 block offsets of the leaf
//             children).  Children of a box are stored together: the inner ones as consecutive records in octant
//             order, the leaf ones as consecutive runs of reference blocks in octant order, so a child's record /
//             first block follows from the masks with a popcount / a byte extract.  Leaves have no record.  A box
//             whose leaf children hold more than 255 blocks (FN_WIDE, rare) keeps 8 explicit 32-bit block indices in
//             `wlinks` instead (one more dependent load).
//   oblocks : uint4 blocks of triangle indices of a leaf, in list order.  The list ends at the first index 0
//             (the sentinel triangle is never in a tree, raytrace.rs:791) or after a full block whose 4th
//             index has bit 31 set.
#pragma once

namespace rtmi {

enum : uint32_t { M_IDLE = 0, M_SELECT = 1, M_LEAF = 2, M_SHADE = 3 };

// What a lane does when its ray's root frame is finished -- the three kernels share one walk (oct_walk):
//   W_TRACE    k_trace_oct     rays come from a queue, the closest hit goes to hit_tf / hit_t (one launch per bounce pass
//                              of the per-pass pipeline; rtmi_trace)
//   W_PRIMARY  k_path_primary  pixel_ray() is evaluated by the lane that takes the path (no ray queue for primary rays),
//                              the finished ray is shaded IN the kernel (color_ray, shade.hpp): terminal paths write their
//                              sample colour, bounce rays are compacted into the bounce queue (ballot + prefix sum, one
//                              atomic per wave).  Whole-wave refills keep the samples of a pixel in lockstep.
//   W_BOUNCE   k_path_bounce   ONE persistent launch for every bounce of every path: a lane pulls a bounce ray, traces
//                              it, shades it in place and -- when the path goes on -- re-seeds ITSELF with the next bounce
//                              ray; when the path ends it folds the surface stack into the sample colour and pulls the
//                              next queued ray.  The recursion project_ray -> color_ray -> project_ray
//                              (raytrace.rs:1233-1251, :1256-1295) without pass boundaries: no per-pass queues, no
//                              per-pass tails of the persistent waves, no hit records in memory.
// Shading is a third step kind ("exchange"): finished lanes wait in M_SHADE until `refill_min` lanes are finished or idle
// (or nothing else is left to do), then they are shaded together and, in the same step, every lane without a ray takes
// one from the queue.  The arithmetic per path is the per-pass pipeline's (same device functions), so the image is
// bit-identical; only which lane evaluates it, and when, differs.
//   W_SLOW     k_path_slow     W_BOUNCE for the slow-path queue (SlowQ, rtmi_device.hip: rays with an exactly-zero direction
//                              component, ~150 x the work of an ordinary ray): one path per WAVE at a time (lane 0), traced
//                              and shaded to its end, so that such a ray runs at the speed of a lone lane while the ordinary
//                              passes go on beside it on their own stream.
enum : int { W_TRACE = 0, W_PRIMARY = 1, W_BOUNCE = 2, W_SLOW = 3 };

struct OctArgs {
    // W_TRACE
    const float4* qo; const float4* qd; uint32_t* hit_tf; float* hit_t; int pass;
    // W_PRIMARY / W_BOUNCE
    DView v; uint64_t seed; uint32_t pix0, npaths;
    float4* bqo; float4* bqd; uint32_t* bqpath;  // bounce queue: filled by W_PRIMARY (ctrl->count[1] entries), drained by W_BOUNCE
    uint16_t* mstack; float4* scol;
    SlowQ slow;       // W_PRIMARY: where zero-component rays go (cap == 0: nowhere, they are traced in place); W_SLOW: the queue
    uint32_t slow_k;  // W_SLOW: which consumer launch this is (its range and cursor in the control block)
    int vote_s, vote_l;  // weights of the SELECT / LEAF vote (3 : 2)
};

// Frame of an inner box: node = index of its record; w = visited octants (bits 0-7) | O_DONE | O_HAS;
// t = best hit time inside this box's subtree so far (what the sibling-local skip rule compares against,
// raytrace.rs:965).  Which triangle that was is NOT kept per frame: the ray keeps one running best (t, tri)
// over its leaf results in visiting order.  That equals the reference's nested per-box merge also when hit
// times are NaN: a NaN leaf result is only ever taken by a frame (or by the running best) that has no hit
// yet, it then blocks every later sibling at every level it reaches (`tmin < NaN` is false), and a frame
// that already has a hit drops it (`NaN < t` is false) exactly like the running best does.
#define O_DONE 0x100u
#define O_HAS 0x200u
#define FN_WIDE 0x10000u

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

// 16-byte record at a 32-bit byte offset from a wave-uniform base: the `global_load_dwordx4 v, v_off, s[base]` form, one
// VALU instruction for the address instead of a 64-bit shift and add (rtmi_scene_create checks that every array of the
// octree form is smaller than 4 GiB)
template <typename T>
__device__ inline T ld_off32(const T* base, uint32_t byte_off) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (size_t)byte_off);
}

template <bool COUNT, bool FAST, int MODE>
__device__ __forceinline__ void oct_walk(const DScene& sc, const OctArgs& a, DCtrl* __restrict__ ctrl, uint32_t* __restrict__ lds,
                                         int refill_min, int xcd_aware) {
    const int lane = threadIdx.x;  // one wave per block
    constexpr int NT = 64;
    // which queue of the control block this launch drains: W_TRACE pass `pass`, W_PRIMARY the implicit queue of all paths
    // of the batch (slot 0), W_BOUNCE the bounce queue (slot 1)
    const int pass = MODE == W_TRACE ? a.pass : (MODE == W_PRIMARY ? 0 : 1);
    const uint32_t count = MODE == W_PRIMARY ? a.npaths : (MODE == W_SLOW ? ctrl->shi[a.slow_k] : ctrl->count[pass]);
    // "Rays": every queued ray of this launch (the slow path counts its rays one by one, below)
    if (MODE != W_SLOW && blockIdx.x == 0 && lane == 0) atomicAdd(&ctrl->rays, (unsigned long long)count);
    const float4* __restrict__ qo = MODE == W_BOUNCE ? a.bqo : (MODE == W_SLOW ? a.slow.o : a.qo);
    const float4* __restrict__ qd = MODE == W_BOUNCE ? a.bqd : (MODE == W_SLOW ? a.slow.d : a.qd);
    if (MODE == W_SLOW) refill_min = 1;
    uint32_t path = 0;    // W_PRIMARY / W_BOUNCE: the path this lane works for (slot of its sample colour)
    uint32_t bounce = 0;  // bounces the path has behind it = the reference's maxdepth - depth of the ray being traced
    uint32_t ncont = 0;   // W_BOUNCE: rays this lane cast beyond the queued ones (the "Rays" statistic)
    unsigned long long cnt[5] = {0, 0, 0, 0, 0};
    // COUNT only: S steps, S lanes, L steps, L lanes, refills, refill lanes, edge steps, edge lanes, then shader-clock
    // cycles (s_memtime) this wave spent in SELECT steps, LEAF steps, refills, and in total
    // dbg[12..15]: plane tests whose `t < 0` follows from the signs and exponents of num / den alone (the quotient is a
    // negative NORMAL number), (LEAF step, reference k) slots in which that holds for EVERY working lane -- the only case in
    // which leaving out the division is a saving for the wave --, all such slots, LEAF steps in which all 4 slots are so
    unsigned long long dbg[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = COUNT ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const float root_half = sc.root_half;
    const uint32_t inf_bits = 0x7F800000u;

    uint32_t mode = M_IDLE;
    bool exhausted = false;  // wave-uniform
    // HW_REG_XCC_ID (id 20, bits 3:0): which XCD this wave runs on; only a locality hint
    const uint32_t nranges = xcd_aware ? 8u : 1u;
    const uint32_t home = xcd_aware == 1 ? (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u) : (blockIdx.x & 7u);
    uint32_t tries = 0;      // ranges this wave has seen exhausted (wave-uniform)
    RayK r = make_rayk(make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 1.f, 0.f));
    uint32_t ridx = 0;
    uint32_t fnode = 0, fw = 0;  // current frame
    float ft = 0.f;
    int lvl = 0;                 // depth of the current frame's box
    bool ghave = false;          // running best over the ray's leaf results
    float gt = 0.f;
    uint32_t gtf = 0;
    uint32_t lblock = 0;
    bool lhave = false;
    float lt = 0.f;
    uint32_t ltf = 0;

    for (;;) {
        const unsigned long long m_idle = __ballot(mode == M_IDLE);
        // lanes whose ray is finished and waits to be shaded (path kernels only)
        const unsigned long long m_shade = MODE == W_TRACE ? 0ull : __ballot(mode == M_SHADE);
        // lanes the exchange step would serve (the slow path keeps one path per wave: only lane 0 ever takes a ray)
        const unsigned long long m_x = m_shade | (exhausted ? 0ull : (MODE == W_SLOW ? (m_idle & 1ull) : m_idle));
        if (MODE == W_TRACE ? (m_idle == ~0ull && exhausted) : ((m_idle | m_shade) == ~0ull && m_x == 0ull)) break;
        if (m_x != 0ull && (__popcll(m_x) >= refill_min || (m_idle | m_shade) == ~0ull)) {
            // ---- exchange step: finished rays are shaded, lanes without a ray take consecutive queued rays
            const unsigned long long t_r0 = COUNT ? __builtin_amdgcn_s_memtime() : 0ull;
            bool start = false;     // this lane begins a new ray below
            float4 no = make_float4(0.f, 0.f, 0.f, 0.f), nd = make_float4(0.f, 0.f, 1.f, 0.f);
            uint32_t npath = path, nbounce = bounce;
            if (MODE != W_TRACE) {
                bool push = false;  // W_PRIMARY: the path goes on -> its bounce ray is queued for k_path_bounce
                RayV nr;
                if (mode == M_SHADE) {
                    uint32_t prow, pcol, sample;
                    path_pixel(a.v, a.pix0, path, prow, pcol, sample);
                    const bool cont = shade_hit(sc, a.v.maxdepth, a.seed, a.npaths, path, prow * a.v.width + pcol, sample, bounce,
                                                ghave ? gtf : 0u, gt, V4{r.ox, r.oy, r.oz, r.ow}, V4{r.dx, r.dy, r.dz, r.dw},
                                                a.mstack, a.scol, nr);
                    mode = M_IDLE;
                    if (cont) {
                        if (MODE == W_BOUNCE || MODE == W_SLOW) {
                            no = make_float4(nr.orig.x, nr.orig.y, nr.orig.z, nr.orig.w);
                            nd = make_float4(nr.dir.x, nr.dir.y, nr.dir.z, nr.dir.w);
                            nbounce = bounce + 1u;
                            ncont++;
                            start = true;
                            mode = M_SELECT;  // (not idle: it keeps its lane)
                        } else push = true;
                    }
                }
                if (MODE == W_PRIMARY) {
                    if (push && a.slow.cap && has_zero_component(nr.dir.x, nr.dir.y, nr.dir.z) &&
                        slow_push(a.slow, ctrl, make_float4(nr.orig.x, nr.orig.y, nr.orig.z, nr.orig.w),
                                  make_float4(nr.dir.x, nr.dir.y, nr.dir.z, nr.dir.w), path, 1u))
                        push = false;  // its path goes on in k_path_slow
                    const unsigned long long mask = __ballot(push);
                    if (mask) {
                        uint32_t qb = 0;
                        if (lane == 0) qb = atomicAdd(&ctrl->count[1], (uint32_t)__popcll(mask));
                        qb = __builtin_amdgcn_readfirstlane(qb);
                        if (push) {
                            const uint32_t slot = qb + (uint32_t)__popcll(mask & lt_mask);
                            store_stream(&a.bqo[slot], make_float4(nr.orig.x, nr.orig.y, nr.orig.z, nr.orig.w));
                            store_stream(&a.bqd[slot], make_float4(nr.dir.x, nr.dir.y, nr.dir.z, nr.dir.w));
                            store_stream(&a.bqpath[slot], path);
                        }
                    }
                }
            }
            const unsigned long long m_want = MODE == W_TRACE ? m_idle : (MODE == W_SLOW ? (__ballot(mode == M_IDLE) & 1ull) : __ballot(mode == M_IDLE));
            if (MODE == W_SLOW) {
                if (!exhausted && m_want != 0ull) {  // lane 0 takes the next entry of this launch's range
                    uint32_t i = 0;
                    if (lane == 0) i = ctrl->slo[a.slow_k] + atomicAdd(&ctrl->shead[a.slow_k], 1u);
                    i = __builtin_amdgcn_readfirstlane(i);
                    if (i >= count) exhausted = true;
                    else if (lane == 0) {
                        no = qo[i]; nd = qd[i];
                        npath = a.slow.path[i]; nbounce = a.slow.bounce[i];
                        ncont += nbounce != 0u ? 1u : 0u;  // a diverted bounce ray was not counted by any queue; a primary ray was
                        start = true;
                    }
                }
            } else
            if (!exhausted && m_want != 0ull) {
                const uint32_t n = (uint32_t)__popcll(m_want);
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
                if (!exhausted && mode == M_IDLE) {
                    const uint32_t i = base + (uint32_t)__popcll(m_want & lt_mask);
                    if (i < hi) {
                        if (MODE == W_PRIMARY) {
                            uint32_t prow, pcol, sample;
                            path_pixel(a.v, a.pix0, i, prow, pcol, sample);
                            const RayV pr = pixel_ray(a.v, prow, pcol, a.seed, prow * a.v.width + pcol, sample);
                            no = make_float4(pr.orig.x, pr.orig.y, pr.orig.z, pr.orig.w);
                            nd = make_float4(pr.dir.x, pr.dir.y, pr.dir.z, pr.dir.w);
                            npath = i; nbounce = 0u;
                            start = true;
                            // an exactly-zero direction component: ~150 x the work of an ordinary ray and the other 63
                            // samples of the pixel would wait for it -> its path is traced by k_path_slow
                            if (a.slow.cap && has_zero_component(nd.x, nd.y, nd.z) && slow_push(a.slow, ctrl, no, nd, i, 0u)) start = false;
                        } else if (MODE == W_BOUNCE) {
                            no = qo[i]; nd = qd[i];
                            npath = a.bqpath[i]; nbounce = 1u;
                            start = true;
                        } else {
                            ridx = i;
                            r = make_rayk(qo[i], qd[i]);
                            // the root box itself is never slab-tested (raytrace.rs:1272 calls
                            // get_object_intersection_for_ray on it directly): start with its frame
                            fnode = 0; fw = 0; ft = 0.f; lvl = 0;
                            ghave = false; gt = 0.f; gtf = 0;
                            mode = M_SELECT;
                        }
                    }
                }
            }
            if (MODE == W_SLOW && __ballot(start) != 0ull) {
                // the slow path's one ray per wave (lane 0's) is held by EVERY lane: its leaves are scanned one reference
                // per lane (the wide LEAF step below)
                auto bc = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
                const float4 bo = make_float4(bc(no.x), bc(no.y), bc(no.z), bc(no.w)), bd = make_float4(bc(nd.x), bc(nd.y), bc(nd.z), bc(nd.w));
                r = make_rayk(bo, bd);
            }
            if (MODE != W_TRACE && start) {  // one place where a path kernel's lane takes a ray: bounce in place, or refill
                if (MODE != W_SLOW) r = make_rayk(no, nd);
                path = npath; bounce = nbounce;
                fnode = 0; fw = 0; ft = 0.f; lvl = 0;
                ghave = false; gt = 0.f; gtf = 0;
                mode = M_SELECT;
            }
            if (COUNT && lane == 0) dbg[10] += __builtin_amdgcn_s_memtime() - t_r0;
            // no `continue`: the step below runs in the same iteration (one back edge, fewer copies of the loop-carried state)
        }
        // 32-bit counts: hipcc compares two popcountll results as 64-bit values, on the VALU
        const unsigned long long mS = __ballot(mode == M_SELECT), mL = __ballot(mode == M_LEAF);
        const int nS = __builtin_popcount((uint32_t)mS) + __builtin_popcount((uint32_t)(mS >> 32));
        const int nL = __builtin_popcount((uint32_t)mL) + __builtin_popcount((uint32_t)(mL >> 32));

        const unsigned long long t_s0 = COUNT ? __builtin_amdgcn_s_memtime() : 0ull;
        // Majority vote, weighted 3 : 2 towards SELECT: a LEAF step costs 1.7 x the instructions of a SELECT step, so it pays
        // to let a few more lanes gather for it (1:1 918, 3:2 925, 2:1 920, 2:3 905 Mrays/s);
        const bool stepS = nS * a.vote_s >= nL * a.vote_l;
        if (COUNT && lane == 0) { if (stepS) { dbg[0]++; dbg[1] += nS; } else { dbg[2]++; dbg[3] += nL; } }
        if (stepS) {  // hysteresis (stay in a phase until its lanes fall below 1/2..1/8 of the other's) measured 1-7 % slower
            // ================================================= SELECT step
            if (mode == M_SELECT) {
                // pop finished frames; the frames of depth 0 .. lvl-1 are in LDS levels 0 .. lvl-1
                while ((fw & O_DONE) && lvl != 0) {
                    const bool have = (fw & O_HAS) != 0;
                    const float ct = ft;
                    lvl--;
                    const uint32_t* fr = lds + lvl * 2 * NT + lane;
                    fnode = fr[0] & 0x3FFFFFu;
                    fw = fr[0] >> 22;
                    ft = __uint_as_float(fr[NT]);
                    if (have) {  // fold step of raytrace.rs:949-1007: first hit is taken, later ones replace iff strictly closer
                        if (!(fw & O_HAS) || ct < ft) ft = ct;
                        fw |= O_HAS;
                    }
                }
                if (fw & O_DONE) {  // the root frame is finished: the ray is
                    if (MODE == W_TRACE) {
                        a.hit_tf[ridx] = ghave ? gtf : 0u;
                        a.hit_t[ridx] = ghave ? gt : 0.f;
                        mode = M_IDLE;
                    } else mode = M_SHADE;  // shaded in the next exchange step, together with the other finished lanes
                } else {
                    const uint4 q0 = ld_off32(sc.fnodes, fnode << 5), q1 = ld_off32(sc.fnodes, (fnode << 5) + 16u);
                    const float cx = __uint_as_float(q0.x), cy = __uint_as_float(q0.y), cz = __uint_as_float(q0.z);
                    const float hc = ldexpf(root_half, -(lvl + 1));  // half edge of the children (depth lvl + 1)
                    if (COUNT && (fw & 0xFFu) == 0u) { cnt[0] += __popc(q0.w & 0xFFu); cnt[3]++; }  // first visit: collides() on every child
                    // candidate planes per axis (child centres: builder's orig.add(off_vec))
                    const float xl = cx + (-hc), xh = cx + hc, yl = cy + (-hc), yh = cy + hc, zl = cz + (-hc), zh = cz + hc;
                    // t1s = tmp1 - tmp2, t2s = tmp1 + tmp2 with tmp2 = inv_dir * len2 (raytrace.rs:866-870); near/far swap
                    // when inv_dir <= 0.  With b = |inv_dir| * len2 the pair is (a - b, a + b) in both cases, bit for
                    // bit: negation is exact and x + y == x - (-y).
                    const float bx = fabsf(r.ix) * hc, by = fabsf(r.iy) * hc, bz = fabsf(r.iz) * hc;
                    const float axl = (xl - r.ox) * r.ix, axh = (xh - r.ox) * r.ix;
                    const float ayl = (yl - r.oy) * r.iy, ayh = (yh - r.oy) * r.iy;
                    const float azl = (zl - r.oz) * r.iz, azh = (zh - r.oz) * r.iz;
                    float nx[2] = {axl - bx, axh - bx}, fx[2] = {axl + bx, axh + bx};
                    float ny[2] = {ayl - by, ayh - by}, fy[2] = {ayl + by, ayh + by};
                    float nz[2] = {azl - bz, azh - bz}, fz[2] = {azl + bz, azh + bz};
                    // a zero direction component skips its slab (raytrace.rs:872, :882, :892): axis 0 then leaves the
                    // initial (-MAX, MAX); for axes 1, 2 a NaN operand makes max3/min3 return the running value.
                    // Rare: whole waves skip this block.  (inv_dir = -inf with dir = -0 keeps the `inv > 0` choice
                    // of the reference because only |inv_dir| is used above.)
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
                        // RTMI_OPT_FAST (off by default, NOT the reference's traversal): ignore boxes that lie entirely
                        // behind the ray origin.  The reference visits them (collides() has no `tmax > 0` test,
                        // raytrace.rs:902).
                        const bool c = FAST ? ((tmv[o] < tmax) & !(tmax < 0.f)) : (tmv[o] < tmax);
                        hits |= c ? (1u << o) : 0u;
                    }
                    // candidates: colliding, present, not visited yet
                    const uint32_t cand = hits & q0.w & ~fw & 0xFFu;
                    const uint32_t nh = (uint32_t)__popc(cand);
                    float tm[8];
#pragma unroll
                    for (int o = 0; o < 8; o++) {
                        const uint32_t ex = (uint32_t)__builtin_amdgcn_sbfe((int)cand, o, 1);
                        tm[o] = __uint_as_float((__float_as_uint(tmv[o]) & ex) | (inf_bits & ~ex));  // a colliding tmin is never NaN and never +inf
                    }
                    const float m1 = min3f(min3f(tm[0], tm[1], tm[2]), min3f(tm[3], tm[4], tm[5]), min3f(tm[6], tm[7], tm[7]));
                    // Next child of the sorted order = smallest tmin among the remaining ones, lowest index on ties
                    // (the insertion sort is stable).  Skip rule raytrace.rs:965: with a hit in this box only a child
                    // with tmin < best t is entered, and since the order is ascending and the best t never grows the
                    // first child that fails ends the box.  Without a hit yet a child whose tmin == f32::MAX is not
                    // entered (raytrace.rs:986; it looks like an empty boxmap slot) -- and then neither is any later
                    // one, because every remaining tmin is >= this one.
                    // One compare does all of that: without a candidate m1 is +inf, with one it is in [-MAX, MAX] (a
                    // colliding tmin is never NaN or +inf), where `!= MAX` is `< MAX`; and `inf < ft` is false for every ft.
                    const bool ok = m1 < ((fw & O_HAS) ? ft : FLT_MAX);
                    uint32_t bit = 0u;
#pragma unroll
                    for (int o = 7; o >= 0; o--) bit = (tm[o] == m1) ? (1u << o) : bit;
                    if (!ok) {
                        fw |= O_DONE;
                    } else {
                        fw |= bit | (nh == 1u ? O_DONE : 0u);  // nothing remains after the last candidate
                        const uint32_t leafmask = (q0.w >> 8) & 0xFFu;
                        if (leafmask & bit) {
                            // first block of this leaf child: base + byte `octant` of the offsets
                            const uint32_t oct = (uint32_t)__ffs((int)bit) - 1u;
                            if (q0.w & FN_WIDE) lblock = sc.wlinks[(size_t)q1.y * 8u + oct];  // rare: explicit indices
                            else lblock = q1.y + __builtin_amdgcn_ubfe(oct < 4u ? q1.z : q1.w, (oct & 3u) * 8u, 8u);
                            lhave = false;  // lt, ltf are dead until the first hit of the leaf sets them
                            if (COUNT) cnt[4]++;
                            mode = M_LEAF;
                        } else {
                            uint32_t* fr = lds + lvl * 2 * NT + lane;
                            fr[0] = fnode | (fw << 22);  // record index (< 2^22, checked at scene creation) | visited bits, DONE, HAS
                            fr[NT] = __float_as_uint(ft);
                            lvl++;
                            // record of this inner child: the inner children before it in octant order
                            fnode = q1.x + (uint32_t)__popc((q0.w & ~leafmask & 0xFFu) & (bit - 1u));
                            fw = 0u; ft = 0.f;
                        }
                    }
                }
            }
        }
        // (a second `if`, not an `else`: hipcc's CFG structurizer then has two plain if-regions to handle instead of an
        // if/else whose join needs copies of every loop-carried register written on either side -- 25 fewer v_mov per iteration,
        // what round 2 got from an LLVM-internal switch that later miscompiled this loop, csrc/Makefile)
        if (!stepS) {
            // ================================================= LEAF step: one block of <= 4 references
            if (MODE == W_SLOW) {
                // ---- wide LEAF step of the slow path: the wave holds ONE ray (lane 0 owns its state, every lane has a copy
                // of the ray), so the leaf is scanned one REFERENCE per lane, 16 blocks per step, and the reference's
                // sequential fold (get_box_min_time_intersection, raytrace.rs:1012-1050: the first hit is taken, a later
                // one replaces it iff strictly closer) is replayed over the hits in list order on the scalar unit.
                const uint32_t lb0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)lblock);
                const uint32_t myb = lb0 + ((uint32_t)lane >> 2);
                const bool inb = myb < sc.noblocks;
                const uint4 blk = inb ? ld_off32(sc.oblocks, myb << 4) : make_uint4(0u, 0u, 0u, 0u);
                const uint32_t cpos = (uint32_t)lane & 3u;
                const uint32_t id = (cpos == 0u ? blk.x : cpos == 1u ? blk.y : cpos == 2u ? blk.z : blk.w) & 0x7FFFFFFFu;
                const bool term = blk.w == 0u || (blk.w >> 31);            // this block ends the list (a block past the array does too)
                const unsigned long long tm = __ballot(term);
                const uint32_t endb = tm ? ((uint32_t)(__ffsll((long long)tm) - 1) >> 2) : 16u;  // first terminating block of the 16
                const bool valid = (((uint32_t)lane >> 2) <= endb) & (id != 0u);              // a 0 is padding behind the list's end
                float t = 0.f;
                uint32_t face = 0u;
                bool hit = false;
                if (valid) hit = tri_test<COUNT>(sc, id, r, t, face, cnt);
                bool have = __builtin_amdgcn_readfirstlane((int)lhave) != 0;
                float best = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(lt)));
                uint32_t btf = (uint32_t)__builtin_amdgcn_readfirstlane((int)ltf);
                const uint32_t mytf = id | (face << 30);
                for (unsigned long long hm = __ballot(hit); hm; hm &= hm - 1ull) {
                    const int l = __ffsll((long long)hm) - 1;
                    const float tl = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t), l));
                    if (!have || tl < best) { best = tl; btf = (uint32_t)__builtin_amdgcn_readlane((int)mytf, l); }
                    have = true;
                }
                if (lane == 0) {
                    lhave = have; lt = best; ltf = btf;
                    if (tm == 0ull) lblock = lb0 + 16u;  // no end among these 16 blocks: the list goes on
                    else {
                        if (lhave) {
                            if (!(fw & O_HAS) || lt < ft) ft = lt;
                            fw |= O_HAS;
                            if (!ghave || lt < gt) { gt = lt; gtf = ltf; }
                            ghave = true;
                        }
                        mode = M_SELECT;
                    }
                }
            } else
            if (mode == M_LEAF) {
                const uint4 blk = ld_off32(sc.oblocks, lblock << 4);
                const uint32_t ids[4] = {blk.x, blk.y, blk.z, blk.w & 0x7FFFFFFFu};
                float4 p0[4], p1[4];
#pragma unroll
                for (int k = 0; k < 4; k++) { p0[k] = ld_off32(sc.tplane, ids[k] << 5); p1[k] = ld_off32(sc.tplane, (ids[k] << 5) + 16u); }
                const bool more = blk.w != 0u && !(blk.w >> 31);  // bit 31 of the 4th index: this full block is the last
                if (more) lblock++;
                // Triangle::intersects (raytrace.rs:400-439), plane part for the 4 references, branch-free (a
                // padding index 0 reads the sentinel's record and is masked out); see tri_test() for the lane-3
                // terms.  A reference that passes `t >= 0` and the bounding-radius test becomes the lane's pending
                // candidate; its edge part runs below, once per step.
                uint32_t ptri = 0u;
                uint32_t allneg = 0u;  // COUNT only
                float pt = 0.f, pix = 0.f, piy = 0.f, piz = 0.f, pden = 0.f;
                auto resolve = [&]() {
                    // all four edge records are requested together and every comparison is evaluated (no
                    // short-circuit): one memory round trip instead of three
                    if (COUNT) { cnt[2]++; const unsigned long long em = __ballot(true); if (lane == __ffsll((long long)em) - 1) { dbg[6]++; dbg[7] += __popcll(em); } }
                    const uint32_t eo = ptri << 6;
                    const float4 e0 = ld_off32(sc.tedge, eo), e1 = ld_off32(sc.tedge, eo + 16u), e2 = ld_off32(sc.tedge, eo + 32u), e3 = ld_off32(sc.tedge, eo + 48u);
                    const float pz = (r.dw * pt + r.ow) * 0.f;  // lane-3 product ip.w * side.w (side.w is +-0); ip.w as the plane part computed it
                    const float d0 = ((pix * e0.x + piy * e0.y) + piz * e0.z) + pz;
                    const float d1 = ((pix * e1.x + piy * e1.y) + piz * e1.z) + pz;
                    const float d2 = ((pix * e2.x + piy * e2.y) + piz * e2.z) + pz;
                    const bool inside = !(d0 > e0.w) & !(d1 > e1.w) & !(d2 > e2.w);
                    const bool edge = (d0 > e3.x) | (d1 > e3.y) | (d2 > e3.z);
                    const uint32_t face = (pden > 0.f ? 1u : 0u) | (edge ? 2u : 0u);  // back face: norm . dir > 0
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
                    if (COUNT) {
                        const bool dec = !real | ((t <= -1.17549435e-38f) & (t >= -FLT_MAX));
                        const unsigned long long wm = __ballot(true), dm = __ballot(dec);
                        if (real & dec) dbg[12]++;
                        if (lane == __ffsll((long long)wm) - 1) { dbg[14]++; if (dm == wm) { dbg[13]++; allneg++; } }
                    }
                    const bool c = real & !(t < 0.f) & !(l2 > p0[k].w);
                    if (c & (ptri != 0u)) resolve();  // second candidate of this lane in one block: rare
                    ptri = c ? ids[k] : ptri;
                    pt = c ? t : pt;
                    pix = c ? ix : pix; piy = c ? iy : piy; piz = c ? iz : piz;
                    pden = c ? den : pden;
                }
                if (ptri != 0u) resolve();
                if (COUNT && allneg == 4u) dbg[15]++;
                if (!more) {
                    if (lhave) {
                        if (!(fw & O_HAS) || lt < ft) ft = lt;
                        fw |= O_HAS;
                        if (!ghave || lt < gt) { gt = lt; gtf = ltf; }
                        ghave = true;
                    }
                    mode = M_SELECT;
                }
            }
        }
        if (COUNT && lane == 0) {  // the step is over for the wave when its slowest lane is (s_memtime is a scalar read)
            const unsigned long long dt = __builtin_amdgcn_s_memtime() - t_s0;
            if (stepS) dbg[8] += dt; else dbg[9] += dt;
        }
    }
    if (MODE == W_BOUNCE || MODE == W_SLOW) {  // "Rays": the queued bounce rays were counted above, the ones cast in place here
        unsigned long long wsum = 0;
        for (unsigned long long m = __ballot(ncont != 0u); m; m &= m - 1ull)
            wsum += (uint32_t)__builtin_amdgcn_readlane((int)ncont, __ffsll((long long)m) - 1);
        if (lane == 0 && wsum) atomicAdd(&ctrl->rays, wsum);
    }
    if (COUNT) {
        if (lane == 0) dbg[11] = __builtin_amdgcn_s_memtime() - t_begin;
#pragma unroll
        for (int k = 0; k < 5; k++)
            if (cnt[k]) atomicAdd(&ctrl->counters[k], cnt[k]);
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (dbg[k]) atomicAdd(&ctrl->dbg[k], dbg[k]);
    }
}

#ifndef RTMI_TRACE_WAVES
#define RTMI_TRACE_WAVES 4
#endif
template <bool COUNT, bool FAST>
__global__ void __launch_bounds__(64, RTMI_TRACE_WAVES) k_trace_oct(DScene sc, OctArgs a, DCtrl* __restrict__ ctrl, int refill_min, int xcd_aware) {
    extern __shared__ uint32_t lds[];
    oct_walk<COUNT, FAST, W_TRACE>(sc, a, ctrl, lds, refill_min, xcd_aware);
}
// 6 waves per SIMD = at most 80 VGPRs, the occupancy k_trace_oct has (24 waves per CU is where this walk peaks, DESIGN.md
// 4.1).  The exchange step (Philox + shading with the whole traversal state live) needs ~95; with the cap hipcc keeps the
// SELECT / LEAF steps spill-free and parks launch constants in scratch that only the (rare) exchange step reloads.
#ifndef RTMI_PATH_WAVES
#define RTMI_PATH_WAVES 6
#endif
template <bool COUNT, bool FAST>
__global__ void __launch_bounds__(64, RTMI_PATH_WAVES) k_path_primary(DScene sc, OctArgs a, DCtrl* __restrict__ ctrl, int refill_min, int xcd_aware) {
    extern __shared__ uint32_t lds[];
    oct_walk<COUNT, FAST, W_PRIMARY>(sc, a, ctrl, lds, refill_min, xcd_aware);
}
template <bool COUNT, bool FAST>
__global__ void __launch_bounds__(64, RTMI_PATH_WAVES) k_path_bounce(DScene sc, OctArgs a, DCtrl* __restrict__ ctrl, int refill_min, int xcd_aware) {
    extern __shared__ uint32_t lds[];
    oct_walk<COUNT, FAST, W_BOUNCE>(sc, a, ctrl, lds, refill_min, xcd_aware);
}
template <bool COUNT, bool FAST>
__global__ void __launch_bounds__(64, RTMI_PATH_WAVES) k_path_slow(DScene sc, OctArgs a, DCtrl* __restrict__ ctrl, int refill_min, int xcd_aware) {
    extern __shared__ uint32_t lds[];
    oct_walk<COUNT, FAST, W_SLOW>(sc, a, ctrl, lds, refill_min, xcd_aware);
}

}  // namespace rtmi
