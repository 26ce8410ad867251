// bvh_fast.hpp — RTMI_OPT_BVH, the opt-in "fast mode" (SURVEY §8 f2).  NOT the reference's octree traversal.
//
// What it computes: for every ray the closest hit over ALL triangles of the scene, each tested with the reference's
// Triangle::intersects arithmetic (raytrace.rs:400-439, tri_test()), the lowest triangle index winning exact ties —
// i.e. exactly what the reference itself computes for a scene whose accelerator is build_trivial_bounding_box
// (raytrace.rs:847-856: one leaf, the list scanned in index order with strict `<`, raytrace.rs:1012-1050).  That
// linear-list render is what this mode is checked against, with ONE stated exception: a "hit" whose time is +-inf or NaN is ignored.
// The reference accepts such hits (a ray exactly parallel to a triangle's plane, norm.dir == 0: lane 3 of
// Vec3::mult(inf) turns NaN and every later comparison passes, raytrace.rs:402-439) wherever the triangle is, so no
// spatial index can find them; they are artefacts (a few per 10^4 samples of a linear-list render).  The tests compare
// this mode with the CPU restatement of the reference's linear list with those hits switched off, bit for bit.  Against the octree traversal the
// result can also differ where the octree's builder lost a triangle (false negative of box_contains_polygon) and
// where two triangles tie exactly; bench.py --bvh reports the differing-pixel count.
//
// How: a BVH built by the library at scene creation: binned surface-area heuristic (16 bins on the centroids), leaves of
// <= 4 triangles, then collapsed to 4-WIDE nodes (one 128-byte record = one cache line per node: the 4 child boxes as
// six float4 planes + 4 links).  A triangle can only be hit in its plane inside its bounding radius (the reference's own
// radius test, raytrace.rs:411-413), so the axis-aligned box of that disc (half-extent r*sqrt(1 - n_k^2), widened) bounds
// every hit point and the records of rtmi_triangle_t suffice; when the caller also hands over the corners
// (rtmi_scene_set_corners; `Triangle.corners`, raytrace.rs:326-337) the box is that disc box INTERSECTED with the
// corners' box -- the three half-plane tests (raytrace.rs:415-424) accept points of the triangle only.  Traversal:
// persistent waves, one lane = one ray, per-lane stack of links in LDS, the wave alternates INNER steps (4 slab tests,
// the hit children pushed far-to-near so that the nearest is visited next) and LEAF steps (<= 4 triangles: four
// branch-free plane tests, the edge part once per step for the lanes that have a candidate, as in k_trace_oct) by
// majority vote; subtrees whose entry distance exceeds the best hit so far are pruned -- the lever the exact mode may
// not use.
#pragma once

namespace rtmi {

// ---------------------------------------------------------------- host: binned SAH build
struct BvhBuild {
    std::vector<float4> nodes;   // binary build: 4 x float4 per inner node: left box, right box, (left link, right link)
    std::vector<float4> wide;    // 4-wide form: 8 x float4 per node: lo.x[4] lo.y[4] lo.z[4] hi.x[4] hi.y[4] hi.z[4] links[4] pad
    std::vector<uint4> leaves;   // <= 4 triangle indices per leaf, 0-padded
    uint32_t depth = 0;          // of the 4-wide tree (inner levels)
    uint32_t root_link = 0;      // link of the root (a leaf when the scene has <= 4 triangles)
};
#define BVH_EMPTY 0xFFFFFFFFu    // link of an unused child slot
#define BVH_CHUNK 512            // queue entries a wave reserves at a time

struct BBox {
    float lo[3], hi[3];
    void reset() { for (int k = 0; k < 3; k++) { lo[k] = FLT_MAX; hi[k] = -FLT_MAX; } }
    void grow(const BBox& o) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], o.lo[k]); hi[k] = std::max(hi[k], o.hi[k]); } }
    float area() const {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return (dx < 0.f || dy < 0.f || dz < 0.f) ? 0.f : 2.f * (dx * dy + dy * dz + dz * dx);
    }
};

// Builds over triangles 1 .. n-1 (0 is the sentinel).  Returns false when a record is not finite (then no BVH).
// corners9: optional, 9 floats per triangle (the corners the records were made from), index 0 = the sentinel
static bool bvh_build(const rtmi_triangle_t* tris, uint64_t ntris, BvhBuild& out, const float* corners9 = nullptr) {
    out = BvhBuild();
    if (ntris < 2) return false;
    const uint32_t n = (uint32_t)(ntris - 1);
    std::vector<BBox> box(n);
    std::vector<uint32_t> idx(n);
    for (uint32_t i = 0; i < n; i++) {
        const rtmi_triangle_t& t = tris[i + 1];
        // Every hit point the reference accepts lies in the triangle's plane (it is dir*t + orig with t solved from
        // the plane equation) and has fl(|p - c|^2) <= r2: it is in the disc of radius r around the centroid in that
        // plane, whose axis-aligned box has half-extent r*sqrt(1 - n_k^2) along axis k.  Widened by a margin far above
        // the rounding of p (~2^-23 of the coordinates involved).
        const float r = std::sqrt(t.bounding_r2) * 1.00001f;
        const float cabs = std::fabs(t.incenter[0]) + std::fabs(t.incenter[1]) + std::fabs(t.incenter[2]);
        const float eps = 2e-5f * (cabs + r + 1.f);
        if (!std::isfinite(r) || r < 0.f) return false;
        for (int k = 0; k < 3; k++) {
            if (!std::isfinite(t.incenter[k]) || !std::isfinite(t.norm[k])) return false;
            const float s2 = 1.f - t.norm[k] * t.norm[k];
            const float half = r * std::sqrt(s2 > 0.f ? s2 : 0.f) * 1.0001f + eps;
            box[i].lo[k] = t.incenter[k] - half;
            box[i].hi[k] = t.incenter[k] + half;
            if (corners9) {
                // an accepted hit point also passes the three half-plane tests (raytrace.rs:415-424): it lies in the triangle
                // (up to the same rounding margin), i.e. inside the box of the corners
                const float* c = corners9 + 9 * (size_t)(i + 1);
                if (!std::isfinite(c[k]) || !std::isfinite(c[3 + k]) || !std::isfinite(c[6 + k])) return false;
                const float clo = std::min(c[k], std::min(c[3 + k], c[6 + k])) - eps;
                const float chi = std::max(c[k], std::max(c[3 + k], c[6 + k])) + eps;
                box[i].lo[k] = std::max(box[i].lo[k], clo);
                box[i].hi[k] = std::min(box[i].hi[k], chi);
                if (box[i].lo[k] > box[i].hi[k]) { box[i].lo[k] = t.incenter[k] - half; box[i].hi[k] = t.incenter[k] + half; }  // inconsistent input: keep the disc box
            }
        }
        idx[i] = i;
    }
    struct Task { uint32_t lo, hi, depth; int32_t parent; int side; };
    std::vector<Task> stack;
    auto emit_leaf = [&](uint32_t lo, uint32_t hi) -> uint32_t {
        // ascending triangle index inside a leaf (ties are resolved by index anyway; this keeps loads ordered)
        std::sort(idx.begin() + lo, idx.begin() + hi);
        uint32_t first = (uint32_t)out.leaves.size();
        for (uint32_t k = lo; k < hi; k += 4) {
            uint32_t v[4] = {0, 0, 0, 0};
            for (uint32_t j = 0; j < 4 && k + j < hi; j++) v[j] = idx[k + j] + 1;
            out.leaves.push_back(make_uint4(v[0], v[1], v[2], v[3]));
        }
        return first | 0x80000000u;
    };
    auto set_link = [&](int32_t parent, int side, uint32_t link, const BBox& b) {
        if (parent < 0) { out.root_link = link; return; }
        float4* q = &out.nodes[4 * (size_t)parent];
        if (side == 0) {
            q[0] = make_float4(b.lo[0], b.lo[1], b.lo[2], b.hi[0]);
            q[1].x = b.hi[1]; q[1].y = b.hi[2];
            q[3].x = __builtin_bit_cast(float, link);
        } else {
            q[1].z = b.lo[0]; q[1].w = b.lo[1];
            q[2] = make_float4(b.lo[2], b.hi[0], b.hi[1], b.hi[2]);
            q[3].y = __builtin_bit_cast(float, link);
        }
    };
    stack.push_back(Task{0, n, 0, -1, 0});
    while (!stack.empty()) {
        const Task tk = stack.back();
        stack.pop_back();
        out.depth = std::max(out.depth, tk.depth);
        BBox b, cb;
        b.reset(); cb.reset();
        for (uint32_t k = tk.lo; k < tk.hi; k++) {
            b.grow(box[idx[k]]);
            const rtmi_triangle_t& t = tris[idx[k] + 1];
            for (int a = 0; a < 3; a++) { cb.lo[a] = std::min(cb.lo[a], t.incenter[a]); cb.hi[a] = std::max(cb.hi[a], t.incenter[a]); }
        }
        const uint32_t cnt = tk.hi - tk.lo;
        if (cnt <= 4) { set_link(tk.parent, tk.side, emit_leaf(tk.lo, tk.hi), b); continue; }
        // binned SAH over the centres, best axis; median split when the centres coincide or the tree gets too deep
        int best_axis = -1, best_bin = -1;
        float best_cost = FLT_MAX;
        constexpr int NB = 16;
        for (int a = 0; a < 3 && tk.depth < 40; a++) {
            const float ext = cb.hi[a] - cb.lo[a];
            if (!(ext > 0.f)) continue;
            BBox bb[NB]; uint32_t bc[NB];
            for (int i = 0; i < NB; i++) { bb[i].reset(); bc[i] = 0; }
            const float scale = (float)NB / ext;
            for (uint32_t k = tk.lo; k < tk.hi; k++) {
                int bi = (int)((tris[idx[k] + 1].incenter[a] - cb.lo[a]) * scale);
                bi = bi < 0 ? 0 : (bi >= NB ? NB - 1 : bi);
                bb[bi].grow(box[idx[k]]); bc[bi]++;
            }
            float right_area[NB]; uint32_t right_cnt[NB];
            BBox acc; acc.reset(); uint32_t c = 0;
            for (int i = NB - 1; i > 0; i--) { acc.grow(bb[i]); c += bc[i]; right_area[i] = acc.area(); right_cnt[i] = c; }
            acc.reset(); c = 0;
            for (int i = 0; i < NB - 1; i++) {
                acc.grow(bb[i]); c += bc[i];
                if (c == 0 || right_cnt[i + 1] == 0) continue;
                const float cost = acc.area() * (float)c + right_area[i + 1] * (float)right_cnt[i + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = i; }
            }
        }
        uint32_t mid;
        if (best_axis >= 0) {
            const float ext = cb.hi[best_axis] - cb.lo[best_axis], scale = (float)NB / ext;
            auto it = std::partition(idx.begin() + tk.lo, idx.begin() + tk.hi, [&](uint32_t i) {
                int bi = (int)((tris[i + 1].incenter[best_axis] - cb.lo[best_axis]) * scale);
                bi = bi < 0 ? 0 : (bi >= NB ? NB - 1 : bi);
                return bi <= best_bin;
            });
            mid = (uint32_t)(it - idx.begin());
        } else {
            mid = tk.lo + cnt / 2;  // coincident centres / depth cap: any balanced split is valid
        }
        if (mid == tk.lo || mid == tk.hi) mid = tk.lo + cnt / 2;
        const int32_t me = (int32_t)(out.nodes.size() / 4);
        out.nodes.resize(out.nodes.size() + 4, make_float4(0.f, 0.f, 0.f, 0.f));
        set_link(tk.parent, tk.side, (uint32_t)me, b);
        stack.push_back(Task{mid, tk.hi, tk.depth + 1, me, 1});
        stack.push_back(Task{tk.lo, mid, tk.depth + 1, me, 0});
    }
    // ---- collapse to 4-wide nodes: a node's children are its binary children, the one with the largest box replaced by
    //      ITS children until there are four (or only leaves are left)
    struct Child { BBox b; uint32_t link; };
    auto child_of = [&](uint32_t node, int side) {
        const float4* q = &out.nodes[4 * (size_t)node];
        Child c;
        if (side == 0) { c.b.lo[0] = q[0].x; c.b.lo[1] = q[0].y; c.b.lo[2] = q[0].z; c.b.hi[0] = q[0].w; c.b.hi[1] = q[1].x; c.b.hi[2] = q[1].y; c.link = __builtin_bit_cast(uint32_t, q[3].x); }
        else { c.b.lo[0] = q[1].z; c.b.lo[1] = q[1].w; c.b.lo[2] = q[2].x; c.b.hi[0] = q[2].y; c.b.hi[1] = q[2].z; c.b.hi[2] = q[2].w; c.link = __builtin_bit_cast(uint32_t, q[3].y); }
        return c;
    };
    out.depth = 0;
    if (!(out.root_link >> 31)) {
        struct WTask { uint32_t bin, wide, depth; };
        std::vector<WTask> todo;
        out.wide.assign(8, make_float4(0.f, 0.f, 0.f, 0.f));
        todo.push_back(WTask{out.root_link, 0u, 1u});
        out.root_link = 0;  // wide node 0
        while (!todo.empty()) {
            const WTask wt = todo.back();
            todo.pop_back();
            out.depth = std::max(out.depth, wt.depth);
            std::vector<Child> ch{child_of(wt.bin, 0), child_of(wt.bin, 1)};
            while (ch.size() < 4) {
                int best = -1;
                float ba = -1.f;
                for (size_t k = 0; k < ch.size(); k++)
                    if (!(ch[k].link >> 31) && ch[k].b.area() > ba) { ba = ch[k].b.area(); best = (int)k; }
                if (best < 0) break;
                const uint32_t inner = ch[(size_t)best].link;
                ch[(size_t)best] = child_of(inner, 0);
                ch.push_back(child_of(inner, 1));
            }
            float lo[3][4], hi[3][4];
            uint32_t link[4];
            for (int k = 0; k < 4; k++) {
                for (int a = 0; a < 3; a++) { lo[a][k] = FLT_MAX; hi[a][k] = -FLT_MAX; }
                link[k] = BVH_EMPTY;
            }
            for (size_t k = 0; k < ch.size(); k++) {
                for (int a = 0; a < 3; a++) { lo[a][k] = ch[k].b.lo[a]; hi[a][k] = ch[k].b.hi[a]; }
                if (ch[k].link >> 31) link[k] = ch[k].link;
                else {
                    link[k] = (uint32_t)(out.wide.size() / 8);
                    out.wide.resize(out.wide.size() + 8, make_float4(0.f, 0.f, 0.f, 0.f));
                    todo.push_back(WTask{ch[k].link, link[k], wt.depth + 1});
                }
            }
            float4* w = &out.wide[8 * (size_t)wt.wide];
            for (int a = 0; a < 3; a++) {
                w[a] = make_float4(lo[a][0], lo[a][1], lo[a][2], lo[a][3]);
                w[3 + a] = make_float4(hi[a][0], hi[a][1], hi[a][2], hi[a][3]);
            }
            w[6] = make_float4(__builtin_bit_cast(float, link[0]), __builtin_bit_cast(float, link[1]), __builtin_bit_cast(float, link[2]), __builtin_bit_cast(float, link[3]));
        }
    }
    return true;
}

// ---------------------------------------------------------------- device: traversal
enum : uint32_t { B_IDLE = 0, B_INNER = 1, B_LEAF = 2 };

// One child's slab test on [0, tbest]: entry distance in `tn`.  Conservative (a box is never missed because of rounding:
// the far plane is widened by 2 ulp, Ize's robust test); NaN operands (0 * inf on a zero direction component whose origin
// lies on the plane) are ignored by v_min/v_max (IEEE minNum/maxNum), which keeps the slab open; an empty slot's inverted
// box (lo = MAX, hi = -MAX) never passes.
__device__ inline bool bvh_slab(float lox, float loy, float loz, float hix, float hiy, float hiz, const RayK& r, float tbest, float& tn) {
    const float tx0 = (lox - r.ox) * r.ix, tx1 = (hix - r.ox) * r.ix;
    const float ty0 = (loy - r.oy) * r.iy, ty1 = (hiy - r.oy) * r.iy;
    const float tz0 = (loz - r.oz) * r.iz, tz1 = (hiz - r.oz) * r.iz;
    const float tmin = fmaxf(max3f(fminf(tx0, tx1), fminf(ty0, ty1), fminf(tz0, tz1)), 0.f);
    const float tmax = min3f(fmaxf(tx0, tx1), fmaxf(ty0, ty1), fmaxf(tz0, tz1)) * 1.0000003f;
    tn = tmin;
    return (tmin <= tmax) & (tmin <= tbest) & (tmin < INFINITY);
}

template <bool COUNT>
__global__ void __launch_bounds__(64, 5) k_trace_bvh(DScene sc, const float4* __restrict__ bnodes, const uint4* __restrict__ bleaves,
                                                  uint32_t root_link, const float4* __restrict__ qo, const float4* __restrict__ qd,
                                                  DCtrl* __restrict__ ctrl, int pass, uint32_t* __restrict__ hit_tf,
                                                  float* __restrict__ hit_t, int refill_min) {
    extern __shared__ uint32_t lds[];  // [level][lane] links waiting to be visited
    const int lane = threadIdx.x;
    constexpr int NT = 64;
    const uint32_t count = ctrl->count[pass];
    if (blockIdx.x == 0 && lane == 0) atomicAdd(&ctrl->rays, (unsigned long long)count);
    unsigned long long cnt[5] = {0, 0, 0, 0, 0};
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t mode = B_IDLE;
    bool exhausted = false;
    RayK r = make_rayk(make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 1.f, 0.f));
    uint32_t ridx = 0, cur = 0;  // link being visited
    int sp = 0;
    // closest hit so far (finite time, lowest index on ties)
    float bt = INFINITY;
    uint32_t btf = 0;
    uint32_t wnext = 0, wend = 0;  // the wave's reserved range of queue entries (wave-uniform)

    for (;;) {
        const unsigned long long m_idle = __ballot(mode == B_IDLE);
        if (m_idle == ~0ull && exhausted) break;
        if (!exhausted && (__popcll(m_idle) >= refill_min || m_idle == ~0ull)) {
            // Rays of this mode are short (tens of steps): a wave reserves BVH_CHUNK queue entries with ONE atomic and deals
            // them out to its idle lanes over several refills -- with one atomic per refill the single queue cursor was the
            // bottleneck of the bounce passes (8-lane refills: 134 ms per frame, 32-lane refills: 79 ms).
            if (wnext >= wend) {
                uint32_t b = 0;
                if (lane == 0) b = atomicAdd(&ctrl->head[pass], (uint32_t)BVH_CHUNK);
                wnext = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
                wend = min(wnext + (uint32_t)BVH_CHUNK, count);
                if (wnext >= count) { exhausted = true; wend = wnext; }
            }
            const uint32_t base = wnext;
            wnext = min(wnext + (uint32_t)__popcll(m_idle), wend);
            if (!exhausted && mode == B_IDLE) {
                const uint32_t i = base + (uint32_t)__popcll(m_idle & lt_mask);
                if (i < wend) {
                    ridx = i;
                    r = make_rayk(qo[i], qd[i]);
                    cur = root_link; sp = 0;
                    bt = INFINITY; btf = 0;
                    mode = (root_link >> 31) ? B_LEAF : B_INNER;
                }
            }
        }
        const unsigned long long mI = __ballot(mode == B_INNER), mL = __ballot(mode == B_LEAF);
        const int nI = __builtin_popcount((uint32_t)mI) + __builtin_popcount((uint32_t)(mI >> 32));
        const int nL = __builtin_popcount((uint32_t)mL) + __builtin_popcount((uint32_t)(mL >> 32));
        const bool stepI = nI >= nL;
        bool pop = false;
        if (stepI) {
            // ================================================= INNER step: the node's 4 children
            if (mode == B_INNER) {
                const float4* q = bnodes + 8 * (size_t)cur;
                const float4 lx = q[0], ly = q[1], lz = q[2], hx = q[3], hy = q[4], hz = q[5], lk = q[6];
                if (COUNT) { cnt[0] += 4; cnt[3]++; }
                float t0, t1, t2, t3;
                uint32_t l[4] = {__float_as_uint(lk.x), __float_as_uint(lk.y), __float_as_uint(lk.z), __float_as_uint(lk.w)};
                const bool h0 = bvh_slab(lx.x, ly.x, lz.x, hx.x, hy.x, hz.x, r, bt, t0) & (l[0] != BVH_EMPTY);
                const bool h1 = bvh_slab(lx.y, ly.y, lz.y, hx.y, hy.y, hz.y, r, bt, t1) & (l[1] != BVH_EMPTY);
                const bool h2 = bvh_slab(lx.z, ly.z, lz.z, hx.z, hy.z, hz.z, r, bt, t2) & (l[2] != BVH_EMPTY);
                const bool h3 = bvh_slab(lx.w, ly.w, lz.w, hx.w, hy.w, hz.w, r, bt, t3) & (l[3] != BVH_EMPTY);
                // sort the four (distance, link) pairs, missed children last (+inf; a hit's distance is finite): 5 compare-exchanges
                float d[4] = {h0 ? t0 : INFINITY, h1 ? t1 : INFINITY, h2 ? t2 : INFINITY, h3 ? t3 : INFINITY};
                auto cx = [&](int a, int b) {
                    const bool sw = d[b] < d[a];
                    const float da = sw ? d[b] : d[a], db = sw ? d[a] : d[b];
                    const uint32_t la = sw ? l[b] : l[a], lb = sw ? l[a] : l[b];
                    d[a] = da; d[b] = db; l[a] = la; l[b] = lb;
                };
                cx(0, 1); cx(2, 3); cx(0, 2); cx(1, 3); cx(1, 2);
                const int nh = (int)h0 + (int)h1 + (int)h2 + (int)h3;
                // push the farther ones, farthest first, so that the nearest waiting one is popped first
                if (nh > 3) { lds[sp * NT + lane] = l[3]; sp++; }
                if (nh > 2) { lds[sp * NT + lane] = l[2]; sp++; }
                if (nh > 1) { lds[sp * NT + lane] = l[1]; sp++; }
                if (nh > 0) { cur = l[0]; mode = (cur >> 31) ? B_LEAF : B_INNER; }
                else pop = true;
            }
        }
        if (!stepI) {
            // ================================================= LEAF step: <= 4 triangles, plane parts branch-free
            if (mode == B_LEAF) {
                const uint4 blk = bleaves[cur & 0x7FFFFFFFu];
                const uint32_t ids[4] = {blk.x, blk.y, blk.z, blk.w};
                float4 p0[4], p1[4];
#pragma unroll
                for (int k = 0; k < 4; k++) { p0[k] = ld_off32(sc.tplane, ids[k] << 5); p1[k] = ld_off32(sc.tplane, (ids[k] << 5) + 16u); }
                if (COUNT) cnt[4]++;
                uint32_t ptri = 0u;
                float pt = 0.f, pix = 0.f, piy = 0.f, piz = 0.f, pden = 0.f;
                auto resolve = [&]() {  // Triangle::intersects' edge part (raytrace.rs:415-437) for the lane's pending candidate
                    if (COUNT) cnt[2]++;
                    const uint32_t eo = ptri << 6;
                    const float4 e0 = ld_off32(sc.tedge, eo), e1 = ld_off32(sc.tedge, eo + 16u), e2 = ld_off32(sc.tedge, eo + 32u), e3 = ld_off32(sc.tedge, eo + 48u);
                    const float pz = (r.dw * pt + r.ow) * 0.f;
                    const float d0 = ((pix * e0.x + piy * e0.y) + piz * e0.z) + pz;
                    const float d1 = ((pix * e1.x + piy * e1.y) + piz * e1.z) + pz;
                    const float d2 = ((pix * e2.x + piy * e2.y) + piz * e2.z) + pz;
                    const bool inside = !(d0 > e0.w) & !(d1 > e1.w) & !(d2 > e2.w);
                    const bool edge = (d0 > e3.x) | (d1 > e3.y) | (d2 > e3.z);
                    const uint32_t face = (pden > 0.f ? 1u : 0u) | (edge ? 2u : 0u);
                    // finite hit times only (see top); strict `<` of the index-order scan: equal times keep the lower index
                    const bool take = inside & (fabsf(pt) < INFINITY) & ((pt < bt) | ((pt == bt) & (ptri < (btf & 0x3FFFFFFFu))));
                    bt = take ? pt : bt;
                    btf = take ? (ptri | (face << 30)) : btf;
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
                    if (c & (ptri != 0u)) resolve();  // second candidate of this lane in one leaf: rare
                    ptri = c ? ids[k] : ptri;
                    pt = c ? t : pt;
                    pix = c ? ix : pix; piy = c ? iy : piy; piz = c ? iz : piz;
                    pden = c ? den : pden;
                }
                if (ptri != 0u) resolve();
                pop = true;
            }
        }
        if (pop) {
            // next waiting subtree; its entry distance is not kept on the stack: it is re-tested when the subtree is visited
            if (sp > 0) {
                sp--;
                cur = lds[sp * NT + lane];
                mode = (cur >> 31) ? B_LEAF : B_INNER;
            } else {
                hit_tf[ridx] = btf;
                hit_t[ridx] = btf ? bt : 0.f;
                mode = B_IDLE;
            }
        }
    }
    if (COUNT) {
#pragma unroll
        for (int k = 0; k < 5; k++)
            if (cnt[k]) atomicAdd(&ctrl->counters[k], cnt[k]);
    }
}

}  // namespace rtmi
