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
// How: a binary BVH built by the library at scene creation with a binned surface-area heuristic (16 bins on the
// centroids), leaves of <= 4 triangles.  A triangle can only be hit in its plane inside its bounding radius (the
// reference's own radius test, raytrace.rs:411-413), so the axis-aligned box of that disc (half-extent
// r*sqrt(1 - n_k^2), widened) bounds every hit point and the records of rtmi_triangle_t suffice — no corners needed.  Traversal: persistent
// waves, one lane = one ray, per-lane stack in LDS, the wave alternates INNER steps (both children's slab tests,
// near child first) and LEAF steps (<= 4 triangles) by majority vote like k_trace_oct; subtrees whose entry distance
// exceeds the best hit so far are pruned — the lever the exact mode may not use.
#pragma once

namespace rtmi {

// ---------------------------------------------------------------- host: binned SAH build
struct BvhBuild {
    std::vector<float4> nodes;   // 4 x float4 per inner node: left box, right box, (left link, right link)
    std::vector<uint4> leaves;   // <= 4 triangle indices per leaf, 0-padded
    uint32_t depth = 0;
    uint32_t root_link = 0;      // link of the root (a leaf when the scene has <= 4 triangles)
};

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
static bool bvh_build(const rtmi_triangle_t* tris, uint64_t ntris, BvhBuild& out) {
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
    return true;
}

// ---------------------------------------------------------------- device: traversal
enum : uint32_t { B_IDLE = 0, B_INNER = 1, B_LEAF = 2 };

// Ray/box slab test on [0, tbest]: entry distance in `tn`.  Conservative (a box is never missed because of rounding:
// the far plane is widened by 2 ulp, Ize's robust test); NaN operands (0 * inf on a zero direction component whose
// origin lies on the plane) are ignored by fmaxf/fminf, which keeps the slab open.
__device__ inline bool bvh_slab(float lox, float loy, float loz, float hix, float hiy, float hiz, const RayK& r, float tbest, float& tn) {
    const float tx0 = (lox - r.ox) * r.ix, tx1 = (hix - r.ox) * r.ix;
    const float ty0 = (loy - r.oy) * r.iy, ty1 = (hiy - r.oy) * r.iy;
    const float tz0 = (loz - r.oz) * r.iz, tz1 = (hiz - r.oz) * r.iz;
    const float tmin = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), 0.f));
    const float tmax = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fmaxf(tz0, tz1)) * 1.0000003f;
    tn = tmin;
    return (tmin <= tmax) & (tmin <= tbest);
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

    for (;;) {
        const unsigned long long m_idle = __ballot(mode == B_IDLE);
        if (m_idle == ~0ull && exhausted) break;
        if (!exhausted && (__popcll(m_idle) >= refill_min || m_idle == ~0ull)) {
            const uint32_t n = (uint32_t)__popcll(m_idle);
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&ctrl->head[pass], n);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base >= count) { exhausted = true; continue; }
            if (mode == B_IDLE) {
                const uint32_t i = base + (uint32_t)__popcll(m_idle & lt_mask);
                if (i < count) {
                    ridx = i;
                    r = make_rayk(qo[i], qd[i]);
                    cur = root_link; sp = 0;
                    bt = INFINITY; btf = 0;
                    mode = (root_link >> 31) ? B_LEAF : B_INNER;
                }
            }
            continue;
        }
        const int nI = __popcll(__ballot(mode == B_INNER));
        const int nL = __popcll(__ballot(mode == B_LEAF));
        bool pop = false;
        if (nI >= nL) {
            // ================================================= INNER step: both children, near one first
            if (mode == B_INNER) {
                const float4* q = bnodes + 4 * (size_t)cur;
                const float4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
                if (COUNT) { cnt[0] += 2; cnt[3]++; }
                float tl, tr;
                const bool hl = bvh_slab(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, r, bt, tl);
                const bool hr = bvh_slab(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, r, bt, tr);
                const uint32_t ll = __float_as_uint(q3.x), lr = __float_as_uint(q3.y);
                if (hl & hr) {
                    const bool lfirst = tl <= tr;
                    lds[sp * NT + lane] = lfirst ? lr : ll;
                    sp++;
                    cur = lfirst ? ll : lr;
                } else if (hl | hr) {
                    cur = hl ? ll : lr;
                } else pop = true;
                if (!pop) mode = (cur >> 31) ? B_LEAF : B_INNER;
            }
        } else {
            // ================================================= LEAF step: <= 4 triangles
            if (mode == B_LEAF) {
                const uint4 blk = bleaves[cur & 0x7FFFFFFFu];
                const uint32_t ids[4] = {blk.x, blk.y, blk.z, blk.w};
                if (COUNT) cnt[4]++;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (ids[k] != 0u) {
                        float t; uint32_t face;
                        if (tri_test<COUNT>(sc, ids[k], r, t, face, cnt) && fabsf(t) < INFINITY) {  // finite hit times only (see top)
                            const uint32_t tf = ids[k] | (face << 30);
                            // strict `<` of the index-order scan: equal times keep the lower index
                            if (t < bt || (t == bt && ids[k] < (btf & 0x3FFFFFFFu))) { bt = t; btf = tf; }
                        }
                    }
                }
                pop = true;
            }
        }
        if (pop) {
            // next waiting subtree that can still hold a closer (or equal, lower-index) hit; its entry distance is not
            // kept on the stack: it is re-tested when the subtree is visited (INNER step) or costs one leaf
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
