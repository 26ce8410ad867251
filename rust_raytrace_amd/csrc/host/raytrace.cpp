// raytrace.cpp — host-side scene construction (see raytrace.hpp).
// Strict f32, reference operation order; build with -ffp-contract=off.
#include "raytrace.hpp"

#include <atomic>
#include <cfloat>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>

namespace raytrace {

// ------------------------------------------------------------------ Vec3
Vec3 make_vec(const float (&v)[3]) { return Vec3{{v[0], v[1], v[2], 0.f}}; }  // raytrace.rs:29-33

Vec3 Vec3::add(const Vec3& o) const { return Vec3{{v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2], v[3] + o.v[3]}}; }
Vec3 Vec3::sub(const Vec3& o) const { return Vec3{{v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2], v[3] - o.v[3]}}; }
Vec3 Vec3::mult(float a) const { return Vec3{{v[0] * a, v[1] * a, v[2] * a, v[3] * a}}; }
Vec3 Vec3::mult_per(const Vec3& o) const { return Vec3{{v[0] * o.v[0], v[1] * o.v[1], v[2] * o.v[2], v[3] * o.v[3]}}; }

// std::simd reduce_sum: ordered, seeded with 0 (raytrace.rs:65-67)
static inline float hsum(const Vec3& p) {
    float s = 0.f;
    s = s + p.v[0];
    s = s + p.v[1];
    s = s + p.v[2];
    s = s + p.v[3];
    return s;
}
float Vec3::len2() const { return hsum(mult_per(*this)); }
float Vec3::len() const { return std::sqrt(len2()); }
float Vec3::dot(const Vec3& o) const { return hsum(mult_per(o)); }
Vec3 Vec3::cross(const Vec3& o) const {  // raytrace.rs:80-90
    const Vec3 a1{{v[1], v[2], v[0], v[3]}}, a2{{v[2], v[0], v[1], v[3]}};
    const Vec3 b1{{o.v[1], o.v[2], o.v[0], o.v[3]}}, b2{{o.v[2], o.v[0], o.v[1], o.v[3]}};
    return a1.mult_per(b2).sub(a2.mult_per(b1));
}
Vec3 Vec3::unit() const { return mult(1.f / len()); }  // raytrace.rs:93-96
Vec3 Vec3::orthogonal() const {                        // raytrace.rs:98-108
    if (std::fabs(v[0]) > 0.1f) return make_vec(-1.f * (v[1] + v[2]) / v[0], 1.f, 1.f).unit();
    if (std::fabs(v[1]) > 0.1f) return make_vec(1.f, -1.f * (v[0] + v[2]) / v[1], 1.f).unit();
    if (std::fabs(v[2]) > 0.1f) return make_vec(1.f, 1.f, -1.f * (v[0] + v[1]) / v[2]).unit();
    return unit().orthogonal();
}
Vec3 Vec3::change_basis(const std::tuple<Vec3, Vec3, Vec3>& b) const {  // raytrace.rs:117-121
    const Vec3 &b0 = std::get<0>(b), &b1 = std::get<1>(b), &b2 = std::get<2>(b);
    return make_vec(make_vec(b0.v[0], b0.v[1], b0.v[2]).dot(*this), make_vec(b1.v[0], b1.v[1], b1.v[2]).dot(*this),
                    make_vec(b2.v[0], b2.v[1], b2.v[2]).dot(*this));
}

Color make_color(uint8_t r, uint8_t g, uint8_t b) { return make_vec((float)r / 255.f, (float)g / 255.f, (float)b / 255.f); }

Ray make_ray(const Point& orig, const Vec3& dir) {
    const Vec3 du = dir.unit();
    return Ray{orig, du, make_vec(1.f / du.v[0], 1.f / du.v[1], 1.f / du.v[2])};
}
static inline Point at(const Ray& r, float t) { return r.dir.mult(t).add(r.orig); }  // raytrace.rs:227-229

// ------------------------------------------------------------------ triangle precompute
namespace {
struct Sol { bool ok; float t1, t2; };
// raytrace.rs:212-224 on two chosen coordinates (i, j)
Sol solve2(const Ray& s, const Ray& r, int i, int j) {
    const float det = r.dir.v[i] * s.dir.v[j] - r.dir.v[j] * s.dir.v[i];
    if (std::fabs(det) < 0.0001f) return Sol{false, 0.f, 0.f};
    const float dx = r.orig.v[i] - s.orig.v[i];
    const float dy = r.orig.v[j] - s.orig.v[j];
    return Sol{true, (dy * r.dir.v[i] - dx * r.dir.v[j]) / det, (dy * s.dir.v[i] - dx * s.dir.v[j]) / det};
}
// Ray::intersect (raytrace.rs:231-267): xy, then xz, then yz
bool meet(const Ray& s, const Ray& r, Point& out) {
    Sol q = solve2(s, r, 0, 1);
    if (!q.ok) q = solve2(s, r, 0, 2);
    if (!q.ok) q = solve2(s, r, 1, 2);
    if (!q.ok) return false;
    const Point p1 = at(s, q.t1), p2 = at(r, q.t2);
    if (p2.sub(p1).len2() < 0.01f) { out = p1; return true; }
    return false;
}
}  // namespace

Triangle make_triangle(const Vec3 (&points)[3], const SurfaceKind& surface, float edge_thickness) {
    const Vec3 &a = points[0], &b = points[1], &c = points[2];
    const Vec3 ab = b.sub(a), ac = c.sub(a), bc = c.sub(b);
    const Ray ra = make_ray(a, ac.add(ab));
    const Ray rb = make_ray(b, bc.add(ab.mult(-1.f)));
    Triangle t;
    if (!meet(ra, rb, t.incenter))
        throw std::runtime_error("make_triangle: degenerate triangle (the reference panics at raytrace.rs:357)");
    for (int k = 0; k < 3; k++) {
        const Vec3 vedge = points[(k + 1) % 3].sub(points[k]);
        const Vec3 po = t.incenter.sub(points[k]);
        const Vec3 pc = vedge.mult(vedge.dot(po) / vedge.len2());
        const Vec3 oc = pc.sub(po);
        t.sides[k] = oc.unit();
        t.side_lens[k] = oc.len();
        t.corners[k] = points[k];
    }
    t.norm = t.sides[0].cross(t.sides[1]).unit();
    float r2 = 0.0f;
    for (int k = 0; k < 3; k++) r2 = std::fmax(r2, points[k].sub(t.incenter).len2());
    t.bounding_r2 = r2;
    t.surface = surface;
    t.edge_thickness = edge_thickness;
    t.num = 0;
    return t;
}

std::vector<Triangle> make_triangles_gpu(const std::vector<Vec3>& corners, const SurfaceKind& surface, float edge_thickness,
                                         int device) {
    if (corners.size() % 3 != 0) throw std::runtime_error("make_triangles_gpu: corners must come in threes");
    const size_t n = corners.size() / 3;
    std::vector<float> pts(n * 9);
    for (size_t i = 0; i < corners.size(); i++) memcpy(&pts[i * 3], corners[i].v, 12);
    rtmi_triangle_t proto{};
    proto.edge_thickness = edge_thickness;
    proto.surface_kind = surface.tag;
    memcpy(proto.color, surface.color.v, 12);
    proto.alpha = surface.alpha; proto.scattering = surface.scattering;
    std::vector<rtmi_triangle_t> rec(n);
    if (rtmi_make_triangles(device, pts.data(), n, &proto, rec.data()) != RTMI_OK)
        throw std::runtime_error(std::string("rtmi_make_triangles: ") + rtmi_last_error());
    std::vector<Triangle> out(n);
    for (size_t i = 0; i < n; i++) {
        Triangle& t = out[i];
        t.incenter = make_vec(rec[i].incenter); t.norm = make_vec(rec[i].norm); t.bounding_r2 = rec[i].bounding_r2;
        for (int k = 0; k < 3; k++) { t.sides[k] = make_vec(rec[i].sides[k]); t.side_lens[k] = rec[i].side_lens[k]; t.corners[k] = corners[3 * i + k]; }
        t.surface = surface; t.edge_thickness = edge_thickness; t.num = 0;
    }
    return out;
}

Triangle make_dummy_triangle() {
    const Vec3 pts[3] = {make_vec(1.f, 0.f, 0.f), make_vec(0.f, 1.f, 0.f), make_vec(0.f, 0.f, 1.f)};
    return make_triangle(pts, SurfaceKind::solid(make_color(255, 0, 0)), 0.f);
}

void populate_triangle_numbers(std::vector<Triangle>& tris) {
    for (size_t i = 0; i < tris.size(); i++) tris[i].num = i;
}

static const float kPi = 3.14159265358979323846f;
static const float kHalfPi = 1.57079632679489661923f;

std::vector<Triangle> make_sphere(const Point& orig, float r, std::pair<size_t, size_t> lat_lon,
                                  const SurfaceKind& surface, float edge_thickness) {
    const size_t nlat = lat_lon.first, nlon = lat_lon.second;
    if (nlat % 2 != 0) throw std::runtime_error("make_sphere: num_lat must be even (assert at raytrace.rs:469)");
    std::vector<Triangle> tris;
    auto on_sphere = [&](float sphi, float cphi, float theta) {
        return orig.add(make_vec(r * sphi, r * cphi * std::cos(theta), r * cphi * std::sin(theta)));
    };
    for (size_t lat = 0; lat < nlat; lat++) {
        const bool even = (lat % 2 == 0);
        const float lo = (float)lat / (float)nlat * kPi, hi = (float)(lat + 1) / (float)nlat * kPi;
        const float phi1 = ((even ? lo : hi) - kHalfPi) * -1.f;
        const float phi23 = ((even ? hi : lo) - kHalfPi) * -1.f;
        const float smudge = even ? 0.f : 0.5f;
        for (size_t lon = 0; lon < nlon; lon++) {
            const float theta1 = ((float)lon + smudge) / (float)nlon * 2.f * kPi;
            const float theta2 = ((float)lon + 0.5f + smudge) / (float)nlon * 2.f * kPi;
            const float theta3 = ((float)lon - 0.5f + smudge) / (float)nlon * 2.f * kPi;
            const float theta4 = ((float)lon + 1.0f + smudge) / (float)nlon * 2.f * kPi;
            const float s1 = std::sin(phi1), c1 = std::cos(phi1), s23 = std::sin(phi23), c23 = std::cos(phi23);
            const Point p1 = on_sphere(s1, c1, theta1), p4 = on_sphere(s1, c1, theta4);
            const Point p2 = on_sphere(s23, c23, theta2), p3 = on_sphere(s23, c23, theta3);
            const Vec3 f1[3] = {p1, p2, p3};
            tris.push_back(make_triangle(f1, surface, edge_thickness));
            if (lat != 0 && lat != nlat - 1) {
                const Vec3 f2[3] = {p1, p2, p4};
                tris.push_back(make_triangle(f2, surface, edge_thickness));
            }
        }
    }
    return tris;
}

std::vector<Triangle> make_disk(const Point& orig, const Vec3& norm, float r, float d, size_t num_tris,
                                const SurfaceKind& surface, const SurfaceKind& side_surface, float edge_thickness) {
    std::vector<Triangle> tris;
    const Vec3 o0 = norm.orthogonal().unit().mult(r);
    const Vec3 o1 = norm.cross(o0).unit().mult(r);
    const Vec3 up = norm.mult(d), down = norm.mult(-1.f * d);
    auto rim = [&](const Vec3& base, float theta) { return orig.add(base).add(o0.mult(std::sin(theta))).add(o1.mult(std::cos(theta))); };
    for (size_t idx = 0; idx < num_tris; idx++) {
        const float n = (float)num_tris, i = (float)idx;
        const float theta1 = i / n * 2.f * kPi;
        const float theta2 = (i + 1.f) / n * 2.f * kPi;
        const float theta3 = (i + 0.5f) / n * 2.f * kPi;
        const float theta4 = (i + 1.5f) / n * 2.f * kPi;
        const Point p1p = orig.add(up), p2p = rim(up, theta1), p3p = rim(up, theta2);
        const Point p1m = orig.add(down), p2m = rim(down, theta3), p3m = rim(down, theta4);
        const Vec3 top[3] = {p1p, p2p, p3p}, bottom[3] = {p1m, p2m, p3m};
        const Vec3 side_a[3] = {p2p, p3p, p2m}, side_b[3] = {p2m, p3m, p3p};
        tris.push_back(make_triangle(top, surface, edge_thickness));
        tris.push_back(make_triangle(bottom, surface, edge_thickness));
        tris.push_back(make_triangle(side_a, side_surface, edge_thickness));
        tris.push_back(make_triangle(side_b, side_surface, edge_thickness));
    }
    return tris;
}

// ------------------------------------------------------------------ octree builder
static inline bool box_contains_point(const Point& orig, float len2, const Point& p) {  // raytrace.rs:636-643
    const Vec3 op = p.sub(orig);
    return std::fabs(op.v[0]) < len2 && std::fabs(op.v[1]) < len2 && std::fabs(op.v[2]) < len2;
}

bool face_contains_triangle(const Point& p, const Vec3& norm, float len2, const Triangle& t) {
    const float h1 = norm.dot(p.add(norm.mult(len2)));
    const float h2 = t.norm.dot(t.incenter);
    const float nn = norm.dot(t.norm);
    const float c1 = (h1 - h2 * nn) / (1.f - nn * nn);
    const float c2 = (h2 - h1 * nn) / (1.f - nn * nn);
    const Ray line_tmp = make_ray(norm.mult(c1).add(t.norm.mult(c2)), norm.cross(t.norm));

    // first slab pass: how far before the box does the line start (raytrace.rs:659-685)
    float tmin = FLT_MAX;
    for (int k = 0; k < 3; k++) {
        if (norm.v[k] != 0.f) continue;
        const float t1 = (p.v[k] - len2 - line_tmp.orig.v[k]) * line_tmp.inv_dir.v[k];
        const float t2 = (p.v[k] + len2 - line_tmp.orig.v[k]) * line_tmp.inv_dir.v[k];
        tmin = std::fmin(tmin, std::fmin(t1, t2));
    }
    const Ray line = (tmin > 0.f) ? line_tmp : make_ray(at(line_tmp, tmin * 2.f), line_tmp.dir);

    // second pass: clip against the two slabs of the face (raytrace.rs:687-716)
    tmin = -FLT_MAX;
    float tmax = FLT_MAX;
    for (int k = 0; k < 3; k++) {
        if (norm.v[k] != 0.f) continue;
        const float t1 = (p.v[k] - len2 - line.orig.v[k]) * line.inv_dir.v[k];
        const float t2 = (p.v[k] + len2 - line.orig.v[k]) * line.inv_dir.v[k];
        tmin = std::fmax(tmin, std::fmin(t1, t2));
        tmax = std::fmin(tmax, std::fmax(t1, t2));
    }
    if (tmax < tmin) return false;

    // does the (infinite) line separate two corners (raytrace.rs:718-728)
    Vec3 off[3];
    for (int k = 0; k < 3; k++) {
        const float tk = t.corners[k].sub(line.orig).dot(line.dir) / line.dir.len2();
        off[k] = at(line, tk).sub(t.corners[k]);
    }
    return off[0].dot(off[1]) < 0.f || off[0].dot(off[2]) < 0.f || off[1].dot(off[2]) < 0.f;
}

bool box_contains_polygon(const Point& orig, float len2, const Triangle& t) {
    if (box_contains_point(orig, len2, t.incenter)) return true;
    for (int k = 0; k < 3; k++)
        if (box_contains_point(orig, len2, t.corners[k])) return true;
    static const float axes[6][3] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
    for (int k = 0; k < 6; k++)
        if (face_contains_triangle(orig, make_vec(axes[k]), len2, t)) return true;
    return false;
}

size_t BoundingBox::num_inner() const { size_t n = 0; for (auto& b : boxes) n += b.is_leaf ? 0 : 1; return n; }
size_t BoundingBox::num_leaves() const { size_t n = 0; for (auto& b : boxes) n += b.is_leaf ? 1 : 0; return n; }
size_t BoundingBox::max_depth() const { size_t d = 0; for (auto& b : boxes) d = std::max<size_t>(d, b.depth); return d; }

BoundingBox build_empty_box() {
    BoundingBox bb;
    bb.boxes.push_back(rtmi_box_t{{0.f, 0.f, 0.f}, 1.f, 0, 0, 1, 0});
    return bb;
}

BoundingBox build_trivial_bounding_box(const std::vector<Triangle>& tris, const Point& orig, float len2) {
    BoundingBox bb;
    for (size_t i = 1; i < tris.size(); i++) bb.tri_refs.push_back((uint32_t)i);
    bb.boxes.push_back(rtmi_box_t{{orig.v[0], orig.v[1], orig.v[2]}, len2, 0, (uint32_t)bb.tri_refs.size(), 1, 0});
    return bb;
}

namespace {
// Level-synchronous restatement of build_bounding_box_helper (raytrace.rs:795-845):
// a box filters its parent's surviving list, becomes a leaf when small or at
// maxdepth, otherwise gets 8 candidate children; a box whose list is empty, or
// whose children all vanish, does not exist.  Every level is filtered in
// parallel over (box, candidate-chunk) work items; survivor order is the
// candidate order, so the result does not depend on the thread count.
struct Tmp {
    Point orig;
    float len2;
    uint32_t depth;
    int32_t parent;            // index in the previous level
    std::vector<uint32_t> objs;  // survivors
    bool leaf = false, alive = false;
    int32_t child[8] = {-1, -1, -1, -1, -1, -1, -1, -1};  // index in the next level
    uint32_t flat = 0;
};

template <typename F>
void parallel_for(size_t n, unsigned threads, F&& f) {
    if (threads <= 1 || n <= 1) { for (size_t i = 0; i < n; i++) f(i); return; }
    std::atomic<size_t> next(0);
    auto work = [&]() { for (;;) { size_t i = next.fetch_add(1); if (i >= n) break; f(i); } };
    std::vector<std::thread> th;
    const unsigned nt = (unsigned)std::min<size_t>(threads, n);
    for (unsigned k = 1; k < nt; k++) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
}
}  // namespace

namespace {
BoundingBox build_bounding_box_impl(const std::vector<Triangle>& tris, const Point& orig, float len2, size_t maxdepth,
                                    size_t minobjs, unsigned threads, rtmi_builder_t* gpu);
}

BoundingBox build_bounding_box(const std::vector<Triangle>& tris, const Point& orig, float len2, size_t maxdepth,
                               size_t minobjs, unsigned threads) {
    return build_bounding_box_impl(tris, orig, len2, maxdepth, minobjs, threads, nullptr);
}

// The same tree with the overlap tests of every level evaluated on the GPU (rtmi_builder_filter, k_box_contains):
// the flags are bit-identical to box_contains_polygon() above, the list bookkeeping stays here.
BoundingBox build_bounding_box_gpu(const std::vector<Triangle>& tris, const Point& orig, float len2, size_t maxdepth,
                                   size_t minobjs, int device) {
    std::vector<float> rec(tris.size() * 15);
    for (size_t i = 0; i < tris.size(); i++) {
        float* o = rec.data() + i * 15;
        memcpy(o, tris[i].incenter.v, 12); memcpy(o + 3, tris[i].norm.v, 12);
        for (int k = 0; k < 3; k++) memcpy(o + 6 + 3 * k, tris[i].corners[k].v, 12);
    }
    rtmi_builder_t* b = nullptr;
    if (rtmi_builder_create(device, rec.data(), tris.size(), &b) != RTMI_OK)
        throw std::runtime_error(std::string("rtmi_builder_create: ") + rtmi_last_error());
    try {
        BoundingBox bb = build_bounding_box_impl(tris, orig, len2, maxdepth, minobjs, 1, b);
        rtmi_builder_destroy(b);
        return bb;
    } catch (...) {
        rtmi_builder_destroy(b);
        throw;
    }
}

namespace {
BoundingBox build_bounding_box_impl(const std::vector<Triangle>& tris, const Point& orig, float len2, size_t maxdepth,
                                    size_t minobjs, unsigned threads, rtmi_builder_t* gpu) {
    if (threads == 0) threads = std::max(1u, std::thread::hardware_concurrency());
    if (tris.size() >= (1ull << 30)) throw std::runtime_error("build_bounding_box: too many triangles");
    std::vector<std::vector<Tmp>> levels;
    std::vector<uint32_t> all;
    for (size_t i = 1; i < tris.size(); i++) all.push_back((uint32_t)i);  // triangle 0 is the sentinel (raytrace.rs:791)

    levels.emplace_back(1);
    levels[0][0].orig = orig; levels[0][0].len2 = len2; levels[0][0].depth = 0; levels[0][0].parent = -1;

    const size_t CH = 512;
    for (size_t d = 0;; d++) {
        std::vector<Tmp>& cur = levels[d];
        // work items: (box, chunk of its candidate list)
        struct Item { uint32_t box; uint32_t lo, hi; };
        std::vector<Item> items;
        std::vector<std::vector<uint8_t>> keep(cur.size());
        for (size_t b = 0; b < cur.size(); b++) {
            const std::vector<uint32_t>& cand = (d == 0) ? all : levels[d - 1][cur[b].parent].objs;
            keep[b].assign(cand.size(), 0);
            for (size_t lo = 0; lo < cand.size(); lo += CH)
                items.push_back(Item{(uint32_t)b, (uint32_t)lo, (uint32_t)std::min(cand.size(), lo + CH)});
        }
        if (gpu) {
            // one device pass per level: the candidate lists (one per PARENT: its 8 children share it) are concatenated,
            // every box names its parent's range, the flags come back in box order
            std::vector<uint32_t> cand_flat;
            std::vector<uint64_t> first_of_parent(d == 0 ? 1 : levels[d - 1].size(), UINT64_MAX);
            std::vector<rtmi_build_box_t> bxs(cur.size());
            uint64_t nkeep = 0;
            for (size_t b = 0; b < cur.size(); b++) {
                const size_t par = d == 0 ? 0 : (size_t)cur[b].parent;
                const std::vector<uint32_t>& cand = (d == 0) ? all : levels[d - 1][par].objs;
                if (first_of_parent[par] == UINT64_MAX) {
                    first_of_parent[par] = cand_flat.size();
                    cand_flat.insert(cand_flat.end(), cand.begin(), cand.end());
                }
                if (cand_flat.size() >= (1ull << 32)) throw std::runtime_error("build_bounding_box_gpu: level above 2^32 candidates");
                rtmi_build_box_t& r = bxs[b];
                for (int k = 0; k < 3; k++) r.orig[k] = cur[b].orig.v[k];
                r.len2 = cur[b].len2;
                r.cand_first = (uint32_t)first_of_parent[par];
                r.cand_count = (uint32_t)cand.size();
                r.keep_first = nkeep;
                nkeep += cand.size();
            }
            std::vector<uint8_t> flat(nkeep);
            if (rtmi_builder_filter(gpu, bxs.data(), bxs.size(), cand_flat.data(), cand_flat.size(), flat.data(), nkeep) != RTMI_OK)
                throw std::runtime_error(std::string("rtmi_builder_filter: ") + rtmi_last_error());
            for (size_t b = 0; b < cur.size(); b++)
                memcpy(keep[b].data(), flat.data() + bxs[b].keep_first, keep[b].size());
        } else
        parallel_for(items.size(), threads, [&](size_t k) {
            const Item& it = items[k];
            const Tmp& bx = cur[it.box];
            const std::vector<uint32_t>& cand = (d == 0) ? all : levels[d - 1][bx.parent].objs;
            for (uint32_t j = it.lo; j < it.hi; j++)
                keep[it.box][j] = box_contains_polygon(bx.orig, bx.len2, tris[cand[j]]) ? 1 : 0;
        });
        std::vector<Tmp> next;
        for (size_t b = 0; b < cur.size(); b++) {
            Tmp& bx = cur[b];
            const std::vector<uint32_t>& cand = (d == 0) ? all : levels[d - 1][bx.parent].objs;
            for (size_t j = 0; j < cand.size(); j++)
                if (keep[b][j]) bx.objs.push_back(cand[j]);
            if (bx.objs.empty()) continue;                       // None
            if (bx.objs.size() < minobjs || bx.depth >= maxdepth) {  // leaf
                bx.leaf = true; bx.alive = true;
                continue;
            }
            const float newlen2 = bx.len2 / 2.f;
            for (int i = 0; i < 8; i++) {
                const float xoff = ((i & 1) == 0) ? -1.f * newlen2 : newlen2;
                const float yoff = ((i & 2) == 0) ? -1.f * newlen2 : newlen2;
                const float zoff = ((i & 4) == 0) ? -1.f * newlen2 : newlen2;
                Tmp c;
                c.orig = bx.orig.add(make_vec(xoff, yoff, zoff));
                c.len2 = newlen2; c.depth = bx.depth + 1; c.parent = (int32_t)b;
                bx.child[i] = (int32_t)next.size();
                next.push_back(std::move(c));
            }
        }
        if (next.empty()) break;
        levels.push_back(std::move(next));
    }
    // a box with no surviving child does not exist (raytrace.rs:835-844)
    for (size_t d = levels.size(); d-- > 0;)
        for (Tmp& bx : levels[d]) {
            if (bx.leaf || bx.objs.empty()) continue;
            for (int i = 0; i < 8; i++)
                if (bx.child[i] >= 0 && levels[d + 1][bx.child[i]].alive) bx.alive = true;
        }
    if (!levels[0][0].alive)
        throw std::runtime_error("build_bounding_box: no triangle inside the root box (the reference panics at raytrace.rs:792)");

    // breadth-first flattening, children of a box contiguous, in octant order
    BoundingBox out;
    std::vector<std::pair<uint32_t, uint32_t>> order;  // (level, index)
    order.push_back({0, 0});
    for (size_t head = 0; head < order.size(); head++) {
        Tmp& bx = levels[order[head].first][order[head].second];
        bx.flat = (uint32_t)head;
        if (bx.leaf) continue;
        for (int i = 0; i < 8; i++)
            if (bx.child[i] >= 0 && levels[order[head].first + 1][bx.child[i]].alive)
                order.push_back({order[head].first + 1, (uint32_t)bx.child[i]});
    }
    out.boxes.resize(order.size());
    uint32_t next_child = 1;
    for (size_t k = 0; k < order.size(); k++) {
        const Tmp& bx = levels[order[k].first][order[k].second];
        rtmi_box_t rb{{bx.orig.v[0], bx.orig.v[1], bx.orig.v[2]}, bx.len2, 0, 0, bx.leaf ? 1u : 0u, bx.depth};
        if (bx.leaf) {
            rb.first = (uint32_t)out.tri_refs.size();
            rb.count = (uint32_t)bx.objs.size();
            out.tri_refs.insert(out.tri_refs.end(), bx.objs.begin(), bx.objs.end());
        } else {
            uint32_t n = 0;
            for (int i = 0; i < 8; i++)
                if (bx.child[i] >= 0 && levels[order[k].first + 1][bx.child[i]].alive) n++;
            rb.first = next_child;
            rb.count = n;
            next_child += n;
        }
        out.boxes[k] = rb;
    }
    return out;
}
}  // namespace

// ------------------------------------------------------------------ camera
float to_radians(float deg) { return deg * (kPi / 180.0f); }

std::tuple<Vec3, Vec3, Vec3> create_transform(const Vec3& dir_in, float d_roll) {
    const Vec3 dir = dir_in.unit();
    const float roll = -1.f * std::atan2(-1.f * dir.v[1], dir.v[2]);
    const float pitch = -1.f * std::asin(dir.v[0]);
    const float yaw = -1.f * d_roll;
    const float cy = std::cos(yaw), sy = std::sin(yaw), cp = std::cos(pitch), sp = std::sin(pitch);
    const float cr = std::cos(roll), sr = std::sin(roll);
    return std::make_tuple(make_vec(cy * cp, sy * cp, -1.f * sp),
                           make_vec(cy * sp * sr - sy * cr, sy * sp * sr + cy * cr, cp * sr),
                           make_vec(cy * sp * cr + sy * sr, sy * sp * cr - cy * sr, cp * cr));
}

Viewport create_viewport(std::pair<uint32_t, uint32_t> px, std::pair<float, float> size, const Point& pos,
                         const Vec3& dir, float fov, float c_roll, size_t maxdepth, size_t samples) {
    const float dist = size.first / (2.f * std::tan(to_radians(fov) / 2.f));
    const auto rot = create_transform(dir, c_roll);
    Viewport v;
    v.width = px.first; v.height = px.second;
    v.orig = pos.add(make_vec(1.f * size.second / 2.f, -1.f * size.first / 2.f, 0.f));
    v.cam = pos.sub(make_vec(0.f, 0.f, dist).change_basis(rot));
    v.vu = make_vec(0.f, size.first, 0.f).change_basis(rot);
    v.vv = make_vec(-1.f * size.second, 0.f, 0.f).change_basis(rot);
    v.maxdepth = maxdepth; v.samples_per_pixel = samples;
    return v;
}

// ------------------------------------------------------------------ the plug-in
std::string ProgressCtx::stats_line() const {  // progress.rs:158-162
    char buf[256];
    const double m = (double)total_rays / 1e6;
    snprintf(buf, sizeof buf, "Processed %.3f million rays in %.3f seconds. %.3f million rays/s", m, seconds,
             seconds > 0 ? m / seconds : 0.0);
    return buf;
}

ProgressCtx RayCaster::walk_rays(const Viewport& v, const Scene& s, Color* data, size_t threads, bool /*show_progress*/) {
    ProgressCtx ctx;
    const auto t0 = std::chrono::steady_clock::now();   // progress::create_ctx (progress.rs:80)
    walk_rays_internal(v, s, data, threads, ctx);
    const auto t1 = std::chrono::steady_clock::now();   // ProgressCtx::finish (progress.rs:146)
    ctx.seconds = std::chrono::duration<double>(t1 - t0).count();
    return ctx;
}

void flatten_triangles(const std::vector<Triangle>& tris, std::vector<rtmi_triangle_t>& out) {
    out.resize(tris.size());
    for (size_t i = 0; i < tris.size(); i++) {
        const Triangle& t = tris[i];
        rtmi_triangle_t& r = out[i];
        for (int k = 0; k < 3; k++) {
            r.incenter[k] = t.incenter.v[k];
            r.norm[k] = t.norm.v[k];
            r.side_lens[k] = t.side_lens[k];
            r.color[k] = t.surface.color.v[k];
            for (int j = 0; j < 3; j++) r.sides[k][j] = t.sides[k].v[j];
        }
        r.bounding_r2 = t.bounding_r2;
        r.edge_thickness = t.edge_thickness;
        r.surface_kind = t.surface.tag;
        r.alpha = t.surface.alpha;
        r.scattering = t.surface.scattering;
    }
}

rtmi_viewport_t to_abi(const Viewport& v) {
    rtmi_viewport_t a;
    a.width = (uint32_t)v.width; a.height = (uint32_t)v.height;
    for (int k = 0; k < 3; k++) { a.orig[k] = v.orig.v[k]; a.cam[k] = v.cam.v[k]; a.vu[k] = v.vu.v[k]; a.vv[k] = v.vv.v[k]; }
    a.maxdepth = (uint32_t)v.maxdepth; a.samples_per_pixel = (uint32_t)v.samples_per_pixel;
    return a;
}

HipRayCaster::HipRayCaster(uint64_t seed_, int device_) : seed(seed_), device(device_) {}
HipRayCaster::~HipRayCaster() { invalidate(); }
void HipRayCaster::invalidate() {
    if (handle_) rtmi_scene_destroy(handle_);
    for (rtmi_scene_t* h : extra_) rtmi_scene_destroy(h);
    extra_.clear();
    handle_ = nullptr; key_scene_ = nullptr;
}

void HipRayCaster::set_devices(const std::vector<int>& devices) {
    if (devices == devices_) return;
    invalidate();
    devices_ = devices;
    if (!devices_.empty()) device = devices_[0];
}

void HipRayCaster::apply_settings() {
    rtmi_scene_set_options(handle_, options_);
    for (rtmi_scene_t* h : extra_) rtmi_scene_set_options(h, options_);
    rtmi_tuning_t t = defaults_;  // what rtmi_scene_create chose (environment or built-in)
    // every handle of a multi-device caster gets the same tuning (and the same reset to the defaults)
    auto set_all = [&](const rtmi_tuning_t& tt) {
        if (rtmi_scene_set_tuning(handle_, &tt) != RTMI_OK) throw std::runtime_error(std::string("rtmi_scene_set_tuning: ") + rtmi_last_error());
        for (rtmi_scene_t* h : extra_)
            if (rtmi_scene_set_tuning(h, &tt) != RTMI_OK) throw std::runtime_error(std::string("rtmi_scene_set_tuning: ") + rtmi_last_error());
    };
    if (!has_tuning_) { set_all(t); return; }
    // 0 = keep the library default (xcd_aware, where 0 is a value, is passed as given + 1)
    if (tuning_.batch_paths) t.batch_paths = tuning_.batch_paths;
    if (tuning_.streams) t.streams = tuning_.streams;
    if (tuning_.subtile_min_paths) t.subtile_min_paths = tuning_.subtile_min_paths;
    if (tuning_.oct_waves_per_cu) t.oct_waves_per_cu = tuning_.oct_waves_per_cu;
    if (tuning_.refill_min0) t.refill_min0 = tuning_.refill_min0;
    if (tuning_.refill_min) t.refill_min = tuning_.refill_min;
    if (tuning_.xcd_aware) t.xcd_aware = tuning_.xcd_aware - 1;
    if (tuning_.kernel) t.kernel = tuning_.kernel;
    if (tuning_.pipeline) t.pipeline = tuning_.pipeline;
    if (tuning_.slow_path_off) t.slow_path_off = tuning_.slow_path_off;
    set_all(t);
}

rtmi_scene_t* HipRayCaster::resident(const Scene& s) {
    if (handle_ && key_scene_ == &s && key_generation_ == s.generation && key_ntris_ == s.tris.size() &&
        key_nboxes_ == s.boxes.boxes.size() && key_nrefs_ == s.boxes.tri_refs.size()) {
        apply_settings();
        return handle_;
    }
    invalidate();
    if (s.boxes.boxes.empty()) throw std::runtime_error("Scene has no bounding box");
    std::vector<rtmi_triangle_t> flat;
    flatten_triangles(s.tris, flat);
    rtmi_scene_t* h = nullptr;
    const int rc = rtmi_scene_create(flat.data(), flat.size(), s.boxes.boxes.data(), s.boxes.boxes.size(),
                                     s.boxes.tri_refs.data(), s.boxes.tri_refs.size(), device, &h);
    if (rc != RTMI_OK) throw std::runtime_error(std::string("rtmi_scene_create: ") + rtmi_last_error());
    handle_ = h;
    for (size_t k = 1; k < devices_.size(); k++) {  // one more resident copy per extra device
        rtmi_scene_t* e = nullptr;
        if (rtmi_scene_create(flat.data(), flat.size(), s.boxes.boxes.data(), s.boxes.boxes.size(), s.boxes.tri_refs.data(),
                              s.boxes.tri_refs.size(), devices_[k], &e) != RTMI_OK) {
            const std::string msg = std::string("rtmi_scene_create (device ") + std::to_string(devices_[k]) + "): " + rtmi_last_error();
            invalidate();
            throw std::runtime_error(msg);
        }
        extra_.push_back(e);
    }
    {   // the corners (`Triangle.corners`): tighter boxes for the opt-in fast mode; the exact modes do not use them
        std::vector<float> corners(9 * s.tris.size());
        for (size_t i = 0; i < s.tris.size(); i++)
            for (int k = 0; k < 3; k++)
                for (int a = 0; a < 3; a++) corners[9 * i + 3 * k + a] = s.tris[i].corners[k].v[a];
        std::vector<rtmi_scene_t*> all{handle_};
        all.insert(all.end(), extra_.begin(), extra_.end());
        for (rtmi_scene_t* hh : all)
            if (rtmi_scene_set_corners(hh, corners.data(), s.tris.size()) != RTMI_OK) {
                const std::string msg = std::string("rtmi_scene_set_corners: ") + rtmi_last_error();
                invalidate();
                throw std::runtime_error(msg);
            }
    }
    if (!s.spheres.empty()) {
        std::vector<rtmi_sphere_t> sp(s.spheres.size());
        for (size_t i = 0; i < sp.size(); i++) {
            const Sphere& q = s.spheres[i];
            for (int k = 0; k < 3; k++) { sp[i].center[k] = q.center.v[k]; sp[i].color[k] = q.surface.color.v[k]; }
            sp[i].radius = q.radius; sp[i].surface_kind = q.surface.tag; sp[i].alpha = q.surface.alpha; sp[i].scattering = q.surface.scattering;
        }
        std::vector<rtmi_scene_t*> all{handle_};
        all.insert(all.end(), extra_.begin(), extra_.end());
        for (rtmi_scene_t* hh : all)
            if (rtmi_scene_set_spheres(hh, sp.data(), sp.size()) != RTMI_OK) {
                const std::string msg = std::string("rtmi_scene_set_spheres: ") + rtmi_last_error();
                invalidate();
                throw std::runtime_error(msg);
            }
    }
    rtmi_scene_get_tuning(handle_, &defaults_);
    key_scene_ = &s; key_generation_ = s.generation; key_ntris_ = s.tris.size();
    key_nboxes_ = s.boxes.boxes.size(); key_nrefs_ = s.boxes.tri_refs.size();
    apply_settings();
    return handle_;
}

void HipRayCaster::walk_rows_device(const Viewport& v, const Scene& s, size_t row0, size_t nrows, void* out_device,
                                    void* hip_stream, ProgressCtx& progress) {
    const rtmi_tile_t tile{(uint32_t)row0, (uint32_t)nrows, nrows ? (uint32_t)nrows : 1u, 0u};
    walk_tile_device(v, s, tile, out_device, hip_stream, progress);
}

void HipRayCaster::walk_tile_device(const Viewport& v, const Scene& s, const rtmi_tile_t& tile, void* out_device,
                                    void* hip_stream, ProgressCtx& progress) {
    rtmi_scene_t* h = resident(s);
    const rtmi_viewport_t av = to_abi(v);
    rtmi_stats_t st;
    const int rc = rtmi_render_tile_device(h, &av, seed, &tile, out_device, hip_stream, &st);
    if (rc != RTMI_OK) throw std::runtime_error(std::string("rtmi_render_tile_device: ") + rtmi_last_error());
    progress.total_rays += st.rays;
    progress.kernel_seconds += st.kernel_ms * 1e-3;
    progress.stats = st;
}

void HipRayCaster::walk_rows(const Viewport& v, const Scene& s, size_t row0, size_t nrows, Color* data, ProgressCtx& progress) {
    rtmi_scene_t* h = resident(s);
    const rtmi_viewport_t av = to_abi(v);
    rtmi_stats_t st;
    static_assert(sizeof(Color) == 16, "Color must be 4 floats");
    const int rc = rtmi_render(h, &av, seed, (uint32_t)row0, (uint32_t)nrows, reinterpret_cast<float*>(data), &st);
    if (rc != RTMI_OK) throw std::runtime_error(std::string("rtmi_render: ") + rtmi_last_error());
    progress.total_rays += st.rays;
    progress.kernel_seconds += st.kernel_ms * 1e-3;
    progress.stats = st;
}

void HipRayCaster::walk_frame_multi(const Viewport& v, const Scene& s, void* data_host, void* data_device, uint32_t stripe_rows,
                                    uint32_t flags, ProgressCtx& progress, std::vector<rtmi_stats_t>* per_device) {
    rtmi_scene_t* h = resident(s);
    std::vector<rtmi_scene_t*> hs{h};
    hs.insert(hs.end(), extra_.begin(), extra_.end());
    const rtmi_viewport_t av = to_abi(v);
    std::vector<rtmi_stats_t> st(hs.size());
    const int rc = rtmi_render_frame_multi(hs.data(), (uint32_t)hs.size(), &av, seed, stripe_rows, flags, data_host, data_device, st.data());
    if (rc != RTMI_OK) throw std::runtime_error(std::string("rtmi_render_frame_multi: ") + rtmi_last_error());
    rtmi_stats_t sum{};
    sum.peer_access = 1;  // 1 only when every device reaches the root directly
    for (const rtmi_stats_t& d : st) {
        if (!d.peer_access) sum.peer_access = 0;
        sum.rays += d.rays; sum.box_tests += d.box_tests; sum.tri_tests += d.tri_tests; sum.full_tests += d.full_tests;
        sum.nodes += d.nodes; sum.leaves += d.leaves; sum.trace_ms += d.trace_ms; sum.trace_launches += d.trace_launches;
        sum.kernel_ms = std::max(sum.kernel_ms, d.kernel_ms);  // the devices run concurrently
        sum.streams = std::max(sum.streams, d.streams);
        sum.render_ms = std::max(sum.render_ms, d.render_ms);
        sum.primary_ms += d.primary_ms; sum.bounce_ms += d.bounce_ms; sum.pipeline = std::max(sum.pipeline, d.pipeline); sum.slow_paths += d.slow_paths;
        sum.band_copy_ms = std::max(sum.band_copy_ms, d.band_copy_ms);
        sum.deinterleave_ms += d.deinterleave_ms;
    }
    progress.total_rays += sum.rays;
    progress.kernel_seconds += sum.kernel_ms * 1e-3;
    progress.stats = sum;
    if (per_device) *per_device = st;
}

// DefaultRayCaster fans rows out over `threads` CPU threads (raytrace.rs:1175-1196); here the fan-out is over the
// caster's devices, inside the library.  `threads` is ignored like the reference's CudaRayCaster does.
void HipRayCaster::walk_rays_internal(const Viewport& v, const Scene& s, Color* data, size_t /*threads*/, ProgressCtx& progress) {
    if (devices_.size() > 1) { walk_frame_multi(v, s, data, nullptr, 0, 0, progress); return; }
    if (!on_progress_) { walk_rows(v, s, 0, v.height, data, progress); return; }
    // progress while rendering: one render call and one tuple per row band (raytrace.rs:1411, :1429-1435)
    const size_t bands = std::min(progress_bands_, std::max<size_t>(v.height, 1));
    rtmi_stats_t sum{};
    for (size_t b = 0; b < bands; b++) {
        const size_t r0 = v.height * b / bands, r1 = v.height * (b + 1) / bands;
        if (r1 == r0) continue;
        ProgressCtx band;
        walk_rows(v, s, r0, r1 - r0, data + r0 * v.width, band);
        progress.total_rays += band.total_rays;
        progress.kernel_seconds += band.kernel_seconds;
        sum.rays += band.stats.rays; sum.box_tests += band.stats.box_tests; sum.tri_tests += band.stats.tri_tests;
        sum.full_tests += band.stats.full_tests; sum.nodes += band.stats.nodes; sum.leaves += band.stats.leaves;
        sum.kernel_ms += band.stats.kernel_ms; sum.trace_ms += band.stats.trace_ms; sum.trace_launches += band.stats.trace_launches;
        sum.primary_ms += band.stats.primary_ms; sum.bounce_ms += band.stats.bounce_ms; sum.slow_paths += band.stats.slow_paths;
        sum.streams = std::max(sum.streams, band.stats.streams); sum.pipeline = band.stats.pipeline;
        on_progress_(0, r1 - 1, (r1 - r0) * v.width, band.total_rays);
    }
    progress.stats = sum;
}

void quantize_rgb8(const Color* data, size_t npixels, uint8_t* rgb) {
    for (size_t i = 0; i < npixels; i++)
        for (int k = 0; k < 3; k++) {
            const float x = data[i].v[k] * 255.f;
            rgb[i * 3 + k] = (x != x) ? 0 : (x <= 0.f ? 0 : (x >= 255.f ? 255 : (uint8_t)x));  // Rust `as u8`
        }
}

}  // namespace raytrace
