// raytrace.hpp — host-side C++ mirror of the reference's scene/camera API
// (raytrace_lib/src/raytrace.rs, obj_parser.rs) plus the `RayCaster` plug-in
// whose MI355X implementation (`HipRayCaster`) calls the C ABI of
// include/rtmi.h.  In a real integration this layer stays in Rust (see
// INTEGRATION.md); it exists here because the image has no Rust toolchain.
//
// Same names, argument meaning and error behaviour as the reference; where the
// reference panics (`unwrap`, `assert!`) these functions throw
// std::runtime_error.  All arithmetic is strict IEEE f32 in the reference's
// operation order (compile with -ffp-contract=off).
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../../../include/rtmi.h"

namespace raytrace {

// raytrace.rs:22-122 — Simd<f32,4>, lane 3 carried like the reference does.
struct Vec3 {
    float v[4];
    Vec3 add(const Vec3& o) const;
    Vec3 sub(const Vec3& o) const;
    Vec3 mult(float a) const;
    Vec3 mult_per(const Vec3& o) const;
    float len2() const;
    float len() const;
    float dot(const Vec3& o) const;
    Vec3 cross(const Vec3& o) const;
    Vec3 unit() const;
    Vec3 orthogonal() const;
    Vec3 change_basis(const std::tuple<Vec3, Vec3, Vec3>& b) const;
};
using Point = Vec3;
using Color = Vec3;

Vec3 make_vec(const float (&v)[3]);
inline Vec3 make_vec(float a, float b, float c) { const float t[3] = {a, b, c}; return make_vec(t); }
Color make_color(uint8_t r, uint8_t g, uint8_t b);  // raytrace.rs:176-180

// raytrace.rs:194-210
struct Ray { Point orig; Vec3 dir; Vec3 inv_dir; };
Ray make_ray(const Point& orig, const Vec3& dir);

// raytrace.rs:303-308
struct SurfaceKind {
    enum Tag : uint32_t { Solid = RTMI_SOLID, Matte = RTMI_MATTE, Reflective = RTMI_REFLECTIVE } tag;
    Color color;
    float alpha;
    float scattering;
    static SurfaceKind solid(const Color& c) { return SurfaceKind{Solid, c, 0.f, 0.f}; }
    static SurfaceKind matte(const Color& c, float alpha) { return SurfaceKind{Matte, c, alpha, 0.f}; }
    static SurfaceKind reflective(float scattering, const Color& c, float alpha) { return SurfaceKind{Reflective, c, alpha, scattering}; }
};

// raytrace.rs:326-337
struct Triangle {
    Point incenter;
    Vec3 norm;
    float bounding_r2;
    Vec3 sides[3];
    float side_lens[3];
    Vec3 corners[3];
    SurfaceKind surface;
    float edge_thickness;
    size_t num;
};

Triangle make_triangle(const Vec3 (&points)[3], const SurfaceKind& surface, float edge_thickness);  // :340-383
// The same for many triangles on the GPU (rtmi_make_triangles): corners[i] = 3 points; bit-identical results.
std::vector<Triangle> make_triangles_gpu(const std::vector<Vec3>& corners, const SurfaceKind& surface, float edge_thickness,
                                         int device = 0);
Triangle make_dummy_triangle();                                                                      // :385-391
void populate_triangle_numbers(std::vector<Triangle>& tris);                                         // :393-397
std::vector<Triangle> make_sphere(const Point& orig, float r, std::pair<size_t, size_t> lat_lon,
                                  const SurfaceKind& surface, float edge_thickness);                 // :464-529
std::vector<Triangle> make_disk(const Point& orig, const Vec3& norm, float r, float d, size_t num_tris,
                                const SurfaceKind& surface, const SurfaceKind& side_surface,
                                float edge_thickness);                                               // :531-592

// raytrace.rs:612-634.  The reference's recursive `BoundingBox` is kept
// flattened, in the layout the C ABI transports (breadth-first, children of a
// box contiguous, triangle indices of a leaf contiguous).
struct BoundingBox {
    std::vector<rtmi_box_t> boxes;  // boxes[0] = root
    std::vector<uint32_t> tri_refs;
    size_t num_inner() const;
    size_t num_leaves() const;
    size_t max_depth() const;
};

bool box_contains_polygon(const Point& orig, float len2, const Triangle& t);                         // :753-779
bool face_contains_triangle(const Point& p, const Vec3& norm, float len2, const Triangle& t);        // :645-729
BoundingBox build_empty_box();                                                                       // :781-788
BoundingBox build_bounding_box(const std::vector<Triangle>& tris, const Point& orig, float len2,
                               size_t maxdepth, size_t minobjs, unsigned threads = 0);               // :790-845
// The same tree, the box/triangle overlap tests of every level evaluated on the GPU (rtmi_builder_*); bit-equal result.
BoundingBox build_bounding_box_gpu(const std::vector<Triangle>& tris, const Point& orig, float len2, size_t maxdepth,
                                   size_t minobjs, int device = 0);
BoundingBox build_trivial_bounding_box(const std::vector<Triangle>& tris, const Point& orig, float len2);  // :847-856

// Analytic sphere: NOT in the reference at this revision (only Triangle is Collidable, raytrace.rs:399; make_sphere
// tessellates).  A build-defined extension named by BASELINE's north_star; semantics in include/rtmi.h (rtmi_sphere_t)
// and, operation by operation, in DESIGN.md 4.6.  Parity with the Rust binary: unpinned.
struct Sphere {
    Point center;
    float radius;
    SurfaceKind surface;
};

// raytrace.rs:1297-1303 (debug_ctx / debug_en: out of scope)
struct Scene {
    std::vector<Triangle> tris;
    BoundingBox boxes;
    std::vector<Sphere> spheres;  // analytic spheres: a flat list tested against every ray after the box tree
    // Bumped by touch(): a HipRayCaster keeps the uploaded copy of a Scene resident and re-uploads when the
    // generation it saw differs.  Code that edits tris/boxes in place must call touch() afterwards.
    uint64_t generation = 0;
    void touch() { generation++; }
};

// raytrace.rs:1305-1318
struct Viewport {
    size_t width, height;
    Point orig, cam;
    Vec3 vu, vv;
    size_t maxdepth, samples_per_pixel;
};
std::tuple<Vec3, Vec3, Vec3> create_transform(const Vec3& dir_in, float d_roll);                     // :1320-1341
Viewport create_viewport(std::pair<uint32_t, uint32_t> px, std::pair<float, float> size, const Point& pos,
                         const Vec3& dir, float fov, float c_roll, size_t maxdepth, size_t samples);  // :1343-1370
float to_radians(float deg);

// progress.rs:95-184 reduced to what print_stats reports.
struct ProgressCtx {
    uint64_t total_rays = 0;
    double seconds = 0.0;        // wall time of walk_rays (create_ctx -> finish)
    double kernel_seconds = 0.0; // device time of the render kernels
    rtmi_stats_t stats{};
    std::string stats_line() const;  // "Processed X million rays in Y seconds. Z million rays/s"
};

// raytrace.rs:1128-1165
class RayCaster {
public:
    virtual ~RayCaster() = default;
    virtual void walk_rays_internal(const Viewport& v, const Scene& s, Color* data, size_t threads, ProgressCtx& progress) = 0;
    ProgressCtx walk_rays(const Viewport& v, const Scene& s, Color* data, size_t threads, bool show_progress);
};

// The MI355X implementation of the plug-in.  `threads` is ignored like the
// reference's CudaRayCaster does (cuda_raytrace.rs:546-572).  Keeps the
// uploaded scene resident between calls on the same Scene object.
class HipRayCaster : public RayCaster {
public:
    explicit HipRayCaster(uint64_t seed = 1, int device = 0);
    ~HipRayCaster() override;
    HipRayCaster(const HipRayCaster&) = delete;
    HipRayCaster& operator=(const HipRayCaster&) = delete;
    void walk_rays_internal(const Viewport& v, const Scene& s, Color* data, size_t threads, ProgressCtx& progress) override;
    // Render only rows [row0, row0+nrows): the unit of multi-GPU image tiling.
    void walk_rows(const Viewport& v, const Scene& s, size_t row0, size_t nrows, Color* data, ProgressCtx& progress);
    void walk_rows_device(const Viewport& v, const Scene& s, size_t row0, size_t nrows, void* out_device,
                          void* hip_stream, ProgressCtx& progress);
    // Multi-GPU inside one process (rtmi_render_frame_multi): the frame is striped over `devices` (one uploaded copy
    // of the scene per entry; an entry may repeat a device), bands are copied once to devices[0] and de-interleaved
    // there.  With more than one entry walk_rays_internal() takes this path.  flags: RTMI_FRAME_RGB8 -> `data` is
    // height*width*3 bytes.
    void set_devices(const std::vector<int>& devices);
    void walk_frame_multi(const Viewport& v, const Scene& s, void* data_host, void* data_device, uint32_t stripe_rows,
                          uint32_t flags, ProgressCtx& progress, std::vector<rtmi_stats_t>* per_device = nullptr);
    // Striped row set (rtmi_tile_t): rank r of N renders {r*S, H/N, S, N*S}.
    void walk_tile_device(const Viewport& v, const Scene& s, const rtmi_tile_t& tile, void* out_device,
                          void* hip_stream, ProgressCtx& progress);
    void set_options(uint32_t opts) { options_ = opts; }
    // Progress while rendering, as DefaultRayCaster reports it (raytrace.rs:1411, :1429-1435 send one tuple per finished
    // row / per 10 k rays: (thread, row, pixels done, {"Rays": n}); progress.rs:95-141 draws from them).  With a callback
    // set, walk_rays_internal renders the frame in `bands` row bands (default 16) and calls it after each one with
    // (thread = 0, last row of the band, pixels of the band, rays of the band).  Same image; a band is a render call of
    // its own (its launches do not overlap the next band's), so a frame costs a few per cent more than in one piece.
    using ProgressFn = std::function<void(size_t thread, size_t row, size_t pixels, uint64_t rays)>;
    void set_progress(ProgressFn fn, size_t bands = 16) { on_progress_ = std::move(fn); progress_bands_ = bands ? bands : 1; }
    // Launch tuning (rtmi_tuning_t); fields left 0 keep the library defaults.  Never changes a pixel.
    void set_tuning(const rtmi_tuning_t& t) { tuning_ = t; has_tuning_ = true; }
    void clear_tuning() { has_tuning_ = false; }
    // Uploads on first use, and again when another Scene object is passed or Scene::generation changed.
    rtmi_scene_t* resident(const Scene& s);
    void invalidate();
    uint64_t seed;
    int device;

private:
    rtmi_scene_t* handle_ = nullptr;
    std::vector<int> devices_;                 // multi-GPU: device of every extra handle (entry 0 = `device`)
    std::vector<rtmi_scene_t*> extra_;         // handles of devices_[1..]
    const Scene* key_scene_ = nullptr;
    uint64_t key_generation_ = 0;
    size_t key_ntris_ = 0, key_nboxes_ = 0, key_nrefs_ = 0;
    uint32_t options_ = 0;
    rtmi_tuning_t tuning_{}, defaults_{};
    bool has_tuning_ = false;
    ProgressFn on_progress_;
    size_t progress_bands_ = 16;
    void apply_settings();
};

void flatten_triangles(const std::vector<Triangle>& tris, std::vector<rtmi_triangle_t>& out);
rtmi_viewport_t to_abi(const Viewport& v);

namespace obj_parser {
// obj_parser.rs:47-73.  ObjMode::Reference (default) is the reference's loader exactly: first three corners of every
// face, 1-based positive indices.  ObjMode::Robust is an opt-in extension: polygons are fan-triangulated, negative
// (relative) indices are resolved, degenerate triangles are skipped.
enum class ObjMode { Reference = 0, Robust = 1 };
std::vector<Triangle> parse_obj(const std::string& path, const Vec3& offset, float scale,
                                const std::tuple<Vec3, Vec3, Vec3>& transform, const SurfaceKind& surface,
                                float edge_thickness, ObjMode mode = ObjMode::Reference);
}  // namespace obj_parser

// raytrace.rs:1460-1478: quantisation only ((c*255.) as u8); PNG encoding stays with the caller.
void quantize_rgb8(const Color* data, size_t npixels, uint8_t* rgb);

}  // namespace raytrace
