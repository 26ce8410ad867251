// capi.cpp — extern "C" view (include/rtmi_host.h) of the C++ host mirror.
#include <chrono>
#include <cstring>
#include <memory>

#include "../../../include/rtmi_host.h"
#include "raytrace.hpp"

using namespace raytrace;

struct rth_scene {
    Scene scene;
    std::unique_ptr<HipRayCaster> caster;
    uint32_t options = 0;
};

static thread_local std::string g_herr;
const char* rth_last_error(void) { return g_herr.c_str(); }

template <typename F>
static int guarded(F&& f) {
    try { f(); return 0; }
    catch (const std::exception& e) { g_herr = e.what(); return 1; }
    catch (...) { g_herr = "unknown C++ exception"; return 1; }
}
static Vec3 v3(const float* p) { return make_vec(p[0], p[1], p[2]); }
static SurfaceKind surf(uint32_t kind, const float* c, float alpha, float scat) {
    if (kind > RTMI_REFLECTIVE) throw std::runtime_error("unknown surface kind");
    return SurfaceKind{(SurfaceKind::Tag)kind, v3(c), alpha, scat};
}
static HipRayCaster& caster_of(rth_scene* s) {
    if (!s->caster) s->caster.reset(new HipRayCaster(1, 0));
    return *s->caster;
}
static Viewport vp_from(uint32_t w, uint32_t h, const float* vp12, uint64_t maxdepth, uint64_t spp) {
    Viewport v;
    v.width = w; v.height = h;
    v.orig = v3(vp12); v.cam = v3(vp12 + 3); v.vu = v3(vp12 + 6); v.vv = v3(vp12 + 9);
    v.maxdepth = (size_t)maxdepth; v.samples_per_pixel = (size_t)spp;
    return v;
}

extern "C" {

void rth_make_color(uint8_t r, uint8_t g, uint8_t b, float* o) { Color c = make_color(r, g, b); memcpy(o, c.v, 12); }
void rth_unit(const float* in3, float* o) { Vec3 c = v3(in3).unit(); memcpy(o, c.v, 12); }
float rth_to_radians(float deg) { return to_radians(deg); }
void rth_create_transform(const float* dir3, float d_roll, float* out9) {
    auto b = create_transform(v3(dir3), d_roll);
    memcpy(out9, std::get<0>(b).v, 12); memcpy(out9 + 3, std::get<1>(b).v, 12); memcpy(out9 + 6, std::get<2>(b).v, 12);
}
void rth_create_viewport(uint32_t w, uint32_t h, float size0, float size1, const float* pos3, const float* dir3, float fov,
                         float c_roll, float* out12) {
    Viewport v = create_viewport({w, h}, {size0, size1}, v3(pos3), v3(dir3), fov, c_roll, 1, 1);
    memcpy(out12, v.orig.v, 12); memcpy(out12 + 3, v.cam.v, 12); memcpy(out12 + 6, v.vu.v, 12); memcpy(out12 + 9, v.vv.v, 12);
}

rth_scene_t* rth_scene_new(int with_dummy) {
    rth_scene* s = new rth_scene();
    if (with_dummy) s->scene.tris.push_back(make_dummy_triangle());
    return s;
}
void rth_scene_free(rth_scene_t* s) { delete s; }
uint64_t rth_num_tris(const rth_scene_t* s) { return s->scene.tris.size(); }

int rth_add_triangle(rth_scene_t* s, const float* p, uint32_t kind, const float* c, float alpha, float scat, float edge) {
    return guarded([&] {
        const Vec3 pts[3] = {v3(p), v3(p + 3), v3(p + 6)};
        s->scene.tris.push_back(make_triangle(pts, surf(kind, c, alpha, scat), edge));
        s->scene.touch();
    });
}
int rth_add_obj(rth_scene_t* s, const char* path, const float* off, float scale, const float* b9, uint32_t kind,
                const float* c, float alpha, float scat, float edge) {
    return rth_add_obj_mode(s, path, off, scale, b9, kind, c, alpha, scat, edge, 0);
}
int rth_add_obj_mode(rth_scene_t* s, const char* path, const float* off, float scale, const float* b9, uint32_t kind,
                     const float* c, float alpha, float scat, float edge, uint32_t robust) {
    return guarded([&] {
        auto t = obj_parser::parse_obj(path, v3(off), scale, std::make_tuple(v3(b9), v3(b9 + 3), v3(b9 + 6)),
                                       surf(kind, c, alpha, scat), edge, robust ? obj_parser::ObjMode::Robust : obj_parser::ObjMode::Reference);
        s->scene.tris.insert(s->scene.tris.end(), t.begin(), t.end());
        s->scene.touch();
    });
}
int rth_add_disk(rth_scene_t* s, const float* orig3, const float* norm3, float r, float d, uint64_t n, uint32_t kind,
                 const float* c, float alpha, float scat, uint32_t skind, const float* sc, float salpha, float sscat, float edge) {
    return guarded([&] {
        auto t = make_disk(v3(orig3), v3(norm3), r, d, (size_t)n, surf(kind, c, alpha, scat), surf(skind, sc, salpha, sscat), edge);
        s->scene.tris.insert(s->scene.tris.end(), t.begin(), t.end());
        s->scene.touch();
    });
}
int rth_add_sphere(rth_scene_t* s, const float* orig3, float r, uint64_t nlat, uint64_t nlon, uint32_t kind, const float* c,
                   float alpha, float scat, float edge) {
    return guarded([&] {
        auto t = make_sphere(v3(orig3), r, {(size_t)nlat, (size_t)nlon}, surf(kind, c, alpha, scat), edge);
        s->scene.tris.insert(s->scene.tris.end(), t.begin(), t.end());
        s->scene.touch();
    });
}
int rth_add_triangles_gpu(rth_scene_t* s, const float* p, uint64_t n, uint32_t kind, const float* c, float alpha, float scat,
                          float edge, int device) {
    return guarded([&] {
        std::vector<Vec3> corners(n * 3);
        for (uint64_t i = 0; i < n * 3; i++) corners[i] = v3(p + 3 * i);
        auto t = make_triangles_gpu(corners, surf(kind, c, alpha, scat), edge, device);
        s->scene.tris.insert(s->scene.tris.end(), t.begin(), t.end());
        s->scene.touch();
    });
}
int rth_add_analytic_sphere(rth_scene_t* s, const float* center3, float r, uint32_t kind, const float* c, float alpha, float scat) {
    return guarded([&] {
        s->scene.spheres.push_back(Sphere{v3(center3), r, surf(kind, c, alpha, scat)});
        s->scene.touch();
    });
}
void rth_populate_triangle_numbers(rth_scene_t* s) { populate_triangle_numbers(s->scene.tris); s->scene.touch(); }

int rth_build_bounding_box(rth_scene_t* s, const float* orig3, float len2, uint64_t maxdepth, uint64_t minobjs, uint32_t threads) {
    return guarded([&] {
        s->scene.boxes = build_bounding_box(s->scene.tris, v3(orig3), len2, (size_t)maxdepth, (size_t)minobjs, threads);
        s->scene.touch();
    });
}
int rth_build_bounding_box_gpu(rth_scene_t* s, const float* orig3, float len2, uint64_t maxdepth, uint64_t minobjs, int device) {
    return guarded([&] {
        s->scene.boxes = build_bounding_box_gpu(s->scene.tris, v3(orig3), len2, (size_t)maxdepth, (size_t)minobjs, device);
        s->scene.touch();
    });
}
int rth_build_trivial_bounding_box(rth_scene_t* s, const float* orig3, float len2) {
    return guarded([&] {
        s->scene.boxes = build_trivial_bounding_box(s->scene.tris, v3(orig3), len2);
        s->scene.touch();
    });
}
int rth_box_contains_polygon(const rth_scene_t* s, const float* orig3, float len2, uint64_t tri) {
    return box_contains_polygon(v3(orig3), len2, s->scene.tris.at(tri)) ? 1 : 0;
}
int rth_face_contains_triangle(const rth_scene_t* s, const float* p3, const float* norm3, float len2, uint64_t tri) {
    return face_contains_triangle(v3(p3), v3(norm3), len2, s->scene.tris.at(tri)) ? 1 : 0;
}

void rth_get_triangles(const rth_scene_t* s, float* out, int32_t* kinds, float* sf) {
    const auto& tris = s->scene.tris;
    for (size_t i = 0; i < tris.size(); i++) {
        const Triangle& t = tris[i];
        float* o = out + i * 29;
        memcpy(o, t.incenter.v, 12); memcpy(o + 3, t.norm.v, 12);
        o[6] = t.bounding_r2;
        for (int k = 0; k < 3; k++) { memcpy(o + 7 + 3 * k, t.sides[k].v, 12); o[16 + k] = t.side_lens[k]; memcpy(o + 20 + 3 * k, t.corners[k].v, 12); }
        o[19] = t.edge_thickness;
        kinds[i] = (int32_t)t.surface.tag;
        memcpy(sf + i * 5, t.surface.color.v, 12);
        sf[i * 5 + 3] = t.surface.alpha; sf[i * 5 + 4] = t.surface.scattering;
    }
}
void rth_tree_sizes(const rth_scene_t* s, uint64_t* nboxes, uint64_t* nrefs) {
    *nboxes = s->scene.boxes.boxes.size(); *nrefs = s->scene.boxes.tri_refs.size();
}
void rth_tree_get(const rth_scene_t* s, float* geo, uint32_t* topo, uint32_t* refs) {
    const auto& bb = s->scene.boxes;
    for (size_t i = 0; i < bb.boxes.size(); i++) {
        const rtmi_box_t& b = bb.boxes[i];
        memcpy(geo + i * 4, b.orig, 12); geo[i * 4 + 3] = b.len2;
        topo[i * 4] = b.first; topo[i * 4 + 1] = b.count; topo[i * 4 + 2] = b.is_leaf; topo[i * 4 + 3] = b.depth;
    }
    if (!bb.tri_refs.empty()) memcpy(refs, bb.tri_refs.data(), bb.tri_refs.size() * 4);
}

int rth_caster_config(rth_scene_t* s, uint64_t seed, int device, uint32_t options) {
    return guarded([&] {
        HipRayCaster& c = caster_of(s);
        if (c.device != device) { c.invalidate(); c.device = device; }
        c.seed = seed;
        c.set_options(options);
    });
}
int rth_caster_set_devices(rth_scene_t* s, const int32_t* devices, uint32_t n) {
    return guarded([&] { caster_of(s).set_devices(std::vector<int>(devices, devices + n)); });
}
int rth_caster_walk_frame_multi(rth_scene_t* s, uint32_t w, uint32_t h, const float* vp12, uint64_t maxdepth, uint64_t spp,
                                uint32_t stripe_rows, uint32_t flags, void* out_host, void* out_device, rtmi_stats_t* stats,
                                rtmi_stats_t* per_device, uint32_t per_device_cap, double* wall) {
    return guarded([&] {
        const Viewport v = vp_from(w, h, vp12, maxdepth, spp);
        ProgressCtx ctx;
        std::vector<rtmi_stats_t> pd;
        const auto t0 = std::chrono::steady_clock::now();
        caster_of(s).walk_frame_multi(v, s->scene, out_host, out_device, stripe_rows, flags, ctx, &pd);
        if (wall) *wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (stats) *stats = ctx.stats;
        for (uint32_t k = 0; per_device && k < per_device_cap && k < pd.size(); k++) per_device[k] = pd[k];
    });
}
int rth_caster_set_tuning(rth_scene_t* s, const rtmi_tuning_t* t) {
    return guarded([&] {
        if (t) caster_of(s).set_tuning(*t);
        else caster_of(s).clear_tuning();
    });
}
int rth_caster_upload(rth_scene_t* s) { return guarded([&] { caster_of(s).resident(s->scene); }); }

int rth_caster_walk_rows(rth_scene_t* s, uint32_t w, uint32_t h, const float* vp12, uint64_t maxdepth, uint64_t spp,
                         uint64_t row0, uint64_t nrows, float* out_host, rtmi_stats_t* stats, double* wall) {
    return guarded([&] {
        const Viewport v = vp_from(w, h, vp12, maxdepth, spp);
        ProgressCtx ctx;
        const auto t0 = std::chrono::steady_clock::now();
        caster_of(s).walk_rows(v, s->scene, (size_t)row0, (size_t)nrows, reinterpret_cast<Color*>(out_host), ctx);
        if (wall) *wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (stats) *stats = ctx.stats;
    });
}
int rth_caster_walk_rows_device(rth_scene_t* s, uint32_t w, uint32_t h, const float* vp12, uint64_t maxdepth, uint64_t spp,
                                uint64_t row0, uint64_t nrows, void* out_device, void* hip_stream, rtmi_stats_t* stats, double* wall) {
    return guarded([&] {
        const Viewport v = vp_from(w, h, vp12, maxdepth, spp);
        ProgressCtx ctx;
        const auto t0 = std::chrono::steady_clock::now();
        caster_of(s).walk_rows_device(v, s->scene, (size_t)row0, (size_t)nrows, out_device, hip_stream, ctx);
        if (wall) *wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (stats) *stats = ctx.stats;
    });
}
int rth_caster_walk_tile_device(rth_scene_t* s, uint32_t w, uint32_t h, const float* vp12, uint64_t maxdepth, uint64_t spp,
                                const rtmi_tile_t* tile, void* out_device, void* hip_stream, rtmi_stats_t* stats, double* wall) {
    return guarded([&] {
        const Viewport v = vp_from(w, h, vp12, maxdepth, spp);
        ProgressCtx ctx;
        const auto t0 = std::chrono::steady_clock::now();
        caster_of(s).walk_tile_device(v, s->scene, *tile, out_device, hip_stream, ctx);
        if (wall) *wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (stats) *stats = ctx.stats;
    });
}
int rth_caster_trace(rth_scene_t* s, uint64_t n, const float* o4, const float* d4, uint32_t* tri, float* t, uint32_t* face,
                     rtmi_stats_t* stats) {
    return guarded([&] {
        rtmi_scene_t* h = caster_of(s).resident(s->scene);
        if (rtmi_trace(h, n, o4, d4, tri, t, face, stats) != RTMI_OK)
            throw std::runtime_error(std::string("rtmi_trace: ") + rtmi_last_error());
    });
}

int rth_caster_quantize_device(rth_scene_t* s, const void* rgba_device, uint64_t npixels, void* rgb_device, void* hip_stream) {
    return guarded([&] {
        if (rtmi_quantize_device(caster_of(s).resident(s->scene), rgba_device, npixels, rgb_device, hip_stream) != RTMI_OK)
            throw std::runtime_error(std::string("rtmi_quantize_device: ") + rtmi_last_error());
    });
}

// development aid (tools/step_stats.py); not declared in rtmi_host.h
int rtmi_debug_counters(rtmi_scene_t* s, unsigned long long* out16);
int rth_debug_counters(rth_scene_t* s, unsigned long long* out16) {
    return guarded([&] { rtmi_debug_counters(caster_of(s).resident(s->scene), out16); });
}

void rth_quantize(const float* rgba, uint64_t npixels, uint8_t* rgb) {
    quantize_rgb8(reinterpret_cast<const Color*>(rgba), (size_t)npixels, rgb);
}

}  // extern "C"
