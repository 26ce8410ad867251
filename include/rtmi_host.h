/* rtmi_host.h — C view of the host-side mirror of the reference's scene API
 * (rust_raytrace_amd/csrc/host/raytrace.hpp) for language bindings (ctypes).
 *
 * This is NOT part of the drop-in boundary: in a real integration everything
 * declared here stays in the Rust `raytrace_lib` crate (make_triangle,
 * make_disk, parse_obj, build_bounding_box, create_viewport, the RayCaster
 * trait) and only include/rtmi.h is bound.  It exists because this image has
 * no Rust toolchain; the C++ mirror keeps the reference's names and semantics
 * so that tests read like the reference's own call sites (raytrace/src/main.rs).
 *
 * Functions returning int return 0 on success; the message of a failure (the
 * places where the reference panics) is read with rth_last_error().
 */
#ifndef RTMI_HOST_H
#define RTMI_HOST_H
#include <stdint.h>
#include "rtmi.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct rth_scene rth_scene_t; /* raytrace::Scene + a HipRayCaster bound to it */

const char* rth_last_error(void);

/* raytrace.rs:176-180, :93-96, f32::to_radians, :1320-1341, :1343-1370 */
void rth_make_color(uint8_t r, uint8_t g, uint8_t b, float* out3);
void rth_unit(const float* in3, float* out3);
float rth_to_radians(float deg);
void rth_create_transform(const float* dir3, float d_roll, float* out9);
void rth_create_viewport(uint32_t w, uint32_t h, float size0, float size1, const float* pos3, const float* dir3,
                         float fov, float c_roll, float* out12 /* orig cam vu vv */);

rth_scene_t* rth_scene_new(int with_dummy /* push make_dummy_triangle() first, main.rs:117 */);
void rth_scene_free(rth_scene_t* s);
uint64_t rth_num_tris(const rth_scene_t* s);

/* make_triangle / parse_obj / make_disk / make_sphere appended to Scene.tris */
int rth_add_triangle(rth_scene_t* s, const float* pts9, uint32_t kind, const float* color3, float alpha, float scattering, float edge);
int rth_add_obj(rth_scene_t* s, const char* path, const float* offset3, float scale, const float* basis9,
                uint32_t kind, const float* color3, float alpha, float scattering, float edge);
/* The same with robust != 0: opt-in loader extension (fan-triangulated polygons, negative indices, degenerate
 * triangles skipped).  robust == 0 is the reference's loader (first three corners, obj_parser.rs:63-65). */
int rth_add_obj_mode(rth_scene_t* s, const char* path, const float* offset3, float scale, const float* basis9, uint32_t kind,
                     const float* color3, float alpha, float scattering, float edge_thickness, uint32_t robust);
int rth_add_disk(rth_scene_t* s, const float* orig3, const float* norm3, float r, float d, uint64_t num_tris,
                 uint32_t kind, const float* color3, float alpha, float scattering,
                 uint32_t side_kind, const float* side_color3, float side_alpha, float side_scattering, float edge);
int rth_add_sphere(rth_scene_t* s, const float* orig3, float r, uint64_t num_lat, uint64_t num_lon,
                   uint32_t kind, const float* color3, float alpha, float scattering, float edge);
/* n triangles at once through the GPU make_triangle kernel (rtmi_make_triangles) */
int rth_add_triangles_gpu(rth_scene_t* s, const float* pts9, uint64_t n, uint32_t kind, const float* color3, float alpha,
                          float scattering, float edge, int device);
/* analytic sphere: a build-defined extension (rtmi_sphere_t in rtmi.h), not a reference API */
int rth_add_analytic_sphere(rth_scene_t* s, const float* center3, float radius, uint32_t kind, const float* color3, float alpha,
                            float scattering);
void rth_populate_triangle_numbers(rth_scene_t* s);

/* build_bounding_box / build_trivial_bounding_box into Scene.boxes */
int rth_build_bounding_box(rth_scene_t* s, const float* orig3, float len2, uint64_t maxdepth, uint64_t minobjs, uint32_t threads);
/* The same tree with every level's box/triangle overlap tests on the GPU (rtmi_builder_*); bit-equal to the host build. */
int rth_build_bounding_box_gpu(rth_scene_t* s, const float* orig3, float len2, uint64_t maxdepth, uint64_t minobjs, int device);
int rth_build_trivial_bounding_box(rth_scene_t* s, const float* orig3, float len2);
int rth_box_contains_polygon(const rth_scene_t* s, const float* orig3, float len2, uint64_t tri);
int rth_face_contains_triangle(const rth_scene_t* s, const float* p3, const float* norm3, float len2, uint64_t tri);

/* inspection: 29 floats per triangle (incenter3 norm3 r2 sides9 side_lens3 edge corners9), kind, (color3 alpha scattering) */
void rth_get_triangles(const rth_scene_t* s, float* rec29, int32_t* kinds, float* surf5);
void rth_tree_sizes(const rth_scene_t* s, uint64_t* nboxes, uint64_t* nrefs);
void rth_tree_get(const rth_scene_t* s, float* geo4, uint32_t* topo4 /* first count is_leaf depth */, uint32_t* refs);

/* HipRayCaster (implements RayCaster, raytrace.rs:1128-1165) */
int rth_caster_config(rth_scene_t* s, uint64_t seed, int device, uint32_t rtmi_options);
int rth_caster_walk_rows(rth_scene_t* s, uint32_t w, uint32_t h, const float* vp12, uint64_t maxdepth, uint64_t spp,
                         uint64_t row0, uint64_t nrows, float* out_host, rtmi_stats_t* stats, double* wall_seconds);
int rth_caster_walk_rows_device(rth_scene_t* s, uint32_t w, uint32_t h, const float* vp12, uint64_t maxdepth, uint64_t spp,
                                uint64_t row0, uint64_t nrows, void* out_device, void* hip_stream, rtmi_stats_t* stats,
                                double* wall_seconds);
int rth_caster_walk_tile_device(rth_scene_t* s, uint32_t w, uint32_t h, const float* vp12, uint64_t maxdepth, uint64_t spp,
                                const rtmi_tile_t* tile, void* out_device, void* hip_stream, rtmi_stats_t* stats,
                                double* wall_seconds);
int rth_caster_trace(rth_scene_t* s, uint64_t n, const float* orig4, const float* dir4, uint32_t* tri, float* t,
                     uint32_t* face, rtmi_stats_t* stats);
/* Multi-GPU inside the process (rtmi_render_frame_multi): the caster keeps one resident copy of the scene per entry of
 * `devices` (an entry may repeat a device); entry 0 is the root that receives the bands.  With more than one entry
 * walk_rays (rth_caster_walk_rows over the whole image) and rth_caster_walk_frame_multi stripe the frame over them. */
int rth_caster_set_devices(rth_scene_t* s, const int32_t* devices, uint32_t n);
int rth_caster_walk_frame_multi(rth_scene_t* s, uint32_t w, uint32_t h, const float* vp12, uint64_t maxdepth, uint64_t spp,
                                uint32_t stripe_rows, uint32_t flags /* RTMI_FRAME_RGB8 */, void* out_host, void* out_device,
                                rtmi_stats_t* stats_sum, rtmi_stats_t* per_device, uint32_t per_device_cap, double* wall_seconds);
int rth_caster_upload(rth_scene_t* s);
/* Launch tuning for this scene's caster: fields that are 0 keep the library default, xcd_aware is passed as value + 1;
 * NULL restores all defaults.  Never changes a pixel. */
int rth_caster_set_tuning(rth_scene_t* s, const rtmi_tuning_t* tuning);
int rth_caster_quantize_device(rth_scene_t* s, const void* rgba_device, uint64_t npixels, void* rgb_device, void* hip_stream); /* make the scene resident now (otherwise on first use) */

/* write_png's quantisation on the host (raytrace.rs:1468-1473) */
void rth_quantize(const float* rgba, uint64_t npixels, uint8_t* rgb);

#ifdef __cplusplus
}
#endif
#endif
