/* rtmi.h — C ABI of the MI355X path-tracing core (librtmi.so).
 *
 * Drop-in boundary for rust_raytrace's `RayCaster` plug-in
 * (raytrace_lib/src/raytrace.rs:1128-1165).  A Rust `impl RayCaster for
 * HipRayCaster` flattens `Scene` once, calls rtmi_scene_create(), then
 * rtmi_render() from walk_rays_internal(); see INTEGRATION.md for the shim.
 *
 * Plain C: pointers and sizes only, no C++/torch types.  Every function
 * returns RTMI_OK (0) or an error code; the message for the calling thread is
 * available from rtmi_last_error().  No C++ exception crosses this boundary.
 * All functions are callable from any host thread (the reference enters its
 * caster from a scoped worker thread, raytrace.rs:1141-1146); one in-flight
 * render per scene handle.
 */
#ifndef RTMI_H
#define RTMI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    RTMI_OK = 0,
    RTMI_ERR_INVALID = 1,      /* bad argument / malformed scene            */
    RTMI_ERR_NO_DEVICE = 2,    /* no HIP device visible / bad device index  */
    RTMI_ERR_UNSUPPORTED = 3,  /* valid input outside what the kernels take */
    RTMI_ERR_OOM = 4,          /* host or device allocation failed          */
    RTMI_ERR_DEVICE = 5        /* any other HIP runtime / kernel failure    */
};

/* SurfaceKind (raytrace.rs:303-308) */
enum { RTMI_SOLID = 0, RTMI_MATTE = 1, RTMI_REFLECTIVE = 2 };

/* One `Triangle` (raytrace.rs:326-337) reduced to the fields the hot path
 * reads (intersects/normal/getsurface, raytrace.rs:399-461).  Lane 3 of every
 * Vec3 is +0 for records produced by make_triangle (raytrace.rs:340-383) and
 * is not transported. */
typedef struct rtmi_triangle {
    float incenter[3];
    float norm[3];
    float bounding_r2;
    float sides[3][3];
    float side_lens[3];
    float edge_thickness;
    uint32_t surface_kind; /* RTMI_SOLID / RTMI_MATTE / RTMI_REFLECTIVE */
    float color[3];
    float alpha;      /* Matte, Reflective */
    float scattering; /* Reflective */
} rtmi_triangle_t;

/* One `BoundingBox` (raytrace.rs:618-623), flattened.  Inner box: its children
 * are boxes[first .. first+count) in the order of `BBSubobj::Boxes`.  Leaf:
 * its triangle indices are tri_refs[first .. first+count) in the order of
 * `BBSubobj::Tris`.  boxes[0] is the root (`Scene.boxes`). */
typedef struct rtmi_box {
    float orig[3];
    float len2; /* half edge length */
    uint32_t first;
    uint32_t count;
    uint32_t is_leaf;
    uint32_t depth;
} rtmi_box_t;

/* `Viewport` (raytrace.rs:1305-1318).  orig/cam/vu/vv are private in the
 * reference; the shim recomputes them with create_viewport's formula
 * (raytrace.rs:1343-1370) or the host crate makes them `pub`. */
typedef struct rtmi_viewport {
    uint32_t width, height;
    float orig[3], cam[3], vu[3], vv[3];
    uint32_t maxdepth;
    uint32_t samples_per_pixel;
} rtmi_viewport_t;

/* Work counters of one render/trace call.  `rays` is the reference's "Rays"
 * statistic: project_ray calls with depth > 0 (raytrace.rs:1278). */
typedef struct rtmi_stats {
    uint64_t rays;
    uint64_t box_tests, tri_tests, full_tests, nodes, leaves; /* filled only when counting is enabled */
    double kernel_ms;  /* device time span of the call (HIP events on the caller's stream)               */
    double trace_ms;   /* sum of the durations of the closest-hit launches (HIP events on their streams)  */
    uint32_t trace_launches;
    uint32_t streams;  /* internal streams that ran concurrently (1..4): launches overlap when > 1       */
    /* rtmi_render_frame_multi only (0 elsewhere): host wall-clock milliseconds of this scene's steps, so that a slow
     * link or a refused peer mapping is visible per device instead of only in the frame time */
    double render_ms;       /* rtmi_render_tile_device of this scene's tile                                  */
    double band_copy_ms;    /* (quantise +) the band's one crossing to the root device, synchronised           */
    double deinterleave_ms; /* scenes[0] only: k_deinterleave (+ the copy to out_host)                         */
    /* fused path pipeline (rtmi_tuning_t.pipeline): the two kernels of trace_ms apart, summed over streams and batches */
    double primary_ms;      /* k_path_primary launches (pixel_ray + closest hit + color_ray of the primary rays)  */
    double bounce_ms;       /* k_path_bounce launches (every bounce of every path, shaded in place)               */
    int32_t peer_access;    /* 1 = this device writes the root device's memory directly (peer access enabled, or the
                             * same device); 0 = the runtime refused: the band is staged (rtmi_last_error() carries a
                             * warning although the call returns RTMI_OK)                                      */
    uint32_t pipeline;      /* which pipeline rendered (rtmi_tuning_t.pipeline: 1, 2 or 3)                         */
    uint32_t slow_paths;    /* paths handed to k_path_slow (rtmi_tuning_t.slow_path_off)                           */
    uint32_t reserved;
} rtmi_stats_t;

/* A set of image rows: `nrows` rows taken in stripes of `stripe_rows`
 * consecutive rows, the k-th stripe starting at row0 + k*stripe_step.
 * {row0, nrows, nrows, 0} is the contiguous band [row0, row0+nrows).
 * Interleaved stripes are how a frame is tiled over the GPUs of a node: rank r
 * of N takes {r*S, H/N, S, N*S}; cost per row is very uneven (sky vs teapot). */
typedef struct rtmi_tile {
    uint32_t row0, nrows, stripe_rows, stripe_step;
} rtmi_tile_t;

/* Analytic sphere.  NOT part of the reference at this revision (its only `Collidable` is `Triangle`, raytrace.rs:399;
 * spheres are tessellated by make_sphere, raytrace.rs:464-529); BASELINE's north_star names an analytic ray-sphere
 * test, so this build defines one -- parity with the Rust binary is unpinned by construction.  Semantics (DESIGN.md 4.6 states
 * them operation by operation): standard quadratic in the reference's Vec3 arithmetic, `t < 0` is a miss like
 * for triangles, the far root counts as a Back-face hit from inside, normal = (point - center).unit(); no edge faces.
 * A scene's spheres are a flat list: every ray is tested against every sphere AFTER the box tree and a sphere replaces
 * the tree's hit iff it is strictly closer.  Reported hit index = ntris + sphere index.
 * Intended, not an oversight: there is no origin-primitive exclusion and no epsilon, exactly as for the reference's
 * triangles (`t < 0` is the only rejection, raytrace.rs:402-405; bounce origins are moved 0.001 along the NEW direction's
 * random part, raytrace.rs:284-296, which can leave them a rounding error inside the surface).  A bounce ray whose
 * rounded origin lies just inside its sphere therefore re-hits it from inside at t ~ 0 (Back face), as a reference
 * bounce ray can re-hit the plane of the triangle it left; Matte/Reflective spheres are darker for it. */
typedef struct rtmi_sphere {
    float center[3];
    float radius;
    uint32_t surface_kind; /* RTMI_SOLID / RTMI_MATTE / RTMI_REFLECTIVE */
    float color[3];
    float alpha;
    float scattering;
} rtmi_sphere_t;

typedef struct rtmi_scene rtmi_scene_t;

/* Number of visible HIP devices (0 when none); never fails. */
int rtmi_device_count(void);

/* Upload a scene to `device` and keep it resident until rtmi_scene_destroy().
 * tris[0] is the never-rendered sentinel (raytrace.rs:791, :849); a hit index
 * of 0 means "miss".  Replaces the per-batch cudaMalloc/cudaMemcpy of
 * exec_cuda_raytrace (cuda_raytrace_lib/src/cuda_rt.cu:381-425). */
int rtmi_scene_create(const rtmi_triangle_t* tris, uint64_t ntris,
                      const rtmi_box_t* boxes, uint64_t nboxes,
                      const uint32_t* tri_refs, uint64_t nrefs,
                      int device, rtmi_scene_t** out);
int rtmi_scene_destroy(rtmi_scene_t* scene);

/* Replace the scene's list of analytic spheres (n may be 0).  At most 4096 spheres (they are not in the tree). */
int rtmi_scene_set_spheres(rtmi_scene_t* scene, const rtmi_sphere_t* spheres, uint64_t n);

/* Optional: the corners the triangle records were made from (`Triangle.corners`, raytrace.rs:326-337), 9 floats per
 * triangle, n = ntris (entry 0 = the sentinel's, ignored).  Only RTMI_OPT_BVH uses them: its boxes become the triangles'
 * own boxes (intersected with the bounding-radius disc's) instead of the disc's alone -- fewer boxes per ray, same hits.
 * The corners must be the ones the records came from (make_triangle, raytrace.rs:340-383); the exact modes ignore them. */
int rtmi_scene_set_corners(rtmi_scene_t* scene, const float* corners9, uint64_t n);

/* Option switches (all default 0): */
enum {
    RTMI_OPT_COUNTERS = 1u << 0, /* fill box/tri/node counters in rtmi_stats_t (slower) */
    RTMI_OPT_GENERIC = 1u << 1,  /* force the generic-tree traversal kernel              */
    RTMI_OPT_FAST = 1u << 2,     /* NOT bit-exact: skip boxes entirely behind the ray origin (octree kernel only).
                                  * The reference visits them; results differ only where a hit would have been found
                                  * first through such a box (exact ties between triangles, rays exactly parallel to a
                                  * triangle's plane).  Measured on config 3 (2048x2048 @ 64 spp): 17 of 4 194 304
                                  * pixels differ from exact mode, 1.6x the rays/s.  Never the default.              */
    RTMI_OPT_BVH = 1u << 3       /* "fast mode", NOT the reference's octree traversal: the closest hit over ALL triangles
                                  * with the lowest index winning exact ties, i.e. what the reference computes for a
                                  * build_trivial_bounding_box scene (raytrace.rs:847-856, :1012-1050), found through a
                                  * 4-wide SAH BVH the library builds over the triangles at scene creation (the
                                  * boxes/tri_refs passed in are ignored for tracing).  Bit-equal to the linear-list
                                  * render except for the reference's t = +-inf / NaN "hits" of triangles a ray does not
                                  * come near; differs from the octree render where the octree builder lost a triangle
                                  * or two triangles tie.  Never the default, never the headline.                     */
};
int rtmi_scene_set_options(rtmi_scene_t* scene, uint32_t options);

/* Launch tuning of one scene handle.  Defaults are taken ONCE, at rtmi_scene_create(), from the environment
 * (RTMI_BATCH_PATHS, RTMI_STREAMS, RTMI_SUBTILE_MIN_PATHS, RTMI_OCT_WAVES_PER_CU, RTMI_REFILL_MIN0,
 * RTMI_REFILL_MIN, RTMI_XCD_AWARE, RTMI_KERNEL, RTMI_PIPELINE; RTMI_VERBOSE=1 prints per-pass timings to stderr) and can be read and changed
 * here.  None of them changes a pixel: any batch size, stream count or stripe split gives the same image. */
typedef struct rtmi_tuning {
    uint64_t batch_paths;       /* paths (pixel samples) per batch of the wavefront pipeline, all streams together; default 256 Mi.
                                   A tile up to 1/8 larger is still rendered as one batch.                    */
    uint32_t streams;           /* 1..4 internal HIP streams (interleaved sub-tiles of a tile); 0 (default) = automatic:
                                   one stream for tiles of 2^26 paths and more, three below                    */
    uint32_t subtile_min_paths; /* tiles with fewer paths are not split over streams; default 32768           */
    uint32_t oct_waves_per_cu;  /* persistent waves per CU and launch of the octree kernel; 0 = automatic: what
                                   fits with one stream, at most 16 when several streams share the CUs       */
    uint32_t refill_min0;       /* idle lanes before a wave refills, primary pass (64 = whole wave); default 64 */
    uint32_t refill_min;        /* the same for bounce passes (and the shading step of k_path_bounce); default 16       */
    uint32_t xcd_aware;         /* 1 = one ray-queue range per XCD (by XCC_ID), 2 = by block index, 0 = one queue (default) */
    uint32_t kernel;            /* octree closest-hit kernel: 0 = automatic, 1 = one ray per lane (k_trace_oct),
                                 * 2 = per-wave ray pool in LDS (k_trace_pool; falls back to 1 for very deep trees) */
    uint32_t pipeline;          /* 0 = automatic (= 3), 1 = one launch per bounce pass (k_gen, then k_trace* + k_shade per
                                 * pass), 2 = fused path kernels: primary rays generated, traced and shaded in one kernel
                                 * (k_path_primary), all bounces in ONE persistent kernel that shades in place
                                 * (k_path_bounce; 62 instead of 106 bytes of workspace per path), 3 = k_path_primary, then
                                 * one closest-hit + one shading launch per bounce pass.  2 and 3 apply to octree scenes;
                                 * anything else (linear list, generic tree, BVH mode, analytic spheres) runs 1.
                                 * Same image whichever runs.  Environment: RTMI_PIPELINE.                              */
    uint32_t slow_path_off;     /* 0 (default): in pipelines 2 and 3 a ray whose unit direction has an exactly-zero component
                                 * (BoundingBox::collides then skips that axis' slab, raytrace.rs:872-900: ~150 x the work of an
                                 * ordinary ray, 14 ms for the lane that traces it) is set aside and its path is traced by
                                 * k_path_slow on a side stream, one path per wave, beside the following passes; 1: such rays
                                 * are traced where they arise (they can hold a small tile's launches for ~10 % of its time).
                                 * Environment: RTMI_SLOW_PATH_OFF.                                                    */
} rtmi_tuning_t;
int rtmi_scene_get_tuning(rtmi_scene_t* scene, rtmi_tuning_t* out);
int rtmi_scene_set_tuning(rtmi_scene_t* scene, const rtmi_tuning_t* in);

/* Render image rows [row0, row0+nrows) of the viewport into `out`
 * (nrows*width*4 floats, row-major, RGB + a zero lane == `[Color]`,
 * raytrace.rs:1183, :1426).  Same pixel values for any row partition.
 * Replaces RayCaster::walk_rays_internal (raytrace.rs:1129-1131, :1175-1196)
 * with the RNG of the build (seed) injected at rand::random's call sites.
 * rtmi_render writes host memory; rtmi_render_device writes device memory on
 * `hip_stream` (a hipStream_t, or NULL for the default stream) and returns
 * after the work is enqueued and the counters are read back. */
/* Internally a tile is rendered as rtmi_tuning_t.streams sub-tiles (rows dealt out one by one), each on its own HIP stream
 * of the library (the tail of one sub-tile's persistent kernels overlaps the bulk of another's; large tiles use one);
 * they start after the work already queued on `hip_stream` and that stream is made to wait for them. */
int rtmi_render(rtmi_scene_t* scene, const rtmi_viewport_t* vp, uint64_t seed,
                uint32_t row0, uint32_t nrows, float* out_host, rtmi_stats_t* stats);
int rtmi_render_device(rtmi_scene_t* scene, const rtmi_viewport_t* vp, uint64_t seed,
                       uint32_t row0, uint32_t nrows, void* out_device, void* hip_stream,
                       rtmi_stats_t* stats);
/* Same for a striped row set; output row i is the i-th row of the tile. */
int rtmi_render_tile_device(rtmi_scene_t* scene, const rtmi_viewport_t* vp, uint64_t seed,
                            const rtmi_tile_t* tile, void* out_device, void* hip_stream,
                            rtmi_stats_t* stats);

/* One whole frame over several devices of this process -- the fan-out the reference does over CPU threads
 * (DefaultRayCaster::walk_rays_internal, raytrace.rs:1175-1196: `threads` workers pulling rows from a queue) done
 * over GPUs, inside the library.  scenes[i] is the SAME scene uploaded to some device (rtmi_scene_create with
 * device = i; two handles may also share a device).  Scene i renders the interleaved stripes
 * {i*S, rows_i, S, n*S} (S = stripe_rows, 0 = default 16) on its own host thread and stream; every band then
 * crosses to scenes[0]'s device ONCE (hipMemcpyPeerAsync: one xGMI link per peer, all links at the same time),
 * where a kernel de-interleaves the stripes into the frame.  No other exchange: pixels are independent and the RNG
 * is keyed by (pixel, sample), so the frame is bit-identical to rtmi_render() of the whole image on one device.
 * flags: RTMI_FRAME_RGB8 quantises every band on its device first ((c*255.) as u8, raytrace.rs:1468-1473), so
 * 3 bytes per pixel cross the links instead of 16 and the output is height*width*3 bytes; otherwise the output is
 * height*width*4 floats (`[Color]`).  out_host and/or out_device (memory of scenes[0]'s device) receive the frame.
 * stats (optional) has nscenes entries, one per scene; "Rays" of the frame is their sum. */
enum {
    RTMI_FRAME_RGB8 = 1u << 0,
    RTMI_FRAME_RCCL = 1u << 1  /* the bands cross to scenes[0]'s device with ONE ncclGather (RCCL over xGMI; rccl.h) on a
                                * communicator the library makes for the scenes' devices (ncclCommInitAll, kept on scenes[0])
                                * instead of one hipMemcpyPeerAsync per band.  librccl.so.1 is loaded on first use (a process
                                * that holds PyTorch's RCCL gets that copy).  Every scene handle must sit on a device of its
                                * own.  Same frame either way; without the flag the library stays free of RCCL.            */
};
int rtmi_render_frame_multi(rtmi_scene_t* const* scenes, uint32_t nscenes, const rtmi_viewport_t* vp, uint64_t seed,
                            uint32_t stripe_rows, uint32_t flags, void* out_host, void* out_device, rtmi_stats_t* stats);

/* Closest hit for n explicit rays: orig (x,y,z,lane3) and unit dir
 * (x,y,z,lane3) as `make_ray` stores them (raytrace.rs:201-210).  Outputs per
 * ray: triangle index (0 = miss), hit time, face (0 front, 1 back, 2 edge
 * front, 3 edge back).  Same function as the reference's native entry point
 * exec_cuda_raytrace (cuda_raytrace_lib/src/cuda_raytrace.rs:20-58,
 * cuda_rt.h:8-15) but with the CPU path's semantics
 * (BoundingBox::get_object_intersection_for_ray, raytrace.rs:909-1010). */
int rtmi_trace(rtmi_scene_t* scene, uint64_t n, const float* orig4, const float* dir4,
               uint32_t* tri, float* t, uint32_t* face, rtmi_stats_t* stats);

/* (c * 255.) as u8 per channel, RGB (raytrace.rs:1468-1473). */
int rtmi_quantize(rtmi_scene_t* scene, const float* rgba_host, uint64_t npixels, uint8_t* rgb_host);
/* Same on device memory, enqueued on `hip_stream`: lets a rank hand 3 bytes per pixel to the gather instead of 16. */
int rtmi_quantize_device(rtmi_scene_t* scene, const void* rgba_device, uint64_t npixels, void* rgb_device, void* hip_stream);

/* `make_triangle` (raytrace.rs:340-383) for n triangles on the GPU: corners (9 floats each) -> the geometric fields
 * of rtmi_triangle_t (incenter, norm, bounding_r2, sides, side_lens); edge_thickness and the surface fields are copied
 * from `proto`.  Bit-identical to the host computation.  Fails with RTMI_ERR_INVALID, naming the first offender, where
 * the reference panics on a degenerate triangle (unwrap at raytrace.rs:357). */
int rtmi_make_triangles(int device, const float* corners9_host, uint64_t n, const rtmi_triangle_t* proto,
                        rtmi_triangle_t* out_host);

/* Octree build on the GPU (raytrace.rs:753-845).  The builder's whole cost is box_contains_polygon (raytrace.rs:753-779)
 * on every (candidate child box, triangle of the parent's list) pair; rtmi_builder_filter evaluates it for all pairs of
 * one tree level, bit-identical to the host computation, so the resulting tree equals the reference builder's.  The
 * caller keeps the list bookkeeping (which is linear): see build_bounding_box_gpu in csrc/host/raytrace.cpp.
 * tris15: ntris x 15 floats (incenter, norm, corner 0, corner 1, corner 2), kept on the device by the handle.
 * boxes[b]: geometry + the range [cand_first, cand_first + cand_count) of `cand` (triangle indices) to test against it;
 * keep[keep_first + j] receives 1 when box b contains candidate j, else 0.  Ranges of different boxes may overlap in
 * `cand` (the 8 children of a box share their parent's list) but not in `keep`. */
typedef struct rtmi_build_box {
    float orig[3];
    float len2;
    uint32_t cand_first, cand_count;
    uint64_t keep_first;
} rtmi_build_box_t;
typedef struct rtmi_builder rtmi_builder_t;
int rtmi_builder_create(int device, const float* tris15, uint64_t ntris, rtmi_builder_t** out);
int rtmi_builder_filter(rtmi_builder_t* builder, const rtmi_build_box_t* boxes, uint64_t nboxes, const uint32_t* cand,
                        uint64_t ncand, uint8_t* keep, uint64_t nkeep);
int rtmi_builder_destroy(rtmi_builder_t* builder);

/* Message of the last error on the calling thread ("" if none). */
const char* rtmi_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RTMI_H */
