// shade.hpp — primary-ray generation and shading of one traced ray: Viewport::pixel_ray (raytrace_lib/src/raytrace.rs:
// 1374-1394), color_ray + the tail of project_ray (raytrace.rs:1199-1295), lambertian_ray / reflect_ray / random_vec /
// mix_color (raytrace.rs:278-301, :188-192) as device functions, shared by the per-pass kernels of rtmi_device.hip
// (k_gen, k_shade) and by the fused path kernels of trace_oct.hpp (k_path_primary, k_path_bounce), so that both
// pipelines execute the same arithmetic.  4-lane V4 values in the reference's operation order (vec4.hpp).
// Included by rtmi_device.hip.
#pragma once

namespace rtmi {

// Division of a u32 by a divisor that is constant for a launch (samples per pixel, image width, stripe height), without
// the ~40-instruction software division: Granlund & Montgomery, "Division by invariant integers using multiplication"
// (1994), figure 4.1 -- exact for every n < 2^32 and every d >= 1:
//   l = ceil(log2 d), m' = floor(2^32 (2^l - d) / d) + 1, sh1 = min(l, 1), sh2 = max(l - 1, 0)
//   q = (t + ((n - t) >> sh1)) >> sh2  with  t = mulhi(m', n)
struct FastDiv { uint32_t mul, sh1, sh2, d; };
inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f{1u, 0u, 0u, d ? d : 1u};
    uint32_t l = 0;
    while (l < 32 && (1ull << l) < f.d) l++;
    f.mul = (uint32_t)((((1ull << l) - f.d) << 32) / f.d) + 1u;
    f.sh1 = l < 1u ? l : 1u;
    f.sh2 = l > 0u ? l - 1u : 0u;
    return f;
}
__host__ __device__ inline uint32_t fdiv(uint32_t n, const FastDiv& f) {
    const uint32_t t = mulhi32(f.mul, n);
    return (t + ((n - t) >> f.sh1)) >> f.sh2;
}

struct DView {
    V4 orig, cam, vu, vv;
    uint32_t width, height, maxdepth, spp;
    uint32_t row0, stripe_rows, stripe_step, pad;  // rtmi_tile_t: which image rows the tile's rows are
    uint32_t sub_mul, sub_off;                     // sub-tile of a stream: rows sub_off, sub_off + sub_mul, ... of the tile
    FastDiv dspp, dwidth, dstripe;                 // n / spp, n / width, n / stripe_rows
};
inline void view_set_divisors(DView& v) {
    v.dspp = make_fastdiv(v.spp);
    v.dwidth = make_fastdiv(v.width);
    v.dstripe = make_fastdiv(v.stripe_rows);
}

// local pixel index of the sub-tile (row-major over ITS rows) -> image (row, col).  Row lr of the sub-tile is row
// lr * sub_mul + sub_off of the tile; row L of the tile is image row row0 + (L / stripe_rows) * stripe_step + L % stripe_rows.
__device__ inline void tile_pixel(const DView& v, uint32_t lp, uint32_t& row, uint32_t& col) {
    const uint32_t lr = fdiv(lp, v.dwidth);
    col = lp - lr * v.width;
    const uint32_t L = lr * v.sub_mul + v.sub_off;
    const uint32_t k = fdiv(L, v.dstripe);
    row = v.row0 + k * v.stripe_step + (L - k * v.stripe_rows);
}
// path index of a batch that starts at local pixel pix0 -> image pixel index (row * width + col) and sample number
__device__ inline void path_pixel(const DView& v, uint32_t pix0, uint32_t path, uint32_t& row, uint32_t& col, uint32_t& sample) {
    const uint32_t q = fdiv(path, v.dspp);
    sample = path - q * v.spp;
    tile_pixel(v, pix0 + q, row, col);
}

struct RayV { V4 orig, dir; };
// make_ray (raytrace.rs:201-210); inv_dir is recomputed by the trace kernel
__device__ inline RayV make_ray(V4 orig, V4 dir) { return RayV{orig, vunit(dir)}; }

// Viewport::pixel_ray (raytrace.rs:1374-1394), px = (row, col)
__device__ inline RayV pixel_ray(const DView& v, uint32_t row, uint32_t col, uint64_t seed, uint32_t pixel, uint32_t sample) {
    float px_x = (float)row, px_y = (float)col;
    V4 vu_delta = vmul(v.vu, 1.f / (float)v.width);
    V4 vv_delta = vmul(v.vv, 1.f / (float)v.height);
    float u_off = 0.5f, v_off = 0.5f;
    if (v.spp != 1) {
        uint32_t w[4];
        rng_block(seed, pixel, sample, 0, w);
        u_off = u32_to_unit_f32(w[0]);
        v_off = u32_to_unit_f32(w[1]);
    }
    V4 vu_frac = vmul(vu_delta, px_y + u_off);
    V4 vv_frac = vmul(vv_delta, px_x + v_off);
    V4 px_u = vadd(vadd(v.orig, vu_frac), vv_frac);
    return make_ray(px_u, vunit(vsub(px_u, v.cam)));
}

// random_vec (raytrace.rs:188-192): k-th call of the path uses RNG block k
__device__ inline V4 random_vec(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t k) {
    uint32_t w[4];
    rng_block(seed, pixel, sample, k, w);
    return vunit(mk(u32_to_unit_f32(w[0]) - 0.5f, u32_to_unit_f32(w[1]) - 0.5f, u32_to_unit_f32(w[2]) - 0.5f));
}
// mix_color (raytrace.rs:299-301)
__device__ inline V4 mix_color(V4 c1, V4 c2, float a) { return vadd(vmul(c1, 1.f - a), vmul(c2, a)); }

// color_ray + the tail of project_ray for ONE traced ray of a path.  `pass` = bounces the path has behind it (the ray's
// remaining depth is maxdepth - pass >= 1); tf = hit triangle | face << 30 (0 = miss), t = hit time; (ro, rd) the ray.
// Returns true when the path goes on: its surface has been pushed on the path's stack (mstack[pass][path]) and `nr` is
// the bounce ray (lambertian_ray / reflect_ray with the path's RNG block pass + 1).  Returns false when the path ends
// here: sky, Solid, edge face (Solid black, raytrace.rs:452-457) or depth exhausted (black, raytrace.rs:1261-1263); the
// nested mix_color calls (raytrace.rs:1233-1251) are then evaluated inside-out over the stack and the sample colour is
// written to scol[path].
__device__ inline bool shade_hit(const DScene& sc, uint32_t maxdepth, uint64_t seed, uint32_t npaths, uint32_t path, uint32_t pixel,
                                 uint32_t sample, uint32_t pass, uint32_t tf, float t, V4 ro, V4 rd, uint16_t* __restrict__ mstack,
                                 float4* __restrict__ scol, RayV& nr) {
    const uint32_t tri = tf & 0x3FFFFFFFu, face = tf >> 30;
    V4 c;
    uint32_t npushed = pass;
    if (tri == 0) {
        c = mk(128.f / 255.f, 180.f / 255.f, 255.f / 255.f);  // raytrace.rs:1264
    } else if (face & 2u) {
        c = mk(0.f / 255.f, 0.f / 255.f, 0.f / 255.f);  // edge faces are Solid black, raytrace.rs:452-457
    } else {
        // a triangle's record, or (hit index >= ntris) an analytic sphere's: the sphere's normal needs the hit point and
        // is filled in below
        const bool is_sphere = tri >= sc.ntris;
        const float4 p1 = is_sphere ? sc.spheres[2 * (tri - sc.ntris) + 1] : sc.tplane[2 * tri + 1];
        const uint32_t mat = __float_as_uint(is_sphere ? p1.x : p1.w);
        const float4 m0 = sc.mats[2 * mat], m1 = sc.mats[2 * mat + 1];
        const uint32_t kind = __float_as_uint(m1.y);
        if (kind == RTMI_SOLID) {
            c = mk(m0.x, m0.y, m0.z);
        } else {
            mstack[(size_t)pass * npaths + path] = (uint16_t)mat;
            npushed = pass + 1;
            c = mk(0.f / 255.f, 0.f / 255.f, 0.f / 255.f);  // project_ray at depth 0, raytrace.rs:1261-1263
            if (maxdepth - pass - 1u != 0u) {
                const V4 point = vadd(vmul(rd, t), ro);                      // Ray::at, raytrace.rs:227-229
                V4 norm = mk(p1.x, p1.y, p1.z);
                if (is_sphere) {  // (point - center).unit()
                    const float4 sc0 = sc.spheres[2 * (tri - sc.ntris)];
                    norm = vunit(vsub(point, mk(sc0.x, sc0.y, sc0.z)));
                }
                if (face & 1u) norm = vmul(norm, -1.f);                       // raytrace.rs:441-449
                const V4 rv = random_vec(seed, pixel, sample, pass + 1u);
                if (kind == RTMI_MATTE) {
                    nr = make_ray(vadd(point, vmul(rv, 0.001f)), vadd(norm, rv));  // lambertian_ray, :292-297
                } else {
                    const float ddot = fabsf(vdot(rd, norm));                 // reflect_ray, :278-290
                    const V4 dir_p = vmul(norm, ddot);
                    const V4 dir_o = vadd(rd, dir_p);
                    const V4 reflect = vadd(dir_p, dir_o);
                    const V4 rvf = vmul(rv, m1.x);
                    const V4 reflect_dir = vunit(vadd(reflect, rvf));
                    nr = make_ray(vadd(point, vmul(reflect_dir, 0.001f)), vunit(vadd(reflect, rvf)));
                }
                return true;
            }
        }
    }
    // inside-out evaluation of the nested mix_color calls (raytrace.rs:1233-1251)
    for (int j = (int)npushed - 1; j >= 0; j--) {
        const uint32_t mj = mstack[(size_t)j * npaths + path];
        const float4 mm = sc.mats[2 * mj];
        c = mix_color(mk(mm.x, mm.y, mm.z), c, mm.w);
    }
    store_stream(&scol[path], make_float4(c.x, c.y, c.z, c.w));
    return false;
}

}  // namespace rtmi
