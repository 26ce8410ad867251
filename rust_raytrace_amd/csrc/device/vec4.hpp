// vec4.hpp — the reference's `Vec3` (a 4-lane f32 SIMD value whose lane 3 is
// normally 0, raytrace_lib/src/raytrace.rs:22-122) for device code.
//
// Used by the ray-generation and shading kernels, which are elementwise and
// can afford to carry lane 3 through every operation exactly as the reference
// does.  Compile with -ffp-contract=off: Rust never contracts a*b+c.
#pragma once
#include <hip/hip_runtime.h>

namespace rtmi {

struct V4 { float x, y, z, w; };

__host__ __device__ inline V4 mk(float a, float b, float c) { return V4{a, b, c, 0.f}; }
__host__ __device__ inline V4 vadd(V4 a, V4 b) { return V4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
__host__ __device__ inline V4 vsub(V4 a, V4 b) { return V4{a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
__host__ __device__ inline V4 vmul(V4 a, float s) { return V4{a.x * s, a.y * s, a.z * s, a.w * s}; }
// ordered sum seeded with +0 (raytrace.rs:65-77)
__host__ __device__ inline float vdot(V4 a, V4 b) {
    return (((0.f + a.x * b.x) + a.y * b.y) + a.z * b.z) + a.w * b.w;
}
__host__ __device__ inline float vlen2(V4 a) { return vdot(a, a); }
__host__ __device__ inline float vlen(V4 a) { return sqrtf(vlen2(a)); }
// raytrace.rs:93-96: multiply by the reciprocal, not divide
__host__ __device__ inline V4 vunit(V4 a) { return vmul(a, 1.f / vlen(a)); }

// rand 0.8 `Standard` for f32: top 24 bits * 2^-24
__host__ __device__ inline float u32_to_unit_f32(uint32_t u) { return (float)(u >> 8) * (1.0f / 16777216.0f); }

// Philox4x32-10; counter {block, sample, pixel, 'RTMI'}, key = seed.
__host__ __device__ inline uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}
__host__ __device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint32_t hi0 = mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__host__ __device__ inline void rng_block(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t blk, uint32_t out[4]) {
    philox4x32_10(blk, sample, pixel, 0x52544d49u, (uint32_t)seed, (uint32_t)(seed >> 32), out);
}

// Streaming stores: sample colours and queue entries are written once and read once by a LATER launch; written with the
// non-temporal hint they do not displace the scene's boxes and triangles from L2 (4 MB per XCD) while a trace kernel runs.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ inline void store_stream(float4* p, float4 v) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f x = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(x, reinterpret_cast<v4f*>(p));
}
__device__ inline void store_stream(uint32_t* p, uint32_t v) { __builtin_nontemporal_store(v, p); }
#else  // host pass of the same translation unit: never executed
__device__ inline void store_stream(float4* p, float4 v) { *p = v; }
__device__ inline void store_stream(uint32_t* p, uint32_t v) { *p = v; }
#endif

}  // namespace rtmi
