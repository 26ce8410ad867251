// Development aid: issue rate of plain (non-packed, non-FMA) f32 VALU instructions per SIMD on gfx950.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(64) kpk(float* out, int iters, float a, float b) {
    v2f x0 = {(float)threadIdx.x, 1.f}, x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f, x4 = x0 + 4.f, x5 = x0 + 5.f, x6 = x0 + 6.f, x7 = x0 + 7.f;
    const v2f a2 = {a, a}, b2 = {b, b};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) { x0 = x0 * a2; x1 = x1 + b2; x2 = x2 * a2; x3 = x3 + b2; x4 = x4 * a2; x5 = x5 + b2; x6 = x6 * a2; x7 = x7 + b2; }
    }
    const v2f s = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
    if (s.x + s.y == 123.456f) out[0] = s.x;
}
template <int KIND>
__global__ void __launch_bounds__(64) k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (KIND == 0) { x0 = x0 * a; x1 = x1 + b; x2 = x2 * a; x3 = x3 + b; x4 = x4 * a; x5 = x5 + b; x6 = x6 * a; x7 = x7 + b; }
            if (KIND == 1) { x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b); x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b); }
            if (KIND == 2) { x0 = fmaxf(x0, a); x1 = x1 > b ? x2 : x1; x2 = fminf(x2, a); x3 = x3 > b ? x4 : x3; x4 = fmaxf(x4, a); x5 = x5 > b ? x6 : x5; x6 = fminf(x6, a); x7 = x7 > b ? x0 : x7; }
        }
    }
    const float s = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
    if (s == 123.456f) out[0] = s;
}
int main() {
    float* out; CHK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int kind = 0; kind < 4; kind++)
        for (int wps : {1, 2, 4, 8}) {
            const int waves = 256 * 4 * wps;
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                CHK(hipEventRecord(e0));
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(waves), dim3(64), 0, 0, out, iters, 1.0001f, 0.5f);
                if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(waves), dim3(64), 0, 0, out, iters, 1.0001f, 0.5f);
                if (kind == 3) hipLaunchKernelGGL(kpk, dim3(waves), dim3(64), 0, 0, out, iters, 1.0001f, 0.5f);
                if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(waves), dim3(64), 0, 0, out, iters, 1.0001f, 0.5f);
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
            }
            const double insts = (double)waves * iters * 64.0 * (kind == 2 ? 1.5 : 1.0);  // kind 2: cmp + cndmask pairs
            printf("kind %d (%s), %d waves/SIMD: %.3f ms, %.2f wave-instr/clk/SIMD at 2.4 GHz, %.1f Tlane-op/s\n", kind,
                   kind == 0 ? "mul/add" : kind == 1 ? "fma" : kind == 2 ? "max/min/cmp+cndmask" : "pk mul/add (instr = packed instr)", wps, best, insts / 1024.0 / (best * 1e-3 * 2.4e9), insts * 64 / best / 1e9);
        }
    return 0;
}
