// Development aid (not part of the product): how fast can a CU pull random 128-B lines out of L2 when
//   A  every lane fetches its own line with 8 x global_load_dwordx4 (the shape of k_trace_oct's LEAF step), or
//   B  8 lanes fetch one line together (8 instructions serve 64 lines), data handed to the owner lane through LDS, or
//   C  as B with LDS-DMA (global_load_lds_dwordx4)?
// build: hipcc --offload-arch=gfx950 -O3 -o gather_lines gather_lines.hip ; run: ./gather_lines [table_KiB] [active_lanes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline uint32_t hash(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ void __launch_bounds__(64, 4) k(const float4* __restrict__ tab, uint32_t nlines, int iters, int active, float* out) {
    extern __shared__ float4 stage[];  // 64 lines x 128 B per wave
    const int lane = threadIdx.x;
    const uint32_t wid = blockIdx.x;
    float acc = 0.f;
    const bool on = lane < active;
    for (int it = 0; it < iters; it++) {
        const uint32_t line = hash(wid * 64u + lane + 0x9E3779B9u * (uint32_t)it) % nlines;
        float4 v[8];
        if (MODE == 0) {
            if (on) {
#pragma unroll
                for (int k = 0; k < 8; k++) v[k] = tab[8 * (size_t)line + k];
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++) v[k] = make_float4(0, 0, 0, 0);
            }
        } else {
            // instruction j serves the owners 8g + j (g = lane >> 3): lane 8g + p fetches piece p ^ j of that owner's line,
            // it lands in region j at lane * 16; the owner o reads piece k at region o & 7, slot (o >> 3) * 8 + (k ^ (o & 7))
            const int g = lane >> 3, p = lane & 7;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const int owner = 8 * g + j;
                const uint32_t ol = (uint32_t)__builtin_amdgcn_ds_bpermute(owner * 4, (int)line);
                const float4* src = tab + 8 * (size_t)ol + (p ^ j);
                if (owner < active) {
                    if (MODE == 1) stage[j * 64 + lane] = *src;
                    else __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(stage + j * 64), 16, 0, 0);
                }
            }
            if (MODE == 2) __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0): the DMA writes have landed
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (on) {
#pragma unroll
                for (int k = 0; k < 8; k++) v[k] = stage[(lane & 7) * 64 + (lane >> 3) * 8 + (k ^ (lane & 7))];
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++) v[k] = make_float4(0, 0, 0, 0);
            }
        }
        // ~140 VALU of dependent-free math per step like the plane tests
#pragma unroll
        for (int k = 0; k < 8; k++) acc += v[k].x * v[k].y + v[k].z * v[k].w;
    }
    if (acc == 123.456f) out[0] = acc;
}

int main(int argc, char** argv) {
    const size_t kib = argc > 1 ? atoi(argv[1]) : 2048;
    const int active = argc > 2 ? atoi(argv[2]) : 36;
    const uint32_t nlines = (uint32_t)(kib * 1024 / 128);
    std::vector<float4> h(8 * (size_t)nlines);
    for (size_t i = 0; i < h.size(); i++) h[i] = make_float4((float)i, 1.f, 2.f, 3.f);
    float4* tab; float* out;
    CHK(hipMalloc(&tab, h.size() * 16)); CHK(hipMalloc(&out, 4));
    CHK(hipMemcpy(tab, h.data(), h.size() * 16, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int wpc : {8, 12, 16}) {
        const int waves = 256 * wpc;
        for (int mode = 0; mode < 3; mode++) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                CHK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(waves), dim3(64), 8192, 0, tab, nlines, iters, active, out);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(waves), dim3(64), 8192, 0, tab, nlines, iters, active, out);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(waves), dim3(64), 8192, 0, tab, nlines, iters, active, out);
                CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
                float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
            }
            const double lines = (double)waves * iters * active;
            printf("table %zu KiB, %d active lanes, %2d waves/CU, mode %c: %.3f ms, %.1f Glines/s, %.2f lines/clk/CU (2.4 GHz), %.0f GB/s\n", kib, active, wpc,
                   "ABC"[mode], best, lines / best / 1e6, lines / (best * 1e-3) / 256 / 2.4e9, lines * 128 / best / 1e6);
        }
    }
    return 0;
}
