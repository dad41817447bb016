// round 6 microbenchmark: issue cost of v_pk_fma_f32 against v_fma_f32 on gfx950 (wave64), 1 / 2 / 4 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float t)
{
    v2f a[16]; 
#pragma unroll
    for (int i = 0; i < 16; i++) a[i] = (v2f){ (float)threadIdx.x + i, (float)i };
    v2f tv = { t, t * 0.5f };
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(tv), "v"(a[(i + 1) & 15]));
                else {
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(tv.x), "v"(a[(i + 1) & 15].x));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].y) : "v"(tv.y), "v"(a[(i + 1) & 15].y));
                }
            }
        }
    }
    float s = 0; for (int i = 0; i < 16; i++) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    float *d; hipMalloc(&d, 256 * 4 * 256 * 8 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        for (int mode = 0; mode < 2; mode++) {
            dim3 grid(256 * wps), block(256);      // 256 CUs x wps workgroups of 4 waves = wps waves per SIMD
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, grid, block, 0, 0, d, iters, 1.0001f);
                else hipLaunchKernelGGL(k<1>, grid, block, 0, 0, d, iters, 1.0001f);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep == 2) {
                    const double instr = (double)iters * 8 * 16 * (mode == 0 ? 1 : 2) * wps;     // per SIMD
                    printf("waves/SIMD %d %s: %.3f ms, %.2f cycles per instruction per SIMD at 2.4 GHz, %.1f TFLOP/s\n", wps, mode == 0 ? "v_pk_fma_f32" : "v_fma_f32   ",
                           ms, ms * 1e-3 * 2.4e9 / instr, (double)iters * 8 * 16 * 2 * 2 * 64 * wps * 1024 / (ms * 1e-3) / 1e12);
                }
            }
        }
    }
    return 0;
}
