// round 6 probe (VERDICT r5 "next" #4a): a guest kernel that does nothing but f32 MFMA (v_mfma_f32_32x32x2_f32, four independent accumulators)
// or nothing but f32 VALU (v_fma_f32), one wave per SIMD on every CU, launched on a stream of its own beside the pipelined receiver.
// Built as a shared library (hipcc -shared) and driven from scratch/r6/mfma_probe.py through ctypes.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void guest_mfma(float *out, int iters)
{
    v16f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float a = (float)threadIdx.x * 1e-3f, b = 1.0f + (float)blockIdx.x * 1e-6f;
    for (int i = 0; i < iters; i++) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
    }
    float s = 0.f;
    for (int k = 0; k < 16; k++) s += c0[k] + c1[k] + c2[k] + c3[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void guest_valu(float *out, int iters)
{
    float acc[16];
    for (int k = 0; k < 16; k++) acc[k] = (float)(threadIdx.x + k);
    const float t = 1.0001f;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int k = 0; k < 16; k++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(t), "v"(acc[(k + 1) & 15]));
    }
    float s = 0.f;
    for (int k = 0; k < 16; k++) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static float *g_out = nullptr;
extern "C" int guest_launch(int kind, int iters, void *stream)
{
    if (!g_out && hipMalloc((void **)&g_out, 256 * 256 * sizeof(float)) != hipSuccess) return 1;
    // 256 workgroups of 256 threads = four waves per CU = one per SIMD
    if (kind == 0) hipLaunchKernelGGL(guest_mfma, dim3(256), dim3(256), 0, (hipStream_t)stream, g_out, iters);
    else hipLaunchKernelGGL(guest_valu, dim3(256), dim3(256), 0, (hipStream_t)stream, g_out, iters);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
// flops of one launch: MFMA 32x32x2 = 2 * 32 * 32 * 2 per instruction and wave; VALU: 2 per lane and instruction
extern "C" double guest_flops(int kind, int iters)
{
    const double waves = 256.0 * 4.0;
    return kind == 0 ? waves * iters * 4.0 * (2.0 * 32 * 32 * 2) : waves * iters * 8.0 * 16.0 * 64.0 * 2.0;
}
