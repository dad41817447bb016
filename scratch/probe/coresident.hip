// coresident.hip -- scheduling probe (measurement helper, not product code): a VALU-bound kernel with a chosen register
// footprint that records, per wave, where and when it ran.  Launched beside the K = 1024 channelizer it shows whether
// waves of another kernel share a CU (and its SIMDs' issue slots) with a resident channelizer workgroup.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int NACC>
__global__ __launch_bounds__(64) void burn_kernel(int iters, float a, float b, unsigned long long *out)
{
    const unsigned long long w0 = wall_clock64();
    const unsigned long long c0 = __builtin_readcyclecounter();
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    float acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = (float)(threadIdx.x + i);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_fmaf(acc[i], a, b);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i];
    const unsigned long long c1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    if (threadIdx.x == 0) {
        unsigned long long *o = out + (size_t)blockIdx.x * 4;
        o[0] = w0; o[1] = w1; o[2] = c1 - c0; o[3] = ((unsigned long long)xcc << 32) | hwid;
    }
    if (s == 12345.678f) out[0] = 0;        // keep the sums alive
}

extern "C" int probe_burn(void *stream, int nwaves, int nacc, int iters, void *d_out)
{
    hipStream_t st = (hipStream_t)stream;
    if (nacc == 72) hipLaunchKernelGGL(burn_kernel<72>, dim3(nwaves), dim3(64), 0, st, iters, 1.0000001f, 1e-9f, (unsigned long long *)d_out);
    else if (nacc == 150) hipLaunchKernelGGL(burn_kernel<150>, dim3(nwaves), dim3(64), 0, st, iters, 1.0000001f, 1e-9f, (unsigned long long *)d_out);
    else hipLaunchKernelGGL(burn_kernel<32>, dim3(nwaves), dim3(64), 0, st, iters, 1.0000001f, 1e-9f, (unsigned long long *)d_out);
    return (int)hipGetLastError();
}
