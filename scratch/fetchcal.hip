// fetchcal.hip -- what does rocprofv3's FETCH_SIZE report for the payload workers' access shape?
// (VERDICT r3 "next" #1: calibrate the counter on a known byte count before trusting roofline.traffic.)
// Every kernel reads each byte it touches exactly once, from a 4 GB buffer (16 x the 256 MB Infinity Cache), and
// prints the bytes it REQUESTED and the bytes of the 128-B lines / 64-B half lines it touched; the rocprofv3 --pmc FETCH_SIZE
// pass of this binary (scratch/fetchcal.sh) puts the counter beside them.
//   stream16    16 B per lane, coalesced (the guide's calibrated case: FETCH_SIZE = 1/2 of the bytes)
//   stream8      8 B per lane, coalesced (512 B contiguous per wave load)
//   gran128      8 B per lane, 16 lanes per 128-B granule, granules 32 KB apart  (MCRX_TILE 16: a channel per line)
//   gran64half   8 B per lane,  8 lanes per  64-B granule, granules 32 KB apart, only the FIRST half of every 128-B line is ever read
//                (MCRX_TILE 8 with the neighbour channel's frames somewhere else in time: the worst case)
//   gran64pair   as gran64half, but another wave reads the second halves much later in the same launch (neighbour channel, other frame)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void stream16(const float4 *p, size_t n, float *sink)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = p[i]; acc += v.x + v.w; }
    if (acc == 1.2345e-30f) *sink = acc;
}
__global__ void stream8(const float2 *p, size_t n, float *sink)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float2 v = p[i]; acc += v.x + v.y; }
    if (acc == 1.2345e-30f) *sink = acc;
}
// a wave = one "window": 64 samples of one channel = 64 / G granules of G samples, granule g of channel c and tile t at
// ((t * C + c) * G) float2.  Wave w takes channel w % C (first `used` channels of every group of `grp` only), windows along t.
template <int G>
__global__ void gran(const float2 *p, unsigned C, unsigned tiles, unsigned grp, unsigned used, unsigned phase, float *sink)
{
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const unsigned cu = wave % (C / grp * used), c = cu / used * grp + (cu % used + phase) % grp;
    const unsigned wins = tiles / (64 / G);
    float acc = 0.f;
    for (unsigned w = wave / (C / grp * used); w < wins; w += (gridDim.x * blockDim.x >> 6) / (C / grp * used)) {
        const unsigned t = w * (64 / G) + lane / G;
        const float2 v = p[((size_t)t * C + c) * G + lane % G];
        acc += v.x + v.y;
    }
    if (acc == 1.2345e-30f) *sink = acc;
}

int main()
{
    const size_t bytes = (size_t)4 << 30;
    void *buf; float *sink;
    CHK(hipMalloc(&buf, bytes)); CHK(hipMalloc((void **)&sink, 4));
    CHK(hipMemset(buf, 0, bytes));
    CHK(hipDeviceSynchronize());
    const unsigned C = 512;                                     // channels: granules of a channel are C * G * 8 bytes apart
    hipLaunchKernelGGL(stream16, dim3(4096), dim3(256), 0, 0, (const float4 *)buf, bytes / 16, sink);
    printf("stream16    requested %zu bytes, lines touched %zu bytes\n", bytes, bytes);
    hipLaunchKernelGGL(stream8, dim3(4096), dim3(256), 0, 0, (const float2 *)buf, bytes / 8, sink);
    printf("stream8     requested %zu bytes, lines touched %zu bytes\n", bytes, bytes);
    {   // TILE 16: every channel, every tile: the whole buffer, 128-B granules
        const unsigned tiles = (unsigned)(bytes / ((size_t)C * 128));
        hipLaunchKernelGGL(gran<16>, dim3(8192), dim3(256), 0, 0, (const float2 *)buf, C, tiles, 1u, 1u, 0u, sink);
        printf("gran128     requested %zu bytes, lines touched %zu bytes\n", (size_t)tiles * C * 128, (size_t)tiles * C * 128);
    }
    {   // TILE 8: only even channels = the first half of every 128-B line
        const unsigned tiles = (unsigned)(bytes / ((size_t)C * 64));
        hipLaunchKernelGGL(gran<8>, dim3(8192), dim3(256), 0, 0, (const float2 *)buf, C, tiles, 2u, 1u, 0u, sink);
        printf("gran64half  requested %zu bytes, 64-B halves touched %zu bytes, 128-B lines touched %zu bytes\n", (size_t)tiles * (C / 2) * 64, (size_t)tiles * (C / 2) * 64, (size_t)tiles * (C / 2) * 128);
        // ... and both halves, by waves that are far apart in the launch (all even channels first, then all odd ones)
        hipLaunchKernelGGL(gran<8>, dim3(8192), dim3(256), 0, 0, (const float2 *)buf, C, tiles, 2u, 1u, 1u, sink);
        printf("gran64odd   requested %zu bytes (the other halves, own launch)\n", (size_t)tiles * (C / 2) * 64);
        hipLaunchKernelGGL(gran<8>, dim3(8192), dim3(256), 0, 0, (const float2 *)buf, C, tiles, 1u, 1u, 0u, sink);
        printf("gran64all   requested %zu bytes, every channel in one launch (neighbouring channels = neighbouring waves)\n", (size_t)tiles * C * 64);
    }
    CHK(hipDeviceSynchronize());
    return 0;
}
