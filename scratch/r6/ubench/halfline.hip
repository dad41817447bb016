// Do 64-byte half-line non-temporal stores (two per 128-byte line, a few microseconds apart, as channelizer_kernel writes its granules) cost
// HBM write bandwidth against whole-line stores?  hipcc --offload-arch=gfx950 -O3 halfline.hip -o halfline && ./halfline
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
// lines: total 128-byte lines; each workgroup owns a contiguous run of lines and walks it in "tiles" of 512 lines (one per channel)
template <int MODE>   // 0: whole lines (8 lanes x 16 B per line); 1: first halves of a tile, then its second halves; 2: like 1 with a delay loop between
__global__ __launch_bounds__(512) void wr(v4f *out, size_t tiles_per_wg, int spin)
{
    const size_t base = (size_t)blockIdx.x * tiles_per_wg * 512 * 8;      // 16-byte units
    const int tid = threadIdx.x;
    v4f v = { (float)tid, 1.f, 2.f, 3.f };
    for (size_t t = 0; t < tiles_per_wg; t++) {
        v4f *p = out + base + t * 512 * 8;
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 8; k++) __builtin_nontemporal_store(v, p + (size_t)(tid + k * 512));       // lane l: line (tid + 512 k) / 8, unit % 8
        } else {
#pragma unroll
            for (int h = 0; h < 2; h++) {
#pragma unroll
                for (int k = 0; k < 4; k++) { const int o = tid + k * 512; const int ch = o >> 2, rp = o & 3; __builtin_nontemporal_store(v, p + (size_t)(ch * 8 + h * 4 + rp)); }
                if (MODE == 2) { for (int s = 0; s < spin; s++) __builtin_amdgcn_s_sleep(32); v.y += 1.f; }
            }
        }
    }
}
int main()
{
    const size_t tiles_per_wg = 200, wgs = 512, units = wgs * tiles_per_wg * 512 * 8;     // 0.84 GB
    v4f *d; hipMalloc(&d, units * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; mode++) for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        if (mode == 0) wr<0><<<wgs, 512>>>(d, tiles_per_wg, 0);
        else if (mode == 1) wr<1><<<wgs, 512>>>(d, tiles_per_wg, 0);
        else wr<2><<<wgs, 512>>>(d, tiles_per_wg, 8);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) printf("mode %d (%s): %.3f ms, %.2f TB/s written\n", mode, mode == 0 ? "whole 128-byte lines" : mode == 1 ? "64-byte halves back to back" : "64-byte halves, a pause between", ms, units * 16 / ms / 1e9);
    }
    return 0;
}
