#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *o)
{
    const int l = threadIdx.x;
    unsigned a = 100 + l, b = 200 + l;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    int v = l;
    int t = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xf, 0x5, false);
    int x4 = __builtin_amdgcn_update_dpp(t, v, 0x114, 0xf, 0xa, false);
    int r8 = __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, true);
    o[l * 6 + 0] = r[0]; o[l * 6 + 1] = r[1]; o[l * 6 + 2] = q[0]; o[l * 6 + 3] = q[1]; o[l * 6 + 4] = x4; o[l * 6 + 5] = r8;
}
int main()
{
    int *d; hipMalloc(&d, 64 * 6 * 4); k<<<1, 64>>>(d); int h[384]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 1) if (l % 16 < 2 || l % 16 == 4 || l%16==9) printf("l=%2d swap32: %d %d  swap16: %d %d  xor4: %d ror8: %d\n", l, h[l*6], h[l*6+1], h[l*6+2], h[l*6+3], h[l*6+4], h[l*6+5]);
    return 0;
}
