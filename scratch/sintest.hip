// accuracy of v_sin_f32 / v_cos_f32 (input in revolutions) and v_rcp/v_rsq on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const float *x, float *s, float *c, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { s[i] = __builtin_amdgcn_sinf(x[i]); c[i] = __builtin_amdgcn_cosf(x[i]); }
}
int main()
{
    const int n = 1 << 22;
    std::vector<float> hx(n), hs(n), hc(n);
    for (int i = 0; i < n; i++) hx[i] = -2.0f + 4.0f * (float)i / (float)n;
    float *dx, *ds, *dc;
    hipMalloc(&dx, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, ds, dc, n);
    hipMemcpy(hs.data(), ds, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hc.data(), dc, n * 4, hipMemcpyDeviceToHost);
    double es = 0, ec = 0, es1 = 0, ec1 = 0;
    for (int i = 0; i < n; i++) {
        double t = (double)hx[i] * 6.283185307179586;
        double a = fabs(hs[i] - sin(t)), b = fabs(hc[i] - cos(t));
        if (a > es) es = a; if (b > ec) ec = b;
        if (hx[i] >= 0 && hx[i] < 1) { if (a > es1) es1 = a; if (b > ec1) ec1 = b; }
    }
    printf("v_sin max abs err [-2,2): %.3e  [0,1): %.3e\nv_cos max abs err [-2,2): %.3e  [0,1): %.3e\n", es, es1, ec, ec1);
    return 0;
}
