// pfb2.hip -- 2x-oversampled polyphase analysis bank (liquid firpfbch2_crcf, analyzer) for gfx950.
//
// liquid-usrp's receiver uses the critically sampled bank (channelizer.hip); this is the alternate front end
// BASELINE.json's north_star names, as a stage-level operator with its own parity
// tests.  Closed form of the two-phase window shuffle of firpfbch2_crcf_execute_analyzer (verified against the
// oracle's state machine and a float64 direct-form model, tests/test_oracle_dsp.py):
//     Z_s[r] = sum_{k < 2m} h[r + k M] u[(s+1) M/2 - 1 - r - k M]          r = 0 .. M-1
//     y_s[n] = (-1)^(n s) / M * sum_r Z_s[r] e^{+j 2 pi r n / M}
// i.e. every step is a polyphase FIR over the last 2mM samples and an M-point inverse FFT, the odd steps with
// alternating output signs.  pfb2_kernel: one workgroup per step, thread r gathers its residue (coalesced across
// r, the 2m-fold reuse of every sample is served by L2), radix-2 Stockham inverse FFT in LDS -- any m, any M.
// pfb2_tile_kernel (m = 7, M >= 64): eight steps per workgroup on register sliding windows, see below.
// Algorithmic HBM bytes per input sample: 8 read + 2 x 8 written (M outputs per M/2 inputs) = 24 B.
// The critically sampled bank (channelizer.hip) is the measured hot path; this one is tuned one level only.
#include "../../include/mcrx_hip.h"
#include "design.hpp"
#include "devmath.h"
#include "devscope.hpp"

#include <hip/hip_runtime.h>
#include <type_traits>
#include <string>
#include <vector>

namespace mcrx {

struct Pfb2Args {
    const float2 *x;            // x[0] = absolute sample first_step * M/2; `lead` valid samples precede it
    const float *taps;          // 2m * M
    float2 *out;                // [nsteps][M]
    long long lead;
    uint32_t nsteps, first_step, p;
};

template <int M>
__global__ void pfb2_kernel(Pfb2Args a)
{
    constexpr int T = (M / 2 < 64) ? 64 : M / 2;
    __shared__ float2 buf[2][M];
    const uint32_t sl = blockIdx.x;                             // step within this call
    const int tid = threadIdx.x;
    const long long A = ((long long)sl + 1) * (M / 2) - 1;      // newest sample of the step, relative to x
    for (int r = tid; r < M; r += T) {
        float2 acc = make_float2(0.f, 0.f);
        for (int k = (int)a.p - 1; k >= 0; k--) {               // oldest first, like the window dot product
            const long long idx = A - r - (long long)k * M;
            const float h = a.taps[r + k * M];
            if (idx >= -a.lead) { const float2 u = a.x[idx]; acc.x += h * u.x; acc.y += h * u.y; }
        }
        buf[0][r] = acc;
    }
    __syncthreads();
    int cur = 0;
    for (int n = M, s = 1; n > 1; n >>= 1, s <<= 1) {           // Stockham autosort, decimation in frequency
        const int m2 = n >> 1;
        for (int q = tid; q < M / 2; q += T) {
            const int p = q / s, r = q % s;
            float sn, cs; sincos_u32((uint32_t)p * (uint32_t)(4294967296.0 / n), sn, cs);
            const float2 w = make_float2(cs, sn);               // e^{+j 2 pi p / n}: inverse transform
            const float2 u = buf[cur][r + s * p], v = buf[cur][r + s * (p + m2)];
            buf[cur ^ 1][r + s * 2 * p] = cadd(u, v);
            buf[cur ^ 1][r + s * (2 * p + 1)] = cmul(csub(u, v), w);
        }
        __syncthreads();
        cur ^= 1;
    }
    const bool odd_step = ((a.first_step + sl) & 1u) != 0;
    const float g = 1.0f / (float)M;
    float2 *dst = a.out + (size_t)sl * M;
    for (int n = tid; n < M; n += T) {
        const float sg = (odd_step && (n & 1)) ? -g : g;
        dst[n] = make_float2(buf[cur][n].x * sg, buf[cur][n].y * sg);
    }
}

// Few channels (M <= 32, up to 32 taps per branch): one step is far too little work for a workgroup.  Here 256 threads
// own 8 x 256/M consecutive steps: their input span (<= 2048 samples) and the taps are staged in LDS once, thread
// (step, r) makes Z_s[r] from LDS, and the M-point inverse transform is a direct DFT over the step's M values (M complex
// multiply-adds per output against an M-entry twiddle table; LDS reads of one step's Z are broadcasts).  Loads and
// stores are fully coalesced; HBM traffic is the algorithmic 24 B per input sample.
constexpr int PS_IT = 8, PS_MAXP = 32;
template <int M>
__global__ __launch_bounds__(256) void pfb2_small_kernel(Pfb2Args a)
{
    constexpr int G = 256 / M, S = PS_IT * G;                   // steps per pass, steps per workgroup
    __shared__ float2 xin[(S - 1) * (M / 2) + PS_MAXP * M];
    __shared__ float taps[PS_MAXP * M];
    __shared__ float2 Z[256], tw[M];
    const int tid = threadIdx.x, r = tid % M, g = tid / M;
    const long long sb = (long long)blockIdx.x * S;
    const int P = (int)a.p;
    const long long total = (long long)a.nsteps * (M / 2);
    const long long lo = (sb + 1) * (M / 2) - (long long)P * M;
    const int steps = (int)min((long long)S, (long long)a.nsteps - sb);
    const int span = (steps - 1) * (M / 2) + P * M;
    for (int i = tid; i < span; i += 256) {
        const long long idx = lo + i;
        xin[i] = (idx >= -a.lead && idx < total) ? a.x[idx] : make_float2(0.f, 0.f);
    }
    for (int i = tid; i < P * M; i += 256) taps[i] = a.taps[i];
    if (tid < M) { float sn, cs; sincos_u32((uint32_t)tid * (uint32_t)(4294967296.0 / M), sn, cs); tw[tid] = make_float2(cs, sn); }
    __syncthreads();
    const float gn = 1.0f / (float)M;
    for (int it = 0; it < PS_IT; it++) {
        const int sl = it * G + g;                              // step within the workgroup
        float2 acc = make_float2(0.f, 0.f);
        if (sl < steps) {
            const int base = sl * (M / 2) - 1 - r + P * M;      // xin index of u[A_s - r], A_s = newest sample of the step
            for (int k = P - 1; k >= 0; k--) {                  // oldest first, like the window dot product
                const float2 u = xin[base - k * M];
                const float h = taps[r + k * M];
                acc.x += h * u.x; acc.y += h * u.y;
            }
        }
        Z[tid] = acc;
        __syncthreads();
        if (sl < steps) {
            float2 y = make_float2(0.f, 0.f);
#pragma unroll
            for (int rr = 0; rr < M; rr++) {
                const float2 z = Z[g * M + rr], w = tw[(rr * r) & (M - 1)];
                y.x += z.x * w.x - z.y * w.y;
                y.y += z.x * w.y + z.y * w.x;
            }
            const bool odd_step = ((a.first_step + (uint32_t)(sb + sl)) & 1u) != 0;
            const float sg = (odd_step && (r & 1)) ? -gn : gn;
            a.out[(size_t)(sb + sl) * M + r] = make_float2(y.x * sg, y.y * sg);
        }
        __syncthreads();
    }
}

// The same for the usual prototype length (m = 7, 14 taps per branch) and M >= 64: a workgroup owns TS = 8
// consecutive steps and a thread one residue r.  Steps of equal parity slide one position along the thread's
// polyphase stream: the thread walks the 17 even-step and 17 odd-step samples of its stream once and adds every
// sample into the (up to four) steps whose window holds it -- 2 x 17 / 8 = 4.25 fetches per output instead of 14,
// accumulated oldest first like the window dot product, with eight accumulators and no register window, so that
// the kernel needs few registers; for M = 1024 a thread owns two residues and two 512-thread workgroups share a
// CU (one loads while the other transforms).
// The inverse FFTs go through a ping-pong LDS tile four at a time (one barrier per stage for all four), twiddles
// from an M-entry table.  Workgroups are dealt round robin to the 8 XCDs, each with its own L2, and neighbouring
// tiles share 13/17 of their input: every XCD gets a contiguous run of tiles, so the tiles in flight on one XCD are
// neighbours and HBM traffic is the algorithmic 24 B per input sample (profiles/r2_resamp_roofline.csv).
constexpr int PF_P = 14, PF_TS = 8, PF_TB = 4;
template <int M> struct Pfb2Tile { static constexpr int NT = (M >= 1024) ? M / 2 : M; };      // threads per workgroup
template <int M>
__global__ __launch_bounds__(Pfb2Tile<M>::NT, 4) void pfb2_tile_kernel(Pfb2Args a)
{
    constexpr int W = PF_P + PF_TS / 2 - 1, NT = Pfb2Tile<M>::NT, R = M / NT;
    extern __shared__ float2 tile[];                            // [2][TB][M], then the twiddle table [M]
    float2 *twd = tile + 2 * PF_TB * M;                         // twd[q] = e^{+j 2 pi q / M}
    const int tid = threadIdx.x;
#pragma unroll
    for (int ri = 0; ri < R; ri++) {
        const int r = tid + ri * NT;
        float sn, cs; sincos_u32((uint32_t)r * (uint32_t)(4294967296.0 / M), sn, cs); twd[r] = make_float2(cs, sn);
    }
    const uint32_t per_xcd = gridDim.x >> 3;                    // (the grid is a multiple of 8 workgroups)
    const long long tile_id = (long long)(blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    const long long sl0 = tile_id * PF_TS;
    if (sl0 >= (long long)a.nsteps) return;
    const long long total = (long long)a.nsteps * (M / 2);
    // u_j (even steps) = x[ub + (M-1-r) + j M], (odd steps) = the same + M/2: a uniform base and a small lane offset
    const long long ub = (sl0 + 1) * (M / 2) - 1 - (long long)(PF_P - 1) * M - (M - 1);
    const bool inside = ub >= -a.lead && ub + (M - 1) + (long long)(W - 1) * M + M / 2 < total;      // (uniform)
    float2 acc[R][PF_TS];
#pragma unroll
    for (int ri = 0; ri < R; ri++) {
        const int r = tid + ri * NT;
        float h[PF_P];
#pragma unroll
        for (int k = 0; k < PF_P; k++) h[k] = a.taps[r + k * M];
#pragma unroll
        for (int sl = 0; sl < PF_TS; sl++) acc[ri][sl] = make_float2(0.f, 0.f);
        const int lane_off = M - 1 - r;
        auto fir = [&](auto clamp) {
            constexpr bool CLAMP = decltype(clamp)::value;
            const float2 *xb = a.x + ub;
#pragma unroll
            for (int j = 0; j < W; j++) {
                float2 e, o;
                if constexpr (CLAMP) {                          // first / last tiles of a call: zeros outside the stream
                    const long long ie = ub + lane_off + (long long)j * M, io = ie + M / 2;
                    const bool ve = ie >= -a.lead && ie < total, vo = io >= -a.lead && io < total;
                    e = a.x[ve ? ie : 0]; o = a.x[vo ? io : 0];
                    e = ve ? e : make_float2(0.f, 0.f);
                    o = vo ? o : make_float2(0.f, 0.f);
                } else {
                    e = xb[lane_off + j * M]; o = xb[lane_off + j * M + M / 2];
                }
#pragma unroll
                for (int sl = 0; sl < PF_TS; sl++) {
                    const int k = sl / 2 + PF_P - 1 - j;        // tap that meets window position j in step sl
                    if (k >= 0 && k < PF_P) {
                        const float2 u = (sl & 1) ? o : e;
                        acc[ri][sl].x += h[k] * u.x; acc[ri][sl].y += h[k] * u.y;
                    }
                }
            }
        };
        if (inside) fir(std::false_type{}); else fir(std::true_type{});
    }
    const float g = 1.0f / (float)M;
#pragma unroll
    for (int bt = 0; bt < PF_TS / PF_TB; bt++) {
#pragma unroll
        for (int ri = 0; ri < R; ri++)
#pragma unroll
            for (int t = 0; t < PF_TB; t++) tile[t * M + tid + ri * NT] = acc[ri][bt * PF_TB + t];
        __syncthreads();
        int cur = 0;
        int n = M, s = 1;
        for (; n >= 4; n >>= 2, s <<= 2) {                      // radix-4 Stockham stages, all TB transforms per stage
            const int n1 = n >> 2;
            float2 *src = tile + cur * (PF_TB * M), *dst = tile + (cur ^ 1) * (PF_TB * M);
            for (int q = tid; q < PF_TB * (M / 4); q += NT) {
                const int t = q / (M / 4), bq = q % (M / 4);
                const int p = bq / s, rr = bq % s;
                const int tq = p * s;                           // p / n revolutions = p * (M / n) / M, and M / n = s
                const float2 w1 = twd[tq], w2 = twd[2 * tq], w3 = twd[3 * tq];
                const float2 *x0 = src + t * M + rr + s * p;
                const float2 xa = x0[0], xb = x0[s * n1], xc = x0[2 * s * n1], xd = x0[3 * s * n1];
                const float2 apc = cadd(xa, xc), amc = csub(xa, xc), bpd = cadd(xb, xd), bmd = csub(xb, xd);
                const float2 jbmd = make_float2(-bmd.y, bmd.x); // +j (b - d): inverse transform
                float2 *y0 = dst + t * M + rr + s * 4 * p;
                y0[0] = cadd(apc, bpd);
                y0[s] = cmul(cadd(amc, jbmd), w1);
                y0[2 * s] = cmul(csub(apc, bpd), w2);
                y0[3 * s] = cmul(csub(amc, jbmd), w3);
            }
            __syncthreads();
            cur ^= 1;
        }
        for (; n > 1; n >>= 1, s <<= 1) {                       // (one radix-2 stage when log2 M is odd)
            const int m2 = n >> 1;
            float2 *src = tile + cur * (PF_TB * M), *dst = tile + (cur ^ 1) * (PF_TB * M);
            for (int q = tid; q < PF_TB * (M / 2); q += NT) {
                const int t = q / (M / 2), bq = q % (M / 2);
                const int p = bq / s, rr = bq % s;
                const float2 w = twd[p * s];
                const float2 u = src[t * M + rr + s * p], v = src[t * M + rr + s * (p + m2)];
                dst[t * M + rr + s * 2 * p] = cadd(u, v);
                dst[t * M + rr + s * (2 * p + 1)] = cmul(csub(u, v), w);
            }
            __syncthreads();
            cur ^= 1;
        }
        const float2 *res = tile + cur * (PF_TB * M);
#pragma unroll
        for (int t = 0; t < PF_TB; t++) {
            const long long sl = sl0 + bt * PF_TB + t;
            if (sl >= (long long)a.nsteps) break;
            const bool odd_step = ((a.first_step + (uint32_t)sl) & 1u) != 0;
#pragma unroll
            for (int ri = 0; ri < R; ri++) {
                const int r = tid + ri * NT;
                const float sg = (odd_step && (r & 1)) ? -g : g;
                a.out[(size_t)sl * M + r] = make_float2(res[t * M + r].x * sg, res[t * M + r].y * sg);
            }
        }
        __syncthreads();                                        // the next batch overwrites the tile
    }
}

}  // namespace mcrx

using namespace mcrx;

static thread_local std::string g_pfb2_err;
#define P2CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { g_pfb2_err = std::string(#x) + ": " + hipGetErrorString(e_); return MCRX_EHIP; } } while (0)

struct mcrx_hip_pfb2_s {
    int device = -1;            // the HIP device the handle was created on: every entry point runs with it current (devscope.hpp)
    unsigned M, m;
    std::vector<float> taps;
    float *d_taps = nullptr;
};

extern "C" const char *mcrx_hip_pfb2_last_error(void) { return g_pfb2_err.c_str(); }

extern "C" int mcrx_hip_pfb2_create(mcrx_hip_pfb2_t *out, unsigned M, unsigned m, float As)
{
    if (!out) return MCRX_EINVAL;
    *out = nullptr;
    // argument checks of firpfbch2_crcf_create: even channel count, filter semi-length at least 1
    if (M < 2 || (M & 1)) { g_pfb2_err = "error: firpfbch2, number of channels must be even (and at least 2)"; return MCRX_EINVAL; }
    if (m < 1) { g_pfb2_err = "error: firpfbch2, filter semi-length must be at least 1"; return MCRX_EINVAL; }
    if ((M & (M - 1)) || M > 1024) { g_pfb2_err = "channel count must be a power of two <= 1024"; return MCRX_EUNSUPP; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_pfb2_err = "no HIP device (no CPU fallback)"; return MCRX_EHIP; }
    mcrx_hip_pfb2_t q = new mcrx_hip_pfb2_s();
    q->device = current_device();
    q->M = M; q->m = m;
    q->taps = pfb2_prototype(M, m, As);
    if (hipMalloc((void **)&q->d_taps, q->taps.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(q->d_taps, q->taps.data(), q->taps.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        if (q->d_taps) (void)hipFree(q->d_taps);
        delete q; g_pfb2_err = "device allocation failed"; return MCRX_ENOMEM;
    }
    *out = q;
    return MCRX_OK;
}

extern "C" int mcrx_hip_pfb2_destroy(mcrx_hip_pfb2_t q)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (!q) return MCRX_OK;
    (void)hipDeviceSynchronize();
    (void)hipFree(q->d_taps);
    delete q;
    return MCRX_OK;
}

extern "C" int mcrx_hip_pfb2_get_taps(mcrx_hip_pfb2_t q, float *h, size_t n)
{
    if (!q || !h || n < q->taps.size()) { g_pfb2_err = "bad argument"; return MCRX_EINVAL; }
    std::copy(q->taps.begin(), q->taps.end(), h);
    return MCRX_OK;
}

extern "C" int mcrx_hip_pfb2_analyze(mcrx_hip_pfb2_t q, const void *d_x, size_t lead_samples, size_t nsteps,
                                     uint64_t first_step, void *d_out, void *stream)
{
    DevScope dev_scope_(q ? q->device : -1);
    if (!q || !d_x || !d_out) { g_pfb2_err = "bad argument"; return MCRX_EINVAL; }
    if (nsteps == 0) return MCRX_OK;
    if (nsteps > 0x7fffffffu) { g_pfb2_err = "too many steps in one call"; return MCRX_EINVAL; }
    Pfb2Args a;
    a.x = (const float2 *)d_x; a.taps = q->d_taps; a.out = (float2 *)d_out;
    a.lead = (long long)lead_samples; a.nsteps = (uint32_t)nsteps; a.first_step = (uint32_t)first_step; a.p = 2 * q->m;
    hipStream_t st = (hipStream_t)stream;
    if (a.p == PF_P && q->M >= 64) {
        const unsigned nwg = ((unsigned)((nsteps + PF_TS - 1) / PF_TS) + 7u) & ~7u;     // whole rounds over the 8 XCDs
        const size_t lds = (size_t)(2 * PF_TB + 1) * q->M * sizeof(float2);
#define P2T(MM) do { static PerDeviceOnce attr_; P2CHK(raise_lds_limit((const void *)pfb2_tile_kernel<MM>, lds, attr_)); \
                     hipLaunchKernelGGL((pfb2_tile_kernel<MM>), dim3(nwg), dim3(Pfb2Tile<MM>::NT), lds, st, a); } while (0)
        switch (q->M) {
        case 64: P2T(64); break;   case 128: P2T(128); break; case 256: P2T(256); break;
        case 512: P2T(512); break; case 1024: P2T(1024); break;
        }
#undef P2T
        P2CHK(hipGetLastError());
        return MCRX_OK;
    }
    if (q->M <= 32 && a.p <= (unsigned)PS_MAXP) {
#define P2S(MM) hipLaunchKernelGGL((pfb2_small_kernel<MM>), dim3((unsigned)((nsteps + PS_IT * (256 / MM) - 1) / (PS_IT * (256 / MM)))), dim3(256), 0, st, a)
        switch (q->M) {
        case 2: P2S(2); break; case 4: P2S(4); break; case 8: P2S(8); break; case 16: P2S(16); break; case 32: P2S(32); break;
        }
#undef P2S
        P2CHK(hipGetLastError());
        return MCRX_OK;
    }
#define P2(MM) hipLaunchKernelGGL((pfb2_kernel<MM>), dim3((unsigned)nsteps), dim3((MM) / 2 < 64 ? 64 : (MM) / 2), 0, st, a)
    switch (q->M) {
    case 2: P2(2); break;       case 4: P2(4); break;     case 8: P2(8); break;     case 16: P2(16); break;
    case 32: P2(32); break;     case 64: P2(64); break;   case 128: P2(128); break; case 256: P2(256); break;
    case 512: P2(512); break;   case 1024: P2(1024); break;
    default: g_pfb2_err = "unsupported channel count"; return MCRX_EUNSUPP;
    }
#undef P2
    P2CHK(hipGetLastError());
    return MCRX_OK;
}
