// msresamp.hip -- multi-stage arbitrary resampler (decimating front end, rate <= 1; interpolating, rate > 1) for gfx950.
//
// Replaces liquid's msresamp_crcf as the reference applications use it in front of a
// synchronizer (pattern: src/flexframe_rx.cc:179 `msresamp_crcf_create(rate, 60.0f)`,
// :240 `msresamp_crcf_execute(q, in, nin, out, &nout)`; rate computed like
// src/multichannel_rx.cc:129-138).  Structure (liquid msresamp / resamp2 / resamp):
//   * half-band decimators while rate < 0.5 (4m+1 = 29-tap Kaiser half-band, m = 7):
//       y[k] = 0.5 * ( x[2k-13] + sum_{i<14} h1[i] * x[2(k-13+i)] )
//   * a 256-branch polyphase arbitrary resampler (14 taps per branch, fc = min(0.515 r, 0.49))
//     stepped by a 24-bit fixed-point phase: output j comes from input n_j = (j*step) >> 24 with
//     branch b_j = ((j*step) mod 2^24) >> 16:   y[j] = sum_{k<14} H[b_j][k] * x[n_j-13+k]
// Interpolating (rate > 1, the transmit applications' msresamp_crcf_create(2.0, 60), src/flexframe_tx.cc:170): the
// arbitrary stage first (rate in (1, 2]: same closed form, step < 2^24), then half-band interpolators while rate > 2:
//       v[2k] = u[k-7],   v[2k+1] = sum_{i<14} h1[i] * u[k-13+i]
// Every output is a closed form of its index, so each stage is one embarrassingly parallel
// kernel (64-bit integer phase, exact); overlapping 14/27-sample windows are served by L1/L2.
// Streaming state = absolute sample counters on the host + retained tails of each stage buffer.
#include "../../include/mcrx_hip.h"
#include "design.hpp"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

using namespace mcrx;

#define RS_M 7
#define RS_TAPS (2 * RS_M)          // taps per polyphase branch / half-band filter branch
#define RS_NPFB 256
#define RS_PHASE_BITS 24
#define RS_KEEP 32                  // samples of history retained per stage (>= 27)

// in: samples with absolute index [in_base, ...); out[k - k0] for k in [k0, k1)
__global__ void halfband_kernel(const float2 *in, long long in_base, float2 *out, long long k0, long long k1,
                                const float *h1)
{
    const long long k = k0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= k1) return;
    float2 acc = make_float2(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < RS_TAPS; i++) {
        const long long t = 2 * (k - (RS_TAPS - 1) + i);
        if (t >= 0) { const float2 v = in[t - in_base]; const float h = h1[i]; acc.x += h * v.x; acc.y += h * v.y; }
    }
    const long long td = 2 * k - (RS_TAPS - 1);            // delay branch: x[2k-13]
    float2 d = make_float2(0.f, 0.f);
    if (td >= 0) d = in[td - in_base];
    out[k - k0] = make_float2(0.5f * (d.x + acc.x), 0.5f * (d.y + acc.y));
}

// half-band interpolator: inputs k in [k0, k1) -> outputs 2k, 2k+1 at out[2 (k - k0)]
__global__ void halfband_interp_kernel(const float2 *in, long long in_base, float2 *out, long long k0, long long k1,
                                       const float *h1)
{
    const long long k = k0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= k1) return;
    float2 acc = make_float2(0.f, 0.f);
#pragma unroll
    for (int i = 0; i < RS_TAPS; i++) {
        const long long t = k - (RS_TAPS - 1) + i;
        if (t >= 0) { const float2 v = in[t - in_base]; const float h = h1[i]; acc.x += h * v.x; acc.y += h * v.y; }
    }
    const long long td = k - RS_M;
    const float2 d = td >= 0 ? in[td - in_base] : make_float2(0.f, 0.f);
    reinterpret_cast<float4 *>(out)[k - k0] = make_float4(d.x, d.y, acc.x, acc.y);
}

__global__ void arbitrary_kernel(const float2 *in, long long in_base, float2 *out, long long j0, long long j1,
                                 unsigned long long step, const float *hpfb)
{
    const long long j = j0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= j1) return;
    const unsigned long long P = (unsigned long long)j * step;
    const long long n = (long long)(P >> RS_PHASE_BITS);
    const unsigned b = (unsigned)((P & ((1ull << RS_PHASE_BITS) - 1)) >> (RS_PHASE_BITS - 8));
    const float *h = hpfb + (size_t)b * RS_TAPS;
    float2 acc = make_float2(0.f, 0.f);
#pragma unroll
    for (int k = 0; k < RS_TAPS; k++) {
        const long long t = n - (RS_TAPS - 1) + k;
        if (t >= 0) { const float2 v = in[t - in_base]; acc.x += h[k] * v.x; acc.y += h[k] * v.y; }
    }
    out[j - j0] = acc;
}

static thread_local std::string g_rs_err;
#define RSCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { g_rs_err = hipGetErrorString(e_); return MCRX_EHIP; } } while (0)

struct StageBuf {                   // device buffer holding samples [base, end) of one stage input
    float2 *d = nullptr; size_t cap = 0; long long base = 0, end = 0;
};

struct msresamp_hip_s {
    float rate = 1, As = 60;
    bool interp = false;
    unsigned num_stages = 0;
    double rate_arb = 1;
    unsigned long long step = 0;
    float *d_h1 = nullptr, *d_hpfb = nullptr;
    std::vector<StageBuf> in;       // decimating: in[0] = resampler input, in[s] = input of half-band s / arbitrary stage;
                                    // interpolating: in[0] = input of the arbitrary stage, in[1 + s] = input of half-band interpolator s
    long long out_count = 0;        // outputs produced so far (absolute j)
    hipStream_t stream = nullptr;
};

static int stage_reserve(msresamp_hip_t q, StageBuf &b, size_t extra, hipStream_t st)
{
    // make room for `extra` more samples, keeping the last RS_KEEP samples
    size_t have = (size_t)(b.end - b.base);
    if (have + extra <= b.cap) return MCRX_OK;
    size_t keep = std::min(have, (size_t)RS_KEEP);
    size_t ncap = std::max(b.cap, 2 * (keep + extra) + 64);
    float2 *nd = nullptr;
    RSCHK(hipMalloc((void **)&nd, ncap * sizeof(float2)));
    if (keep) RSCHK(hipMemcpyAsync(nd, b.d + (have - keep), keep * sizeof(float2), hipMemcpyDeviceToDevice, st));
    RSCHK(hipStreamSynchronize(st));
    if (b.d) (void)hipFree(b.d);
    b.d = nd; b.cap = ncap; b.base = b.end - (long long)keep;
    (void)q;
    return MCRX_OK;
}

extern "C" int msresamp_hip_create(msresamp_hip_t *out, float rate, float As)
{
    if (!out || !(rate > 0.0f) || rate > 1024.0f) { g_rs_err = "msresamp: rate must be in (0, 1024]"; return MCRX_EINVAL; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_rs_err = "no HIP device (no CPU fallback)"; return MCRX_EHIP; }
    msresamp_hip_t q = new msresamp_hip_s();
    q->rate = rate; q->As = As; q->rate_arb = rate;
    q->interp = rate > 1.0f;
    if (q->interp) while (q->rate_arb > 2.0) { q->num_stages++; q->rate_arb *= 0.5; }
    else           while (q->rate_arb < 0.5) { q->num_stages++; q->rate_arb *= 2.0; }
    // half-band branch filter: odd taps of a 29-tap Kaiser design with fc = 0.25, reversed
    std::vector<float> h = firdes_kaiser(4 * RS_M + 1, 0.25f, As), h1(RS_TAPS);
    unsigned j = 0;
    for (unsigned i = 1; i < 4 * RS_M + 1; i += 2) h1[j++] = h[4 * RS_M + 1 - i - 1];
    // arbitrary resampler bank, unity DC gain per branch
    float fc = 0.515f * (float)q->rate_arb; if (fc > 0.49f) fc = 0.49f;
    const unsigned n = 2 * RS_M * RS_NPFB + 1;
    std::vector<float> hf = firdes_kaiser(n, fc / (float)RS_NPFB, As), hp((size_t)RS_NPFB * RS_TAPS);
    double gain = 0; for (float v : hf) gain += v;
    gain = (double)RS_NPFB / gain;
    for (unsigned b = 0; b < RS_NPFB; b++)
        for (unsigned k = 0; k < RS_TAPS; k++)
            hp[(size_t)b * RS_TAPS + (RS_TAPS - 1 - k)] = (float)((double)hf[b + k * RS_NPFB] * gain);
    q->step = (unsigned long long)std::llrint((double)(1u << RS_PHASE_BITS) / q->rate_arb);
    if (hipMalloc((void **)&q->d_h1, h1.size() * sizeof(float)) != hipSuccess ||
        hipMalloc((void **)&q->d_hpfb, hp.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(q->d_h1, h1.data(), h1.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(q->d_hpfb, hp.data(), hp.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipStreamCreate(&q->stream) != hipSuccess) { g_rs_err = "device allocation failed"; delete q; return MCRX_EHIP; }
    q->in.resize(q->num_stages + 1);     // (interpolating: in[1 .. num_stages] feed the half-band interpolators)
    *out = q;
    return MCRX_OK;
}

extern "C" int msresamp_hip_destroy(msresamp_hip_t q)
{
    if (!q) return MCRX_OK;
    (void)hipDeviceSynchronize();
    for (auto &b : q->in) if (b.d) (void)hipFree(b.d);
    (void)hipFree(q->d_h1); (void)hipFree(q->d_hpfb);
    if (q->stream) (void)hipStreamDestroy(q->stream);
    delete q;
    return MCRX_OK;
}

extern "C" int msresamp_hip_reset(msresamp_hip_t q)
{
    if (!q) return MCRX_EINVAL;
    for (auto &b : q->in) { b.base = 0; b.end = 0; }
    q->out_count = 0;
    return MCRX_OK;
}

extern "C" float msresamp_hip_get_delay(msresamp_hip_t q)
{
    if (!q) return 0.f;
    float d = (float)RS_M;
    for (unsigned i = 0; i < q->num_stages; i++) d = 2.0f * d + (float)(2 * RS_M - 1);
    return d;
}

extern "C" size_t msresamp_hip_max_output(msresamp_hip_t q, size_t nin)
{ return q ? (size_t)((double)nin * q->rate * 1.0001) + (q->interp ? (4u << q->num_stages) : 4) : 0; }

// d_in: nin new input samples in device memory; d_out receives *nout <= out_cap samples
extern "C" int msresamp_hip_execute_device(msresamp_hip_t q, const void *d_in, size_t nin, void *d_out,
                                           size_t out_cap, size_t *nout, void *stream)
{
    if (!q || !nout || (!d_in && nin) || !d_out) { g_rs_err = "null argument"; return MCRX_EINVAL; }
    hipStream_t st = stream ? (hipStream_t)stream : q->stream;
    *nout = 0;
    // append the new samples to the first stage buffer
    StageBuf &b0 = q->in[0];
    int rc;
    if ((rc = stage_reserve(q, b0, nin, st))) return rc;
    if (nin) RSCHK(hipMemcpyAsync(b0.d + (b0.end - b0.base), d_in, nin * sizeof(float2), hipMemcpyDeviceToDevice, st));
    b0.end += (long long)nin;
    if (q->interp) {
        // arbitrary stage over in[0]: outputs j with n_j < end go to the first half-band interpolator's input (or out)
        const long long j0 = q->out_count;
        const unsigned long long lim = (unsigned long long)b0.end << RS_PHASE_BITS;
        long long j1 = (long long)((lim + q->step - 1) / q->step);
        if (j1 < j0) j1 = j0;
        const size_t total_out = (size_t)(j1 - j0) << q->num_stages;
        if (total_out > out_cap) { g_rs_err = "output buffer too small"; return MCRX_EINVAL; }
        float2 *dst = (float2 *)d_out;
        if (q->num_stages) {
            StageBuf &b1 = q->in[1];
            if ((rc = stage_reserve(q, b1, (size_t)(j1 - j0), st))) return rc;
            dst = b1.d + (b1.end - b1.base);
            b1.end += j1 - j0;
        }
        if (j1 > j0) {
            const unsigned n = (unsigned)(j1 - j0);
            hipLaunchKernelGGL(arbitrary_kernel, dim3((n + 255) / 256), dim3(256), 0, st, b0.d, b0.base, dst, j0, j1, q->step, q->d_hpfb);
            RSCHK(hipGetLastError());
        }
        q->out_count = j1;
        // half-band interpolators: stage s consumes the new samples of in[1 + s] (all of them: no look-ahead needed)
        long long k0 = j0;
        for (unsigned s = 0; s < q->num_stages; s++) {
            StageBuf &bi = q->in[1 + s];
            const long long k1 = bi.end;
            float2 *o = (float2 *)d_out;
            if (s + 1 < q->num_stages) {
                StageBuf &bo = q->in[2 + s];
                if ((rc = stage_reserve(q, bo, (size_t)(2 * (k1 - k0)), st))) return rc;
                o = bo.d + (bo.end - bo.base);
                bo.end += 2 * (k1 - k0);
            }
            if (k1 > k0) {
                const unsigned n = (unsigned)(k1 - k0);
                hipLaunchKernelGGL(halfband_interp_kernel, dim3((n + 255) / 256), dim3(256), 0, st, bi.d, bi.base, o, k0, k1, q->d_h1);
                RSCHK(hipGetLastError());
            }
            k0 *= 2;
        }
        *nout = total_out;
        return MCRX_OK;
    }
    // half-band stages: stage s consumes in[s], appends to in[s+1]
    for (unsigned s = 0; s < q->num_stages; s++) {
        StageBuf &bi = q->in[s], &bo = q->in[s + 1];
        const long long k0 = bo.end, k1 = bi.end / 2;       // output k exists once x[2k+1] has arrived
        if (k1 > k0) {
            if ((rc = stage_reserve(q, bo, (size_t)(k1 - k0), st))) return rc;
            const unsigned n = (unsigned)(k1 - k0);
            hipLaunchKernelGGL(halfband_kernel, dim3((n + 255) / 256), dim3(256), 0, st,
                               bi.d, bi.base, bo.d + (bo.end - bo.base), k0, k1, q->d_h1);
            RSCHK(hipGetLastError());
            bo.end = k1;
        }
    }
    // arbitrary stage over in[num_stages]: outputs j with n_j < end
    StageBuf &ba = q->in[q->num_stages];
    const long long j0 = q->out_count;
    // largest j with (j*step >> 24) < end  <=>  j*step < end << 24
    const unsigned long long lim = (unsigned long long)ba.end << RS_PHASE_BITS;
    long long j1 = (long long)((lim + q->step - 1) / q->step);          // first j with j*step >= lim
    if (j1 < j0) j1 = j0;
    if ((size_t)(j1 - j0) > out_cap) { g_rs_err = "output buffer too small"; return MCRX_EINVAL; }
    if (j1 > j0) {
        const unsigned n = (unsigned)(j1 - j0);
        hipLaunchKernelGGL(arbitrary_kernel, dim3((n + 255) / 256), dim3(256), 0, st,
                           ba.d, ba.base, (float2 *)d_out, j0, j1, q->step, q->d_hpfb);
        RSCHK(hipGetLastError());
    }
    q->out_count = j1;
    *nout = (size_t)(j1 - j0);
    return MCRX_OK;
}

extern "C" const char *msresamp_hip_last_error(void) { return g_rs_err.c_str(); }
