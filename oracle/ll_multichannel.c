/*
 * ll_multichannel.c -- CPU ORACLE (test infrastructure): the reference's own sample
 * flows restated over the liquidlite objects, plus the multi-stage resampler.
 *
 *   ll_mcrx_*  follows /root/reference/lib/multichannelrx.cc:
 *              ctor :45-104 (N syncs; analysis bank K=2N, m=7, As=60; VCO at
 *              -0.5(N-1)/N*pi), Reset :135-153 (NCO deliberately not reset),
 *              Execute :155-182 (mix down, step, buffer K samples),
 *              RunChannelizer :185-195 (analyzer, bins 0..N-1 -> synchronizer i).
 *   ll_mctx_*  follows /root/reference/lib/multichanneltx.cc:
 *              ctor :41-100 (N framegens CRC-32/none/Hamming128/QPSK; synthesis bank
 *              K=2N, m=13), Reset :135-149, IsChannelReadyForData :152-162,
 *              UpdateData :165-189, GenerateSamples :192-227, GenerateFrameSamples :230-242.
 *   ll_msresamp_* restates liquid-dsp src/filter/src/{msresamp,resamp.fixed,resamp2}.c as
 *              called by the reference's front-end pattern (src/flexframe_rx.cc:179,240):
 *              half-band decimators while r < 0.5, then a 256-branch polyphase arbitrary
 *              resampler (m=7, fc=min(0.515 r,0.49)) stepped by a 24-bit fixed-point phase.
 *              Half-band stage design parameters are [UPSTREAM-UNVERIFIED] (m=7 each).
 */
#include "liquidlite.h"
#include <omp.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

/* ================================================================== multichannelrx */
struct ll_resamp2_s;
struct ll_mcrx_s {
    unsigned N, M, cp, taper;
    /* alternate front end (ll_mcrx_set_front_end(q, 1)): 2N-channel 2x-oversampled bank + a half-band decimator per channel */
    int front_end; ll_firpfbch2 ch2; struct ll_resamp2_s *hb; ll_cf *Y2, *pair; unsigned step;
    ll_firpfbch ch;
    ll_cf *x, *X;
    unsigned buffer_index;
    ll_ofdmflexframesync *fs;
    ll_nco nco;
    ll_cf *par_buf; size_t par_cap;     /* ll_mcrx_execute_parallel: channel streams [N][nblocks] */
};

static float mc_offset(unsigned N)
{
    float f = -0.5f * (float)(N - 1) / (float)N;
    return (float)((double)f * M_PI);
}

/* a callback that only counts (benchmark legs: no per-frame work outside the receiver itself); userdata -> ll_frame_counter */
int ll_counting_callback(unsigned char *header, int header_valid, unsigned char *payload, unsigned payload_len,
                         int payload_valid, ll_framesyncstats stats, void *userdata)
{
    (void)header; (void)payload; (void)stats;
    ll_frame_counter *k = (ll_frame_counter *)userdata;
    if (!k) return 0;
    __atomic_add_fetch(&k->frames, 1, __ATOMIC_RELAXED);
    if (header_valid) __atomic_add_fetch(&k->headers_valid, 1, __ATOMIC_RELAXED);
    if (payload_valid) { __atomic_add_fetch(&k->payloads_valid, 1, __ATOMIC_RELAXED); __atomic_add_fetch(&k->bytes, payload_len, __ATOMIC_RELAXED); }
    return 0;
}

ll_mcrx ll_mcrx_create(unsigned N, unsigned M, unsigned cp, unsigned taper, const unsigned char *p,
                       void **userdata, ll_framesync_callback *cb)
{
    if (N < 1 || M < 8 || cp < 1 || taper > cp) return NULL;
    ll_mcrx q = (ll_mcrx)calloc(1, sizeof(*q));
    q->N = N; q->M = M; q->cp = cp; q->taper = taper;
    q->fs = (ll_ofdmflexframesync *)calloc(N, sizeof(ll_ofdmflexframesync));
    for (unsigned i = 0; i < N; i++)
        q->fs[i] = ll_ofdmflexframesync_create(M, cp, taper, p, cb ? cb[i] : NULL, userdata ? userdata[i] : NULL);
    q->ch = ll_firpfbch_create_kaiser(LL_ANALYZER, 2 * N, 7, 60.0f);
    q->X = (ll_cf *)calloc(2 * N, sizeof(ll_cf));
    q->x = (ll_cf *)calloc(2 * N, sizeof(ll_cf));
    ll_nco_reset(&q->nco);
    ll_nco_set_frequency(&q->nco, mc_offset(N));
    ll_mcrx_reset(q);
    return q;
}
static void mcrx_free_oversampled(ll_mcrx q);
void ll_mcrx_destroy(ll_mcrx q)
{
    if (!q) return;
    mcrx_free_oversampled(q);
    for (unsigned i = 0; i < q->N; i++) ll_ofdmflexframesync_destroy(q->fs[i]);
    ll_firpfbch_destroy(q->ch);
    free(q->fs); free(q->X); free(q->x); free(q->par_buf); free(q);
}
static void mcrx_reset_oversampled(ll_mcrx q);
void ll_mcrx_reset(ll_mcrx q)
{
    mcrx_reset_oversampled(q);
    for (unsigned i = 0; i < q->N; i++) ll_ofdmflexframesync_reset(q->fs[i]);
    ll_firpfbch_reset(q->ch);
    memset(q->X, 0, sizeof(ll_cf) * 2 * q->N);
    memset(q->x, 0, sizeof(ll_cf) * 2 * q->N);
    q->buffer_index = 0;
}
void ll_mcrx_set_soft(ll_mcrx q, int s)
{ for (unsigned i = 0; i < q->N; i++) ll_ofdmflexframesync_set_soft(q->fs[i], s); }

static void mcrx_execute_oversampled(ll_mcrx q, const ll_cf *xin, unsigned n);
void ll_mcrx_execute(ll_mcrx q, const ll_cf *xin, unsigned n)
{
    if (q->front_end) { mcrx_execute_oversampled(q, xin, n); return; }
    unsigned K = 2 * q->N;
    for (unsigned i = 0; i < n; i++) {
        q->x[q->buffer_index] = ll_nco_mix_down(&q->nco, xin[i]);
        ll_nco_step(&q->nco);
        if (++q->buffer_index == K) {
            q->buffer_index = 0;
            ll_firpfbch_analyzer_execute(q->ch, q->x, q->X);
            for (unsigned c = 0; c < q->N; c++) ll_ofdmflexframesync_execute(q->fs[c], &q->X[c], 1);
        }
    }
}
/* The same result as ll_mcrx_execute on whole blocks, using every host core: the bench's "all cores" CPU leg.
 * The reference's Execute() is single threaded (lib/multichannelrx.cc:155-195); this is what its loop allows:
 * the analysis bank is independent across time given the 13 preceding blocks (each thread warms a private bank
 * on them; the oscillator phase is theta0 + t * dtheta), the synchronizers are independent across channels.
 * Callbacks arrive grouped by channel instead of by time.  Falls back to the serial loop for short inputs. */
void ll_mcrx_execute_parallel(ll_mcrx q, const ll_cf *xin, unsigned n, int nthreads)
{
    unsigned K = 2 * q->N, N = q->N;
    unsigned nblocks = n / K;
    if (nthreads < 1) nthreads = 1;
    if (q->buffer_index != 0 || nblocks < 64 * (unsigned)nthreads || nthreads == 1) { ll_mcrx_execute(q, xin, n); return; }
    double t_start = omp_get_wtime();
    if ((size_t)nblocks * N > q->par_cap) { free(q->par_buf); q->par_cap = (size_t)nblocks * N; q->par_buf = (ll_cf *)malloc(sizeof(ll_cf) * q->par_cap); }
    ll_cf *chan = q->par_buf;                               /* kept between calls: first-touch page faults are paid once */
    const uint32_t theta0 = q->nco.theta, dth = q->nco.d_theta;
    const unsigned H = 13;                              /* 2m - 1 blocks of history */
    ll_firpfbch *bank = (ll_firpfbch *)calloc((size_t)nthreads, sizeof(ll_firpfbch));
    bank[0] = q->ch;
#pragma omp parallel for num_threads(nthreads) schedule(static, 1)
    for (int t = 0; t < nthreads; t++) {
        unsigned b0 = (unsigned)((unsigned long long)nblocks * (unsigned)t / (unsigned)nthreads);
        unsigned b1 = (unsigned)((unsigned long long)nblocks * (unsigned)(t + 1) / (unsigned)nthreads);
        ll_cf *x = (ll_cf *)malloc(sizeof(ll_cf) * K), *X = (ll_cf *)malloc(sizeof(ll_cf) * K);
        if (t) bank[t] = ll_firpfbch_create_kaiser(LL_ANALYZER, K, 7, 60.0f);       /* private bank, designed in parallel */
        for (unsigned b = (t ? b0 - H : b0); b < b1; b++) {
            ll_nco nco = { theta0 + (uint32_t)((unsigned long long)b * K) * dth, dth };
            for (unsigned i = 0; i < K; i++) { x[i] = ll_nco_mix_down(&nco, xin[(size_t)b * K + i]); ll_nco_step(&nco); }
            ll_firpfbch_analyzer_execute(bank[t], x, X);
            if (b >= b0) for (unsigned c = 0; c < N; c++) chan[(size_t)c * nblocks + b] = X[c];   /* [channel][block] */
        }
        free(x); free(X);
    }
    double t_bank = omp_get_wtime();
    if (nthreads > 1) ll_firpfbch_copy_state(q->ch, bank[nthreads - 1]);
    for (int t = 1; t < nthreads; t++) ll_firpfbch_destroy(bank[t]);
    free(bank);
    q->nco.theta = theta0 + (uint32_t)((unsigned long long)nblocks * K) * dth;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
    for (unsigned c = 0; c < N; c++)
        for (unsigned b = 0; b < nblocks; b++) ll_ofdmflexframesync_execute(q->fs[c], &chan[(size_t)c * nblocks + b], 1);
    if (getenv("LL_ORACLE_TIMING")) fprintf(stderr, "ll_mcrx_execute_parallel: banks %.3f s, synchronizers %.3f s (%d threads)\n", t_bank - t_start, omp_get_wtime() - t_bank, nthreads);
    if (n > nblocks * K) ll_mcrx_execute(q, xin + (size_t)nblocks * K, n - nblocks * K);
}

void ll_mcrx_channelize(ll_mcrx q, const ll_cf *xin, unsigned nblocks, ll_cf *out)
{
    unsigned K = 2 * q->N;
    for (unsigned b = 0; b < nblocks; b++) {
        for (unsigned i = 0; i < K; i++) {
            q->x[i] = ll_nco_mix_down(&q->nco, xin[(size_t)b * K + i]);
            ll_nco_step(&q->nco);
        }
        ll_firpfbch_analyzer_execute(q->ch, q->x, q->X);
        memcpy(out + (size_t)b * q->N, q->X, sizeof(ll_cf) * q->N);
    }
}

/* ================================================================== multichanneltx */
struct ll_mctx_s {
    unsigned N, M, cp, taper;
    ll_firpfbch ch;
    ll_cf *X, *x;
    ll_ofdmflexframegen *fg;
    ll_cf **fgbuffer;
    unsigned fgbuffer_len, fgbuffer_index;
    ll_nco nco;
};

ll_mctx ll_mctx_create(unsigned N, unsigned M, unsigned cp, unsigned taper, const unsigned char *p)
{
    if (N < 1 || M < 8 || cp < 1 || taper > cp) return NULL;
    ll_mctx q = (ll_mctx)calloc(1, sizeof(*q));
    q->N = N; q->M = M; q->cp = cp; q->taper = taper;
    ll_ofdmflexframegenprops props = { LL_CRC_32, LL_FEC_NONE, LL_FEC_HAMMING128, LL_MODEM_QPSK };
    q->fg = (ll_ofdmflexframegen *)calloc(N, sizeof(ll_ofdmflexframegen));
    q->fgbuffer = (ll_cf **)calloc(N, sizeof(ll_cf *));
    q->fgbuffer_len = M + cp;
    for (unsigned i = 0; i < N; i++) {
        q->fg[i] = ll_ofdmflexframegen_create(M, cp, taper, p, &props);
        q->fgbuffer[i] = (ll_cf *)calloc(q->fgbuffer_len, sizeof(ll_cf));
    }
    q->ch = ll_firpfbch_create_kaiser(LL_SYNTHESIZER, 2 * N, 13, 60.0f);
    q->X = (ll_cf *)calloc(2 * N, sizeof(ll_cf));
    q->x = (ll_cf *)calloc(2 * N, sizeof(ll_cf));
    ll_nco_reset(&q->nco);
    ll_nco_set_frequency(&q->nco, mc_offset(N));
    ll_mctx_reset(q);
    return q;
}
void ll_mctx_destroy(ll_mctx q)
{
    if (!q) return;
    for (unsigned i = 0; i < q->N; i++) { ll_ofdmflexframegen_destroy(q->fg[i]); free(q->fgbuffer[i]); }
    ll_firpfbch_destroy(q->ch);
    free(q->fg); free(q->fgbuffer); free(q->X); free(q->x); free(q);
}
void ll_mctx_reset(ll_mctx q)
{
    for (unsigned i = 0; i < q->N; i++) {
        ll_ofdmflexframegen_reset(q->fg[i]);
        memset(q->fgbuffer[i], 0, sizeof(ll_cf) * q->fgbuffer_len);
    }
    ll_firpfbch_reset(q->ch);
    memset(q->X, 0, sizeof(ll_cf) * 2 * q->N);
    memset(q->x, 0, sizeof(ll_cf) * 2 * q->N);
    q->fgbuffer_index = q->fgbuffer_len;
}
int ll_mctx_is_channel_ready(ll_mctx q, unsigned ch)
{ if (ch >= q->N) return -1; return ll_ofdmflexframegen_is_assembled(q->fg[ch]) ? 0 : 1; }

int ll_mctx_update_data(ll_mctx q, unsigned ch, const unsigned char *header,
                        const unsigned char *payload, unsigned payload_len, int mod, int fec0, int fec1)
{
    if (ch >= q->N) return -1;
    if (!ll_mctx_is_channel_ready(q, ch)) return 1;     /* reference: warning + return */
    ll_ofdmflexframegenprops props = { LL_CRC_32, (unsigned)fec0, (unsigned)fec1, (unsigned)mod };
    ll_ofdmflexframegen_setprops(q->fg[ch], &props);
    ll_ofdmflexframegen_assemble(q->fg[ch], header, payload, payload_len);
    return 0;
}
void ll_mctx_generate_samples(ll_mctx q, ll_cf *buf)
{
    unsigned K = 2 * q->N;
    if (q->fgbuffer_index >= q->fgbuffer_len) {
        for (unsigned i = 0; i < q->N; i++) {
            if (ll_ofdmflexframegen_is_assembled(q->fg[i])) ll_ofdmflexframegen_writesymbol(q->fg[i], q->fgbuffer[i]);
            else memset(q->fgbuffer[i], 0, sizeof(ll_cf) * q->fgbuffer_len);
        }
        q->fgbuffer_index = 0;
    }
    for (unsigned i = 0; i < q->N; i++) q->X[i] = q->fgbuffer[i][q->fgbuffer_index];
    ll_firpfbch_synthesizer_execute(q->ch, q->X, buf);
    for (unsigned i = 0; i < K; i++) {
        buf[i] = ll_nco_mix_up(&q->nco, buf[i]);
        ll_nco_step(&q->nco);
    }
    q->fgbuffer_index++;
}

/* ================================================================== msresamp */
#define RS_PHASE_BITS 24
typedef struct ll_resamp2_s { unsigned m; float *h1; ll_cf *w0, *w1; } ll_resamp2;   /* half-band decimator */

struct ll_msresamp_s {
    float rate, As;
    int interp;                  /* rate > 1: arbitrary stage first, then half-band interpolators */
    unsigned num_stages;         /* half-band stages: decimators (rate < 0.5) or interpolators (rate > 2) */
    ll_resamp2 *hb;
    ll_cf *hb_buf; unsigned hb_count;
    /* arbitrary stage */
    double rate_arb;
    unsigned npfb, m, nbits;
    float *hpfb;                 /* [npfb][2m] reversed per branch */
    ll_cf *win;                  /* 2m, oldest first */
    uint32_t phase, step;
};

static void resamp2_init(ll_resamp2 *r, unsigned m, float As)
{
    unsigned h_len = 4 * m + 1;
    float *h = (float *)malloc(sizeof(float) * h_len);
    /* fc = 0.25 makes the shared Kaiser routine's sinc(2 fc t) equal sinc(t/2) */
    ll_firdes_kaiser(h_len, 0.25f, As, 0.0f, h);
    r->m = m;
    r->h1 = (float *)malloc(sizeof(float) * 2 * m);
    unsigned j = 0;
    for (unsigned i = 1; i < h_len; i += 2) r->h1[j++] = h[h_len - i - 1];
    r->w0 = (ll_cf *)calloc(2 * m, sizeof(ll_cf));
    r->w1 = (ll_cf *)calloc(2 * m, sizeof(ll_cf));
    free(h);
}
static ll_cf resamp2_decim(ll_resamp2 *r, ll_cf x0, ll_cf x1)
{
    unsigned n = 2 * r->m;
    memmove(r->w1, r->w1 + 1, sizeof(ll_cf) * (n - 1)); r->w1[n - 1] = x0;
    ll_cf y1 = { 0, 0 };
    for (unsigned i = 0; i < n; i++) { y1.re += r->h1[i] * r->w1[i].re; y1.im += r->h1[i] * r->w1[i].im; }
    memmove(r->w0, r->w0 + 1, sizeof(ll_cf) * (n - 1)); r->w0[n - 1] = x1;
    ll_cf y0 = r->w0[r->m - 1];
    ll_cf y = { 0.5f * (y0.re + y1.re), 0.5f * (y0.im + y1.im) };
    return y;
}

/* half-band interpolator (liquid resamp2_crcf_interp_execute): delay branch first, filter branch second */
static void resamp2_interp(ll_resamp2 *r, ll_cf x, ll_cf *y)
{
    unsigned n = 2 * r->m;
    memmove(r->w0, r->w0 + 1, sizeof(ll_cf) * (n - 1)); r->w0[n - 1] = x;
    y[0] = r->w0[r->m - 1];
    memmove(r->w1, r->w1 + 1, sizeof(ll_cf) * (n - 1)); r->w1[n - 1] = x;
    ll_cf y1 = { 0, 0 };
    for (unsigned i = 0; i < n; i++) { y1.re += r->h1[i] * r->w1[i].re; y1.im += r->h1[i] * r->w1[i].im; }
    y[1] = y1;
}

ll_msresamp ll_msresamp_create(float rate, float As)
{
    if (!(rate > 0.0f) || rate > 1024.0f) return NULL;
    ll_msresamp q = (ll_msresamp)calloc(1, sizeof(*q));
    q->rate = rate; q->As = As;
    q->rate_arb = (double)rate;
    q->interp = rate > 1.0f;
    /* liquid msresamp_crcf_create: the arbitrary stage works in [0.5, 1] (decimating) or (1, 2] (interpolating,
     * as the transmit applications use it: src/flexframe_tx.cc:170 msresamp_crcf_create(2.0, 60)) */
    if (q->interp) while (q->rate_arb > 2.0) { q->num_stages++; q->rate_arb *= 0.5; }
    else           while (q->rate_arb < 0.5) { q->num_stages++; q->rate_arb *= 2.0; }
    q->hb = (ll_resamp2 *)calloc(q->num_stages ? q->num_stages : 1, sizeof(ll_resamp2));
    for (unsigned i = 0; i < q->num_stages; i++) resamp2_init(&q->hb[i], 7, As);
    q->hb_buf = (ll_cf *)calloc(2u << q->num_stages, sizeof(ll_cf));
    q->npfb = 256; q->nbits = 8; q->m = 7;
    float fc = 0.515f * (float)q->rate_arb; if (fc > 0.49f) fc = 0.49f;
    unsigned n = 2 * q->m * q->npfb + 1;
    float *hf = (float *)malloc(sizeof(float) * n);
    ll_firdes_kaiser(n, fc / (float)q->npfb, As, 0.0f, hf);
    double gain = 0; for (unsigned i = 0; i < n; i++) gain += hf[i];
    gain = (double)q->npfb / gain;
    unsigned hs = 2 * q->m;
    q->hpfb = (float *)malloc(sizeof(float) * q->npfb * hs);
    for (unsigned b = 0; b < q->npfb; b++)
        for (unsigned k = 0; k < hs; k++)
            q->hpfb[b * hs + (hs - 1 - k)] = (float)((double)hf[b + k * q->npfb] * gain);
    free(hf);
    q->win = (ll_cf *)calloc(hs, sizeof(ll_cf));
    q->step = (uint32_t)llrint((double)(1u << RS_PHASE_BITS) / q->rate_arb);
    ll_msresamp_reset(q);
    return q;
}
void ll_msresamp_destroy(ll_msresamp q)
{
    if (!q) return;
    for (unsigned i = 0; i < q->num_stages; i++) { free(q->hb[i].h1); free(q->hb[i].w0); free(q->hb[i].w1); }
    free(q->hb); free(q->hb_buf); free(q->hpfb); free(q->win); free(q);
}
void ll_msresamp_reset(ll_msresamp q)
{
    for (unsigned i = 0; i < q->num_stages; i++) {
        memset(q->hb[i].w0, 0, sizeof(ll_cf) * 2 * q->hb[i].m);
        memset(q->hb[i].w1, 0, sizeof(ll_cf) * 2 * q->hb[i].m);
    }
    memset(q->win, 0, sizeof(ll_cf) * 2 * q->m);
    q->hb_count = 0; q->phase = 0;
}
float ll_msresamp_get_delay(ll_msresamp q)
{
    float d = (float)q->m;                       /* arbitrary stage, in its input samples */
    for (unsigned i = 0; i < q->num_stages; i++) d = 2.0f * d + (float)(2 * q->hb[i].m - 1);
    return d;
}
static unsigned resamp_arb(ll_msresamp q, ll_cf x, ll_cf *y)
{
    unsigned hs = 2 * q->m, n = 0;
    memmove(q->win, q->win + 1, sizeof(ll_cf) * (hs - 1)); q->win[hs - 1] = x;
    while (q->phase < (1u << RS_PHASE_BITS)) {
        unsigned b = q->phase >> (RS_PHASE_BITS - q->nbits);
        const float *h = q->hpfb + b * hs;
        ll_cf acc = { 0, 0 };
        for (unsigned k = 0; k < hs; k++) { acc.re += h[k] * q->win[k].re; acc.im += h[k] * q->win[k].im; }
        y[n++] = acc;
        q->phase += q->step;
    }
    q->phase -= (1u << RS_PHASE_BITS);
    return n;
}
void ll_msresamp_execute(ll_msresamp q, const ll_cf *x, unsigned nx, ll_cf *y, unsigned *ny)
{
    unsigned n = 0, D = 1u << q->num_stages;
    if (q->interp) {
        /* msresamp_crcf_interp_execute: arbitrary stage, then every sample through the half-band interpolators */
        ll_cf arb[4];
        ll_cf *a = q->hb_buf, *b = q->hb_buf + D;
        for (unsigned i = 0; i < nx; i++) {
            unsigned na = resamp_arb(q, x[i], arb);
            for (unsigned j = 0; j < na; j++) {
                a[0] = arb[j];
                unsigned cnt = 1;
                for (unsigned s = 0; s < q->num_stages; s++) {
                    for (unsigned k = 0; k < cnt; k++) resamp2_interp(&q->hb[s], a[k], b + 2 * k);
                    cnt *= 2;
                    ll_cf *t = a; a = b; b = t;
                }
                for (unsigned k = 0; k < cnt; k++) y[n++] = a[k];
            }
        }
        *ny = n;
        return;
    }
    for (unsigned i = 0; i < nx; i++) {
        ll_cf v = x[i];
        if (q->num_stages) {
            q->hb_buf[q->hb_count++] = v;
            if (q->hb_count < D) continue;
            q->hb_count = 0;
            unsigned cnt = D;
            for (unsigned s = 0; s < q->num_stages; s++) {
                for (unsigned k = 0; k < cnt / 2; k++)
                    q->hb_buf[k] = resamp2_decim(&q->hb[s], q->hb_buf[2 * k], q->hb_buf[2 * k + 1]);
                cnt /= 2;
            }
            v = q->hb_buf[0];
        }
        n += resamp_arb(q, v, y + n);
    }
    *ny = n;
}

/* ================================================================== multichannelrx, oversampled front end
 * The channelizer BASELINE.json's north_star names: liquid's firpfbch2_crcf (2N channels, every N input samples one
 * sample on each channel = twice the channel rate, prototype cut off at the neighbouring channel's centre) in place of
 * the critically sampled firpfbch of lib/multichannelrx.cc:89-91, followed per kept channel by a half-band decimator
 * (liquid resamp2_crcf, m = 7, 60 dB) that brings the stream back to the rate the frame synchronizers expect.  Same
 * oscillator, same channel <-> bin assignment, same synchronizers; not the reference's receiver (different filters),
 * so it is compared with the GPU build of the same chain, and with what the transmitter sent. */
void ll_mcrx_set_front_end(ll_mcrx q, int oversampled)
{
    q->front_end = oversampled ? 1 : 0;
    if (q->front_end && !q->ch2) {
        q->ch2 = ll_firpfbch2_create_kaiser(2 * q->N, 7, 60.0f);
        q->hb = (ll_resamp2 *)calloc(q->N, sizeof(ll_resamp2));
        for (unsigned c = 0; c < q->N; c++) resamp2_init(&q->hb[c], 7, 60.0f);
        q->Y2 = (ll_cf *)calloc(2 * q->N, sizeof(ll_cf));
        q->pair = (ll_cf *)calloc(q->N, sizeof(ll_cf));
    }
    q->step = 0; q->buffer_index = 0;
}
static void mcrx_reset_oversampled(ll_mcrx q)
{
    if (!q->ch2) return;
    ll_firpfbch2_reset(q->ch2);
    for (unsigned c = 0; c < q->N; c++) {
        memset(q->hb[c].w0, 0, sizeof(ll_cf) * 2 * q->hb[c].m);
        memset(q->hb[c].w1, 0, sizeof(ll_cf) * 2 * q->hb[c].m);
    }
    q->step = 0;
}
static void mcrx_free_oversampled(ll_mcrx q)
{
    if (!q->ch2) return;
    ll_firpfbch2_destroy(q->ch2);
    for (unsigned c = 0; c < q->N; c++) { free(q->hb[c].h1); free(q->hb[c].w0); free(q->hb[c].w1); }
    free(q->hb); free(q->Y2); free(q->pair);
}
static void mcrx_execute_oversampled(ll_mcrx q, const ll_cf *xin, unsigned n)
{
    const unsigned N = q->N;
    for (unsigned i = 0; i < n; i++) {
        q->x[q->buffer_index] = ll_nco_mix_down(&q->nco, xin[i]);
        ll_nco_step(&q->nco);
        if (++q->buffer_index == N) {                       /* one step of the oversampled bank per N samples */
            q->buffer_index = 0;
            ll_firpfbch2_analyzer_execute(q->ch2, q->x, q->Y2);
            if (q->step & 1) {
                for (unsigned c = 0; c < N; c++) {
                    ll_cf y = resamp2_decim(&q->hb[c], q->pair[c], q->Y2[c]);
                    ll_ofdmflexframesync_execute(q->fs[c], &y, 1);
                }
            } else memcpy(q->pair, q->Y2, sizeof(ll_cf) * N);
            q->step++;
        }
    }
}
/* the channel streams after the adapter, [block][N], for stage-level parity */
void ll_mcrx_channelize_oversampled(ll_mcrx q, const ll_cf *xin, unsigned nblocks, ll_cf *out)
{
    const unsigned N = q->N;
    ll_mcrx_set_front_end(q, 1);
    for (unsigned b = 0; b < nblocks; b++)
        for (unsigned half = 0; half < 2; half++) {
            for (unsigned i = 0; i < N; i++) { q->x[i] = ll_nco_mix_down(&q->nco, xin[((size_t)2 * b + half) * N + i]); ll_nco_step(&q->nco); }
            ll_firpfbch2_analyzer_execute(q->ch2, q->x, q->Y2);
            if (half) for (unsigned c = 0; c < N; c++) out[(size_t)b * N + c] = resamp2_decim(&q->hb[c], q->pair[c], q->Y2[c]);
            else memcpy(q->pair, q->Y2, sizeof(ll_cf) * N);
        }
}
