/*
 * ll_firpfbch2.c -- CPU ORACLE (test infrastructure): 2x-oversampled polyphase analysis bank.
 *
 * Restates liquid-dsp src/multichannel/src/firpfbch2.c (firpfbch2_crcf_create_kaiser, _create,
 * _execute_analyzer, _reset) from its published algorithm.  liquid-usrp itself never calls it (its
 * multichannelrx uses the critically sampled firpfbch with 2N channels, lib/multichannelrx.cc:89-91);
 * it is the channelizer BASELINE.json's north_star names, kept here as the alternate front end of
 * SURVEY.md section 8(f) item 2.  PARITY UNPINNED like the rest of the oracle; pinned only by the
 * float64 direct-form model in tests/test_oracle_dsp.py.
 *
 * M channels (even), M/2 new samples per call, M outputs per call (each channel at 2 fs / M).
 * Prototype: h_len = 2*M*m + 1, Kaiser, fc = 1/M for the analyzer, scaled so the taps sum to M; the
 * first 2*M*m taps are used.  Branch i holds h_sub[2m-1-n] = h[i + n*M] against a newest-last window.
 * A call pushes its M/2 samples into windows base-1, base-2, ... with base = M/2 on even calls and M on odd
 * calls; branch i's dot product is taken on window i (even calls) or (i + M/2) mod M (odd calls) and lands in
 * that same slot of the transform input; backward FFT; result scaled by 1/M.
 */
#include "liquidlite.h"
#include <stdlib.h>
#include <string.h>

struct ll_firpfbch2_s {
    unsigned M, M2, p;
    float *h;          /* p*M taps */
    float *hsub;       /* [M][p] reversed per branch */
    ll_cf *win;        /* [M][p] oldest first */
    ll_cf *X, *x;
    int flag;
};

ll_firpfbch2 ll_firpfbch2_create_kaiser(unsigned M, unsigned m, float As)
{
    if (M < 2 || (M & 1) || m < 1) return NULL;
    unsigned h_len = 2 * M * m + 1;
    float *h = (float *)malloc(sizeof(float) * h_len);
    ll_firdes_kaiser(h_len, 1.0f / (float)M, As, 0.0f, h);
    float sum = 0.0f;
    for (unsigned i = 0; i < h_len; i++) sum += h[i];
    for (unsigned i = 0; i < h_len; i++) h[i] = h[i] * (float)M / sum;

    ll_firpfbch2 q = (ll_firpfbch2)calloc(1, sizeof(*q));
    q->M = M; q->M2 = M / 2; q->p = 2 * m;
    q->h = (float *)malloc(sizeof(float) * q->p * M);
    memcpy(q->h, h, sizeof(float) * q->p * M);
    free(h);
    q->hsub = (float *)malloc(sizeof(float) * q->p * M);
    for (unsigned i = 0; i < M; i++)
        for (unsigned n = 0; n < q->p; n++)
            q->hsub[i * q->p + (q->p - 1 - n)] = q->h[i + n * M];
    q->win = (ll_cf *)malloc(sizeof(ll_cf) * q->p * M);
    q->X = (ll_cf *)malloc(sizeof(ll_cf) * M);
    q->x = (ll_cf *)malloc(sizeof(ll_cf) * M);
    ll_firpfbch2_reset(q);
    return q;
}

void ll_firpfbch2_destroy(ll_firpfbch2 q)
{ if (!q) return; free(q->h); free(q->hsub); free(q->win); free(q->X); free(q->x); free(q); }

void ll_firpfbch2_reset(ll_firpfbch2 q)
{
    memset(q->win, 0, sizeof(ll_cf) * q->p * q->M);
    q->flag = 0;
}

unsigned ll_firpfbch2_get_taps(ll_firpfbch2 q, float *h)
{ if (h) memcpy(h, q->h, sizeof(float) * q->p * q->M); return q->p * q->M; }

void ll_firpfbch2_analyzer_execute(ll_firpfbch2 q, const ll_cf *x, ll_cf *y)
{
    unsigned M = q->M, M2 = q->M2, p = q->p;
    unsigned base = q->flag ? M : M2;
    for (unsigned i = 0; i < M2; i++) {                 /* push: window shifts left, newest last */
        ll_cf *w = q->win + (size_t)(base - i - 1) * p;
        memmove(w, w + 1, sizeof(ll_cf) * (p - 1));
        w[p - 1] = x[i];
    }
    for (unsigned i = 0; i < M; i++) {
        unsigned off = q->flag ? (i + M2) % M : i;
        const ll_cf *w = q->win + (size_t)off * p;
        const float *hs = q->hsub + (size_t)i * p;
        float re = 0.0f, im = 0.0f;
        for (unsigned n = 0; n < p; n++) { re += hs[n] * w[n].re; im += hs[n] * w[n].im; }
        q->X[off].re = re; q->X[off].im = im;
    }
    ll_fft(M, q->X, q->x, 1);
    float g = 1.0f / (float)M;
    for (unsigned i = 0; i < M; i++) { y[i].re = q->x[i].re * g; y[i].im = q->x[i].im * g; }
    q->flag = 1 - q->flag;
}

/* convenience for the tests: nsteps consecutive calls */
void ll_firpfbch2_analyze(ll_firpfbch2 q, const ll_cf *x, unsigned nsteps, ll_cf *y)
{
    for (unsigned s = 0; s < nsteps; s++)
        ll_firpfbch2_analyzer_execute(q, x + (size_t)s * q->M2, y + (size_t)s * q->M);
}
