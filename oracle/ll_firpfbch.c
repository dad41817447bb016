/*
 * ll_firpfbch.c -- CPU ORACLE (test infrastructure): critically-sampled polyphase
 * filterbank channelizer.
 *
 * Restates liquid-dsp src/multichannel/src/firpfbch.c (firpfbch_crcf_create_kaiser,
 * _create, _analyzer_execute, _synthesizer_execute, _reset), as called from
 * /root/reference/lib/multichannelrx.cc:89-91,142,188 (analysis, K=2N, m=7, As=60)
 * and /root/reference/lib/multichanneltx.cc:85-87,213 (synthesis, K=2N, m=13, As=60).
 *
 * Prototype: h_len = 2*K*m+1, fc = 0.5/K, Kaiser; only the first p*K = 2*m*K taps are
 * used.  Branch i holds h_sub[p-1-n] = h[i + n*K] against a newest-last window.
 * Analyzer: the K inputs of a block are pushed into branches K-1, K-2, ..., 0; branch
 * i's dot product lands in FFT input slot K-1-i; forward FFT, no scaling.
 * Synthesizer: backward FFT of the K inputs, then sample i is pushed into branch i and
 * branch i's dot product is output sample i.
 */
#include "liquidlite.h"
#include <stdlib.h>
#include <string.h>

struct ll_firpfbch_s {
    int type;
    unsigned K, p;
    float *h;          /* prototype, p*K taps used */
    float *hsub;       /* [K][p], reversed per branch */
    ll_cf *win;        /* [K][p], oldest first */
    ll_cf *X, *x;
    unsigned filter_index;
};

ll_firpfbch ll_firpfbch_create_kaiser(int type, unsigned K, unsigned m, float As)
{
    unsigned h_len = 2 * K * m + 1;
    float *h = (float *)malloc(sizeof(float) * h_len);
    ll_firdes_kaiser(h_len, 0.5f / (float)K, As, 0.0f, h);

    ll_firpfbch q = (ll_firpfbch)calloc(1, sizeof(*q));
    q->type = type; q->K = K; q->p = 2 * m;
    q->h = (float *)malloc(sizeof(float) * q->p * K);
    memcpy(q->h, h, sizeof(float) * q->p * K);
    free(h);
    q->hsub = (float *)malloc(sizeof(float) * q->p * K);
    for (unsigned i = 0; i < K; i++)
        for (unsigned n = 0; n < q->p; n++)
            q->hsub[i * q->p + (q->p - 1 - n)] = q->h[i + n * K];
    q->win = (ll_cf *)malloc(sizeof(ll_cf) * q->p * K);
    q->X = (ll_cf *)malloc(sizeof(ll_cf) * K);
    q->x = (ll_cf *)malloc(sizeof(ll_cf) * K);
    ll_firpfbch_reset(q);
    return q;
}

void ll_firpfbch_destroy(ll_firpfbch q)
{ if (!q) return; free(q->h); free(q->hsub); free(q->win); free(q->X); free(q->x); free(q); }

void ll_firpfbch_reset(ll_firpfbch q)
{
    memset(q->win, 0, sizeof(ll_cf) * q->p * q->K);
    q->filter_index = q->K - 1;
}

unsigned ll_firpfbch_get_taps(ll_firpfbch q, float *h)
{ if (h) memcpy(h, q->h, sizeof(float) * q->p * q->K); return q->p * q->K; }

static inline void win_push(ll_cf *w, unsigned p, ll_cf v)
{ memmove(w, w + 1, sizeof(ll_cf) * (p - 1)); w[p - 1] = v; }

static inline ll_cf dot(const float *h, const ll_cf *w, unsigned p)
{
    ll_cf r = { 0.0f, 0.0f };
    for (unsigned n = 0; n < p; n++) { r.re += h[n] * w[n].re; r.im += h[n] * w[n].im; }
    return r;
}

void ll_firpfbch_analyzer_execute(ll_firpfbch q, const ll_cf *x, ll_cf *y)
{
    unsigned K = q->K, p = q->p;
    for (unsigned i = 0; i < K; i++) {
        win_push(q->win + q->filter_index * p, p, x[i]);
        q->filter_index = (q->filter_index + K - 1) % K;
    }
    for (unsigned i = 0; i < K; i++)
        q->X[K - 1 - i] = dot(q->hsub + i * p, q->win + i * p, p);
    ll_fft(K, q->X, y, 0);
}

void ll_firpfbch_synthesizer_execute(ll_firpfbch q, const ll_cf *X, ll_cf *y)
{
    unsigned K = q->K, p = q->p;
    ll_fft(K, X, q->x, 1);
    for (unsigned i = 0; i < K; i++) {
        win_push(q->win + i * p, p, q->x[i]);
        y[i] = dot(q->hsub + i * p, q->win + i * p, p);
    }
}

/* test/benchmark helper: make dst continue exactly where src stands (same K, p) */
void ll_firpfbch_copy_state(ll_firpfbch dst, const struct ll_firpfbch_s *src)
{
    memcpy(dst->win, src->win, sizeof(ll_cf) * src->p * src->K);
    dst->filter_index = src->filter_index;
}
