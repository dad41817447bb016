/*
 * ll_modem.c -- CPU ORACLE (test infrastructure): linear modems with hard and
 * soft-decision demodulation.
 *
 * Restates liquid-dsp src/modem/src/modem_bpsk.c, modem_qpsk.c, modem_qam.c and the
 * soft-demod neighbour table of modem_common.c (modem_demodsoft_gentab,
 * modem_demodulate_soft_table).  Used by the frame generator / synchronizer the
 * reference builds at /root/reference/lib/multichanneltx.cc:70-81 and
 * lib/multichannelrx.cc:82 (modulation chosen per frame, lib/multichanneltx.cc:184).
 *
 *  BPSK : bit 1 -> -1.         soft: LLR = -2*re*4.0,  soft = (int)(LLR*16+127) clamped to 0..255
 *  QPSK : bit0 -> sign(re), bit1 -> sign(im), +-1/sqrt(2).  gamma = 5.8; soft[0] <- im, soft[1] <- re
 *  QAM  : square, per-axis Gray, levels (2s-L+1)*alpha, symbol = (gray(i) << m_q) | gray(q);
 *         soft: min distances over {hard symbol + p nearest neighbours}, gamma = 1.2*M.
 *
 * Deviation: the neighbour table is built from exact integer grid
 * distances with ties broken towards the lower symbol index (liquid breaks float ties
 * the same way but through cabsf rounding).
 */
#include "liquidlite.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct ll_modem_s {
    int scheme;
    unsigned bps, M;
    unsigned m_i, m_q;          /* QAM */
    float alpha;
    ll_cf x_hat, r;             /* last re-modulated / received */
    unsigned soft_p;
    unsigned char *soft_nb;     /* [M][soft_p] */
};

static unsigned gray_encode(unsigned x) { return x ^ (x >> 1); }
static unsigned gray_decode(unsigned x)
{ unsigned y = x; while (x >>= 1) y ^= x; return y; }

static void qam_grid(ll_modem q, unsigned sym, int *gi, int *gq)
{
    unsigned s_i = gray_decode(sym >> q->m_q);
    unsigned s_q = gray_decode(sym & ((1u << q->m_q) - 1));
    *gi = 2 * (int)s_i - (int)(1u << q->m_i) + 1;
    *gq = 2 * (int)s_q - (int)(1u << q->m_q) + 1;
}

ll_modem ll_modem_create(int scheme)
{
    ll_modem q = (ll_modem)calloc(1, sizeof(*q));
    q->scheme = scheme;
    switch (scheme) {
    case LL_MODEM_BPSK: q->bps = 1; break;
    case LL_MODEM_QPSK: q->bps = 2; break;
    case LL_MODEM_QAM16: q->bps = 4; q->alpha = (float)(1.0 / sqrt(10.0)); break;
    case LL_MODEM_QAM64: q->bps = 6; q->alpha = (float)(1.0 / sqrt(42.0)); break;
    default: free(q); return NULL;
    }
    q->M = 1u << q->bps;
    if (scheme == LL_MODEM_QAM16 || scheme == LL_MODEM_QAM64) {
        q->m_i = q->m_q = q->bps / 2;
        q->soft_p = 4;
        q->soft_nb = (unsigned char *)malloc(q->M * q->soft_p);
        for (unsigned i = 0; i < q->M; i++) {
            int ai, aq; qam_grid(q, i, &ai, &aq);
            for (unsigned k = 0; k < q->soft_p; k++) {
                long dmin = 1L << 40; unsigned best = q->M;
                for (unsigned j = 0; j < q->M; j++) {
                    int ok = (j != i);
                    for (unsigned l = 0; l < k; l++) if (q->soft_nb[i * q->soft_p + l] == j) ok = 0;
                    if (!ok) continue;
                    int bi, bq; qam_grid(q, j, &bi, &bq);
                    long d = (long)(ai - bi) * (ai - bi) + (long)(aq - bq) * (aq - bq);
                    if (d < dmin) { dmin = d; best = j; }
                }
                q->soft_nb[i * q->soft_p + k] = (unsigned char)best;
            }
        }
    }
    return q;
}
void ll_modem_destroy(ll_modem q) { if (!q) return; free(q->soft_nb); free(q); }
unsigned ll_modem_bps(ll_modem q) { return q->bps; }
const unsigned char *ll_modem_soft_neighbors(ll_modem q, unsigned *p) { if (p) *p = q->soft_p; return q->soft_nb; }

ll_cf ll_modem_modulate(ll_modem q, unsigned sym)
{
    ll_cf y = { 0, 0 };
    switch (q->scheme) {
    case LL_MODEM_BPSK: y.re = sym ? -1.0f : 1.0f; break;
    case LL_MODEM_QPSK: {
        const float a = (float)M_SQRT1_2;
        y.re = (sym & 1) ? -a : a;
        y.im = (sym & 2) ? -a : a;
    } break;
    default: {
        int gi, gq; qam_grid(q, sym, &gi, &gq);
        y.re = (float)gi * q->alpha;
        y.im = (float)gq * q->alpha;
    }
    }
    return y;
}

/* successive-approximation slicer against reference levels 2^k * alpha
 * (liquid modem_demodulate_linear_array_ref) */
static unsigned slice_axis(float v, unsigned b, float alpha)
{
    unsigned s = 0;
    for (unsigned k = b; k > 0; k--) {
        float ref = (float)(1u << (k - 1)) * alpha;
        s <<= 1;
        if (v > 0) { s |= 1; v -= ref; } else { v += ref; }
    }
    return s;
}

unsigned ll_modem_demodulate(ll_modem q, ll_cf r)
{
    unsigned s;
    switch (q->scheme) {
    case LL_MODEM_BPSK: s = (r.re > 0) ? 0 : 1; break;
    case LL_MODEM_QPSK: s = ((r.re > 0) ? 0 : 1) + ((r.im > 0) ? 0 : 2); break;
    default: {
        unsigned s_i = slice_axis(r.re, q->m_i, q->alpha);
        unsigned s_q = slice_axis(r.im, q->m_q, q->alpha);
        s = (gray_encode(s_i) << q->m_q) + gray_encode(s_q);
    }
    }
    q->r = r;
    q->x_hat = ll_modem_modulate(q, s);
    return s;
}

float ll_modem_get_evm(ll_modem q)
{
    float dr = q->x_hat.re - q->r.re, di = q->x_hat.im - q->r.im;
    return sqrtf(dr * dr + di * di);
}

static unsigned char soft_clamp(float v)
{
    int sb = (int)v;
    if (sb > 255) sb = 255;
    if (sb < 0) sb = 0;
    return (unsigned char)sb;
}

unsigned ll_modem_demodulate_soft(ll_modem q, ll_cf r, unsigned char *soft)
{
    unsigned s = ll_modem_demodulate(q, r);
    switch (q->scheme) {
    case LL_MODEM_BPSK: {
        float LLR = -2.0f * r.re * 4.0f;
        soft[0] = soft_clamp(LLR * 16.0f + 127.0f);
    } break;
    case LL_MODEM_QPSK: {
        float LLR = -2.0f * r.im * 5.8f;
        soft[0] = soft_clamp(LLR * 16.0f + 127.0f);
        LLR = -2.0f * r.re * 5.8f;
        soft[1] = soft_clamp(LLR * 16.0f + 127.0f);
    } break;
    default: {
        unsigned bps = q->bps;
        float gamma = 1.2f * (float)q->M;
        float dmin0[8], dmin1[8];
        for (unsigned k = 0; k < bps; k++) { dmin0[k] = 8.0f; dmin1[k] = 8.0f; }
        float dr = r.re - q->x_hat.re, di = r.im - q->x_hat.im;
        float d = dr * dr + di * di;
        for (unsigned k = 0; k < bps; k++) {
            if ((s >> (bps - k - 1)) & 1) dmin1[k] = d; else dmin0[k] = d;
        }
        for (unsigned i = 0; i < q->soft_p; i++) {
            unsigned nb = q->soft_nb[s * q->soft_p + i];
            ll_cf xh = ll_modem_modulate(q, nb);
            dr = r.re - xh.re; di = r.im - xh.im;
            d = dr * dr + di * di;
            for (unsigned k = 0; k < bps; k++) {
                if ((nb >> (bps - k - 1)) & 1) { if (d < dmin1[k]) dmin1[k] = d; }
                else                           { if (d < dmin0[k]) dmin0[k] = d; }
            }
        }
        for (unsigned k = 0; k < bps; k++)
            soft[k] = soft_clamp(((dmin0[k] - dmin1[k]) * gamma) * 16.0f + 127.0f);
    }
    }
    return s;
}
