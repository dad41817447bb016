/*
 * ll_math.c -- CPU ORACLE (test infrastructure): filter design, FFT, NCO, m-sequence.
 *
 * Restates (from the published liquid-dsp algorithms; liquid-dsp itself is absent,
 * see liquidlite.h "PARITY UNPINNED"):
 *   liquid-dsp src/math/src/math.bessel.c      besseli0f  (power series)
 *   liquid-dsp src/math/src/windows.c          kaiser(): r = 2t/N  (N, not N-1)
 *   liquid-dsp src/filter/src/firdes.c         liquid_firdes_kaiser(), kaiser_beta_As()
 *   liquid-dsp src/fft/src/fft_common.c        unnormalised forward/backward DFT
 *   liquid-dsp src/nco/src/nco.proto.c         32-bit phase accumulator oscillator
 *   liquid-dsp src/sequence/src/msequence.c    Fibonacci LFSR m-sequence
 * Call sites in the reference: /root/reference/lib/multichannelrx.cc:89-100,163-164.
 *
 * Deviation: window/sinc/bessel are evaluated in double and rounded
 * once to float; liquid evaluates them with float libm calls.
 */
#include "liquidlite.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ bessel / kaiser */
static double besseli0_d(double z)
{
    /* I0(z) = sum_k ((z/2)^k / k!)^2 */
    double y = 1.0, t = 1.0;
    for (int k = 1; k < 64; k++) {
        t *= (0.5 * z) / (double)k;
        y += t * t;
        if (t * t < 1e-18 * y) break;
    }
    return y;
}
float ll_besseli0(float z) { return (float)besseli0_d((double)z); }

float ll_kaiser_beta_As(float As)
{
    As = fabsf(As);
    if (As > 50.0f) return 0.1102f * (As - 8.7f);
    if (As > 21.0f) return 0.5842f * powf(As - 21.0f, 0.4f) + 0.07886f * (As - 21.0f);
    return 0.0f;
}

void ll_firdes_kaiser(unsigned n, float fc, float As, float mu, float *h)
{
    double beta = (double)ll_kaiser_beta_As(As);
    double ib = besseli0_d(beta);
    for (unsigned i = 0; i < n; i++) {
        double t = (double)i - (double)(n - 1) / 2.0 + (double)mu;
        double x = 2.0 * (double)fc * t;
        double s = (fabs(x) < 1e-9) ? 1.0 : sin(M_PI * x) / (M_PI * x);
        double r = 2.0 * t / (double)n;
        double a = 1.0 - r * r;
        double w = besseli0_d(beta * sqrt(a < 0 ? 0 : a)) / ib;
        h[i] = (float)(s * w);
    }
}

/* ------------------------------------------------------------------ FFT (any size) */
typedef struct fftplan_s {
    unsigned n;
    ll_cf *w;          /* w[k] = exp(-j 2 pi k / n) */
    unsigned *rev;     /* bit reversal (power of two only) */
    struct fftplan_s *next;
} fftplan;
static fftplan *g_plans = NULL;

static fftplan *plan_get(unsigned n)
{
    for (fftplan *p = g_plans; p; p = p->next) if (p->n == n) return p;
    fftplan *p = (fftplan *)calloc(1, sizeof(fftplan));
    p->n = n;
    p->w = (ll_cf *)malloc(sizeof(ll_cf) * n);
    for (unsigned k = 0; k < n; k++) {
        double a = -2.0 * M_PI * (double)k / (double)n;
        p->w[k].re = (float)cos(a);
        p->w[k].im = (float)sin(a);
    }
    if ((n & (n - 1)) == 0) {
        unsigned lg = 0; while ((1u << lg) < n) lg++;
        p->rev = (unsigned *)malloc(sizeof(unsigned) * n);
        for (unsigned i = 0; i < n; i++) {
            unsigned r = 0;
            for (unsigned b = 0; b < lg; b++) if (i & (1u << b)) r |= 1u << (lg - 1 - b);
            p->rev[i] = r;
        }
    }
    p->next = g_plans; g_plans = p;
    return p;
}

static inline ll_cf cmul(ll_cf a, ll_cf b)
{ ll_cf r = { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re }; return r; }
static inline ll_cf cmulc(ll_cf a, ll_cf b) /* a * conj(b) */
{ ll_cf r = { a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im }; return r; }

/* twiddle with direction: w^k (forward) or conj(w^k) (backward) */
static inline ll_cf tw(const fftplan *p, unsigned k, int backward)
{ ll_cf w = p->w[k]; if (backward) w.im = -w.im; return w; }

static void fft_rec(const fftplan *p, unsigned n, unsigned stride, const ll_cf *x, ll_cf *y,
                    ll_cf *scratch, int backward)
{
    /* mixed-radix decimation in time on x[0], x[stride], ... (n points) */
    if (n == 1) { y[0] = x[0]; return; }
    unsigned f = 2;
    while (n % f) f++;
    unsigned m = n / f;
    unsigned tws = p->n / n;            /* twiddle stride in the root table */
    for (unsigned r = 0; r < f; r++)
        fft_rec(p, m, stride * f, x + r * stride, scratch + r * m, y, backward);
    /* note: the recursive calls used y as their scratch; results are in scratch[r*m + k] */
    for (unsigned k = 0; k < m; k++) {
        for (unsigned q = 0; q < f; q++) {
            unsigned kk = k + q * m;
            ll_cf acc = scratch[k];
            for (unsigned r = 1; r < f; r++) {
                unsigned idx = (unsigned)(((uint64_t)r * kk) % n) * tws;
                ll_cf t = cmul(scratch[r * m + k], tw(p, idx, backward));
                acc.re += t.re; acc.im += t.im;
            }
            y[kk] = acc;
        }
    }
}

void ll_fft(unsigned n, const ll_cf *x, ll_cf *y, int backward)
{
    const fftplan *p = plan_get(n);
    if (p->rev) {
        /* iterative radix-2 decimation in time */
        if (x != y) { for (unsigned i = 0; i < n; i++) y[p->rev[i]] = x[i]; }
        else {
            for (unsigned i = 0; i < n; i++) {
                unsigned r = p->rev[i];
                if (r > i) { ll_cf t = y[i]; y[i] = y[r]; y[r] = t; }
            }
        }
        for (unsigned len = 2; len <= n; len <<= 1) {
            unsigned half = len >> 1, step = n / len;
            for (unsigned i = 0; i < n; i += len) {
                for (unsigned k = 0; k < half; k++) {
                    ll_cf w = tw(p, k * step, backward);
                    ll_cf a = y[i + k];
                    ll_cf b = cmul(y[i + k + half], w);
                    y[i + k].re = a.re + b.re;        y[i + k].im = a.im + b.im;
                    y[i + k + half].re = a.re - b.re; y[i + k + half].im = a.im - b.im;
                }
            }
        }
        return;
    }
    /* generic size: recursive mixed radix, O(n * sum of prime factors) */
    ll_cf *tmp = (ll_cf *)malloc(sizeof(ll_cf) * n * 2);
    ll_cf *in = tmp, *sc = tmp + n;
    memcpy(in, x, sizeof(ll_cf) * n);
    /* fft_rec needs (y, scratch) distinct buffers of n points; the recursion alternates them */
    ll_cf *out = (ll_cf *)malloc(sizeof(ll_cf) * n);
    fft_rec(p, n, 1, in, out, sc, backward);
    memcpy(y, out, sizeof(ll_cf) * n);
    free(out); free(tmp);
}

/* ------------------------------------------------------------------ NCO (32-bit phase) */
#define LL_TWO32 4294967296.0

uint32_t ll_nco_rad2u32(float rad)
{
    double p = (double)rad * (1.0 / (2.0 * M_PI));
    p -= floor(p);                                  /* [0,1) */
    uint64_t v = (uint64_t)llrint(p * LL_TWO32);    /* may equal 2^32 -> wraps to 0 */
    return (uint32_t)v;
}
float ll_nco_u32rad(uint32_t u)
{ return (float)((double)(int32_t)u * (2.0 * M_PI / LL_TWO32)); }

void ll_nco_reset(ll_nco *q) { q->theta = 0; q->d_theta = 0; }
void ll_nco_set_frequency(ll_nco *q, float dtheta) { q->d_theta = ll_nco_rad2u32(dtheta); }
void ll_nco_adjust_frequency(ll_nco *q, float df) { q->d_theta += ll_nco_rad2u32(df); }
float ll_nco_get_frequency(const ll_nco *q) { return ll_nco_u32rad(q->d_theta); }
void ll_nco_step(ll_nco *q) { q->theta += q->d_theta; }

void ll_nco_sincos_u32(uint32_t theta, float *s, float *c)
{
    double a = (double)theta * (2.0 * M_PI / LL_TWO32);
    *s = (float)sin(a);
    *c = (float)cos(a);
}
ll_cf ll_nco_mix_down(const ll_nco *q, ll_cf x)
{
    float s, c; ll_nco_sincos_u32(q->theta, &s, &c);
    ll_cf y = { x.re * c + x.im * s, x.im * c - x.re * s };      /* x * conj(e^{j theta}) */
    return y;
}
ll_cf ll_nco_mix_up(const ll_nco *q, ll_cf x)
{
    float s, c; ll_nco_sincos_u32(q->theta, &s, &c);
    ll_cf y = { x.re * c - x.im * s, x.im * c + x.re * s };
    return y;
}

/* ------------------------------------------------------------------ m-sequence */
static const unsigned ll_mseq_genpoly[16] = {
    0, 0, 0x0007, 0x000B, 0x0013, 0x0025, 0x0043, 0x0089,
    0x011D, 0x0211, 0x0409, 0x0805, 0x1053, 0x201b, 0x402b, 0x8003 };

void ll_msequence_init_default(ll_msequence *ms, unsigned m)
{
    ms->m = m;
    ms->g = ll_mseq_genpoly[m] >> 1;     /* generator polynomial with the top bit clipped */
    ms->a = 1u << (m - 1);               /* initial state 0..01, bit-reversed */
    ms->n = (1u << m) - 1;
    ms->v = ms->a;
    ms->b = 0;
}
void ll_msequence_reset(ll_msequence *ms) { ms->v = ms->a; ms->b = 0; }
unsigned ll_msequence_advance(ll_msequence *ms)
{
    ms->b = (unsigned)__builtin_parity(ms->v & ms->g);
    ms->v = ((ms->v << 1) | ms->b) & ms->n;
    return ms->b;
}
unsigned ll_msequence_generate_symbol(ll_msequence *ms, unsigned bps)
{
    unsigned s = 0;
    for (unsigned i = 0; i < bps; i++) s = (s << 1) | ll_msequence_advance(ms);
    return s;
}
