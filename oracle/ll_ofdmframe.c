/*
 * ll_ofdmframe.c -- CPU ORACLE (test infrastructure): OFDM frame structure, symbol
 * generator and the sample-serial frame synchronizer state machine.
 *
 * Restates liquid-dsp src/multichannel/src/{ofdmframe.common.c, ofdmframegen.c,
 * ofdmframesync.c}; the reference reaches them through ofdmflexframe{gen,sync}
 * (/root/reference/lib/multichannelrx.cc:82,194; lib/multichanneltx.cc:78,234-236).
 *
 * Synchronizer, per input sample (ofdmframesync_execute): mix down with nco_rx unless
 * seeking; push into an (M+cp)-sample window; then by state
 *   SEEKPLCP   every M samples: g = M / sum|r|^2 over the newest M samples; S0 gain
 *              estimate G0 = FFT(r) * conj(S0) * sqrt(M_S0)/M on even bins;
 *              s = g * sum_k G0[k+2] conj(G0[k]) / M_S0; detect if |s| > 0.35
 *   S0a, S0b   two more S0 estimates M/2 apart; fine timing from arg(s0+s1);
 *              CFO by the time-domain half-symbol ML estimate
 *   S1         S1 gain estimate; accept when |g| > 0.30 and |arg g| < 0.1 pi, else retry
 *              every M/2 samples (16 tries); equaliser = order-4 polynomial fit of |G|
 *              and unwrapped arg G over the enabled subcarriers
 *   RXSYMBOLS  every M+cp samples: FFT, X *= 1/G, pilot phases -> linear fit (p0,p1),
 *              de-rotate, nco frequency += 1e-3 * (p0 - p0_prev), symbol callback.
 *
 * Deviations from liquid: D2 -- both least-squares fits use projection
 * matrices computed once in double (liquid solves float normal equations per call);
 * D3 -- nco_rx is a 32-bit phase accumulator with exact sin/cos (liquid LIQUID_NCO is a
 * 256/1024-entry table); the phase-unwrap step of the NCO trim uses 2 pi.
 */
#include "liquidlite.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

static int ll_dbg(void)
{ static int d = -1; if (d < 0) d = getenv("LL_DEBUG") ? 1 : 0; return d; }
#define DBG(...) do { if (ll_dbg()) fprintf(stderr, __VA_ARGS__); } while (0)

static inline ll_cf cf(float re, float im) { ll_cf r = { re, im }; return r; }
static inline ll_cf cmul(ll_cf a, ll_cf b)
{ return cf(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re); }
static inline ll_cf cmulc(ll_cf a, ll_cf b)     /* a * conj(b) */
{ return cf(a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im); }
static inline float cabs_(ll_cf a) { return sqrtf(a.re * a.re + a.im * a.im); }
static inline float carg_(ll_cf a) { return atan2f(a.im, a.re); }

static unsigned nextpow2(unsigned x)
{ unsigned n = 0; if (x) x--; while (x) { x >>= 1; n++; } return n; }

/* ================================================================== common */
void ll_ofdmframe_init_default_sctype(unsigned M, unsigned char *p)
{
    unsigned M2 = M / 2;
    unsigned G = M / 10; if (G < 2) G = 2;
    unsigned P = (M > 34) ? 8 : 4, P2 = P / 2;
    for (unsigned i = 0; i < M; i++) p[i] = LL_SCTYPE_NULL;
    for (unsigned i = 1; i < M2 - G; i++) {
        unsigned char t = (((i + P2) % P) == 0) ? LL_SCTYPE_PILOT : LL_SCTYPE_DATA;
        p[i] = t;           /* upper band */
        p[M - i] = t;       /* lower band */
    }
}

int ll_ofdmframe_validate_sctype(const unsigned char *p, unsigned M,
                                 unsigned *M_null, unsigned *M_pilot, unsigned *M_data)
{
    unsigned n0 = 0, n1 = 0, n2 = 0;
    for (unsigned i = 0; i < M; i++) {
        if (p[i] == LL_SCTYPE_NULL) n0++;
        else if (p[i] == LL_SCTYPE_PILOT) n1++;
        else if (p[i] == LL_SCTYPE_DATA) n2++;
        else return -1;
    }
    if (M_null) *M_null = n0;
    if (M_pilot) *M_pilot = n1;
    if (M_data) *M_data = n2;
    if (n1 + n2 == 0 || n2 == 0 || n1 < 2) return -2;
    return 0;
}

static void init_S(const unsigned char *p, unsigned M, ll_cf *S, ll_cf *s, unsigned *M_S, int long_seq)
{
    unsigned m = nextpow2(M);
    if (m < 4) m = 4; else if (m > 8) m = 8;
    if (long_seq) m++;
    ll_msequence ms; ll_msequence_init_default(&ms, m);
    unsigned cnt = 0;
    for (unsigned i = 0; i < M; i++) {
        unsigned b = ll_msequence_generate_symbol(&ms, 3) & 1;
        if (p[i] == LL_SCTYPE_NULL || (!long_seq && (i % 2) != 0)) S[i] = cf(0, 0);
        else { S[i] = cf(b ? 1.0f : -1.0f, 0); cnt++; }
    }
    *M_S = cnt;
    ll_fft(M, S, s, 1);
    float g = 1.0f / sqrtf((float)cnt);
    for (unsigned i = 0; i < M; i++) { s[i].re *= g; s[i].im *= g; }
}
void ll_ofdmframe_init_S0(const unsigned char *p, unsigned M, ll_cf *S0, ll_cf *s0, unsigned *M_S0)
{ init_S(p, M, S0, s0, M_S0, 0); }
void ll_ofdmframe_init_S1(const unsigned char *p, unsigned M, ll_cf *S1, ll_cf *s1, unsigned *M_S1)
{ init_S(p, M, S1, s1, M_S1, 1); }

/* least-squares polynomial fit as a projection, solved in double:
 * coefficient matrix C [k][n] = (X^T X)^-1 X^T for abscissae x[n] */
static void lsq_coeff(const double *x, unsigned n, unsigned k, double *C)
{
    double A[16][32];     /* augmented [k][k + ... ] */
    if (k > 11) k = 11;
    double XtX[11][11];
    for (unsigned a = 0; a < k; a++)
        for (unsigned b = 0; b < k; b++) {
            double s = 0; for (unsigned i = 0; i < n; i++) s += pow(x[i], (double)(a + b));
            XtX[a][b] = s;
        }
    /* invert by Gauss-Jordan with partial pivoting */
    for (unsigned a = 0; a < k; a++)
        for (unsigned b = 0; b < 2 * k; b++)
            A[a][b] = (b < k) ? XtX[a][b] : ((b - k) == a ? 1.0 : 0.0);
    for (unsigned c = 0; c < k; c++) {
        unsigned piv = c;
        for (unsigned r = c + 1; r < k; r++) if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
        if (piv != c) for (unsigned b = 0; b < 2 * k; b++) { double t = A[c][b]; A[c][b] = A[piv][b]; A[piv][b] = t; }
        double d = A[c][c];
        for (unsigned b = 0; b < 2 * k; b++) A[c][b] /= d;
        for (unsigned r = 0; r < k; r++) if (r != c) {
            double f = A[r][c];
            if (f != 0.0) for (unsigned b = 0; b < 2 * k; b++) A[r][b] -= f * A[c][b];
        }
    }
    for (unsigned a = 0; a < k; a++)
        for (unsigned i = 0; i < n; i++) {
            double s = 0;
            for (unsigned b = 0; b < k; b++) s += A[a][k + b] * pow(x[i], (double)b);
            C[a * n + i] = s;
        }
}

/* S[i][n]: smoothed value at subcarrier i = sum_n S[i][n] * y[n], y in fft-shifted
 * enabled-subcarrier order (restates ofdmframesync_estimate_eqgain_poly) */
void ll_ofdmframe_eq_smoother(const unsigned char *p, unsigned M, unsigned order, float *S)
{
    unsigned M2 = M / 2, Nen = 0;
    for (unsigned i = 0; i < M; i++) if (p[i] != LL_SCTYPE_NULL) Nen++;
    if (order > Nen - 1) order = Nen - 1;
    if (order > 10) order = 10;
    unsigned k = order + 1;
    double *x = (double *)calloc(Nen, sizeof(double));
    double *C = (double *)malloc(sizeof(double) * k * Nen);
    unsigned n = 0;
    for (unsigned i = 0; i < M; i++) {
        unsigned kk = (i + M2) % M;
        if (p[kk] != LL_SCTYPE_NULL) {
            double f = (kk > M2) ? (double)kk - (double)M : (double)kk;
            x[n++] = f / (double)M;
        }
    }
    lsq_coeff(x, Nen, k, C);
    for (unsigned i = 0; i < M; i++) {
        double f = ((i > M2) ? (double)i - (double)M : (double)i) / (double)M;
        for (unsigned nn = 0; nn < Nen; nn++) {
            double s = 0;
            if (p[i] != LL_SCTYPE_NULL)
                for (unsigned a = 0; a < k; a++) s += pow(f, (double)a) * C[a * Nen + nn];
            S[i * Nen + nn] = (float)s;
        }
    }
    free(x); free(C);
}

/* P[0][n], P[1][n]: intercept / slope of the pilot phase line (x = signed bin index) */
void ll_ofdmframe_pilot_fit(const unsigned char *p, unsigned M, float *P)
{
    unsigned M2 = M / 2, Np = 0;
    for (unsigned i = 0; i < M; i++) if (p[i] == LL_SCTYPE_PILOT) Np++;
    double *x = (double *)calloc(Np, sizeof(double));
    double *C = (double *)malloc(sizeof(double) * 2 * Np);
    unsigned n = 0;
    for (unsigned i = 0; i < M; i++) {
        unsigned k = (i + M2) % M;
        if (p[k] == LL_SCTYPE_PILOT) x[n++] = (k > M2) ? (double)k - (double)M : (double)k;
    }
    lsq_coeff(x, Np, 2, C);
    for (unsigned i = 0; i < 2 * Np; i++) P[i] = (float)C[i];
    free(x); free(C);
}

/* ================================================================== generator */
struct ll_ofdmframegen_s {
    unsigned M, cp, taper;
    unsigned char *p;
    unsigned M_null, M_pilot, M_data, M_S0, M_S1;
    float g_data;
    ll_cf *X, *x, *S0, *s0, *S1, *s1, *postfix;
    float *tw;
    ll_msequence ms_pilot;
};

ll_ofdmframegen ll_ofdmframegen_create(unsigned M, unsigned cp, unsigned taper, const unsigned char *p)
{
    ll_ofdmframegen q = (ll_ofdmframegen)calloc(1, sizeof(*q));
    q->M = M; q->cp = cp; q->taper = taper;
    q->p = (unsigned char *)malloc(M);
    if (p) memcpy(q->p, p, M); else ll_ofdmframe_init_default_sctype(M, q->p);
    ll_ofdmframe_validate_sctype(q->p, M, &q->M_null, &q->M_pilot, &q->M_data);
    q->X = (ll_cf *)calloc(M, sizeof(ll_cf)); q->x = (ll_cf *)calloc(M, sizeof(ll_cf));
    q->S0 = (ll_cf *)calloc(M, sizeof(ll_cf)); q->s0 = (ll_cf *)calloc(M, sizeof(ll_cf));
    q->S1 = (ll_cf *)calloc(M, sizeof(ll_cf)); q->s1 = (ll_cf *)calloc(M, sizeof(ll_cf));
    ll_ofdmframe_init_S0(q->p, M, q->S0, q->s0, &q->M_S0);
    ll_ofdmframe_init_S1(q->p, M, q->S1, q->s1, &q->M_S1);
    q->tw = (float *)calloc(taper ? taper : 1, sizeof(float));
    q->postfix = (ll_cf *)calloc(taper ? taper : 1, sizeof(ll_cf));
    for (unsigned i = 0; i < taper; i++) {
        double t = ((double)i + 0.5) / (double)taper;
        double g = sin(M_PI_2 * t);
        q->tw[i] = (float)(g * g);
    }
    q->g_data = 1.0f / sqrtf((float)(q->M_pilot + q->M_data));
    ll_msequence_init_default(&q->ms_pilot, 8);
    return q;
}
void ll_ofdmframegen_destroy(ll_ofdmframegen q)
{
    if (!q) return;
    free(q->p); free(q->X); free(q->x); free(q->S0); free(q->s0); free(q->S1); free(q->s1);
    free(q->tw); free(q->postfix); free(q);
}
void ll_ofdmframegen_reset(ll_ofdmframegen q)
{
    ll_msequence_reset(&q->ms_pilot);
    memset(q->postfix, 0, sizeof(ll_cf) * (q->taper ? q->taper : 1));
}

/* q->x (M time samples) -> y (cp+M samples) with cyclic prefix and overlap taper */
static void gensymbol(ll_ofdmframegen q, ll_cf *y)
{
    memcpy(y, q->x + q->M - q->cp, sizeof(ll_cf) * q->cp);
    memcpy(y + q->cp, q->x, sizeof(ll_cf) * q->M);
    for (unsigned i = 0; i < q->taper; i++) {
        float a = q->tw[i], b = q->tw[q->taper - i - 1];
        y[i].re = y[i].re * a + q->postfix[i].re * b;
        y[i].im = y[i].im * a + q->postfix[i].im * b;
    }
    memcpy(q->postfix, q->x, sizeof(ll_cf) * q->taper);
}
void ll_ofdmframegen_write_S0a(ll_ofdmframegen q, ll_cf *y)
{
    for (unsigned i = 0; i < q->M + q->cp; i++) y[i] = q->s0[(i + q->M - 2 * q->cp) % q->M];
    for (unsigned i = 0; i < q->taper; i++) { y[i].re *= q->tw[i]; y[i].im *= q->tw[i]; }
}
void ll_ofdmframegen_write_S0b(ll_ofdmframegen q, ll_cf *y)
{
    for (unsigned i = 0; i < q->M + q->cp; i++) y[i] = q->s0[(i + q->M - q->cp) % q->M];
    memcpy(q->postfix, q->s0, sizeof(ll_cf) * q->taper);
}
void ll_ofdmframegen_write_S1(ll_ofdmframegen q, ll_cf *y)
{
    memcpy(q->x, q->s1, sizeof(ll_cf) * q->M);
    gensymbol(q, y);
}
void ll_ofdmframegen_writesymbol(ll_ofdmframegen q, const ll_cf *X, ll_cf *y)
{
    unsigned M = q->M;
    for (unsigned i = 0; i < M; i++) {
        unsigned k = (i + M / 2) % M;           /* pilots are drawn in fft-shifted order */
        if (q->p[k] == LL_SCTYPE_NULL) q->X[k] = cf(0, 0);
        else if (q->p[k] == LL_SCTYPE_PILOT)
            q->X[k] = cf((ll_msequence_advance(&q->ms_pilot) ? 1.0f : -1.0f) * q->g_data, 0);
        else q->X[k] = cf(X[k].re * q->g_data, X[k].im * q->g_data);
    }
    ll_fft(M, q->X, q->x, 1);
    gensymbol(q, y);
}
void ll_ofdmframegen_writetail(ll_ofdmframegen q, ll_cf *y)
{
    for (unsigned i = 0; i < q->taper; i++) {
        float b = q->tw[q->taper - i - 1];
        y[i] = cf(q->postfix[i].re * b, q->postfix[i].im * b);
    }
}

/* ================================================================== synchronizer */
enum { ST_SEEKPLCP = 0, ST_S0A, ST_S0B, ST_S1, ST_RXSYMBOLS };

struct ll_ofdmframesync_s {
    unsigned M, M2, cp;
    unsigned char *p;
    unsigned M_null, M_pilot, M_data, M_S0, M_S1, Nen;
    ll_cf *X, *x, *win;         /* win: M+cp samples, oldest first (view into wbuf) */
    ll_cf *wbuf; unsigned wcap, wi;
    ll_cf *S0, *s0, *S1, *s1;
    float g0;
    ll_cf *G0a, *G0b, *G, *B, *R;
    float *Ssm;                 /* [M][Nen] equaliser smoother */
    float *Pfit;                /* [2][M_pilot] */
    float *ytmp, *atmp;
    int state;
    ll_nco nco_rx;
    ll_msequence ms_pilot;
    float phi_prime, p1_prime;
    unsigned timer, num_symbols, backoff;
    ll_cf s_hat_0, s_hat_1;
    float plcp_detect_thresh, plcp_sync_thresh;
    ll_ofdmframesync_callback cb; void *ud;
};

ll_ofdmframesync ll_ofdmframesync_create(unsigned M, unsigned cp, unsigned taper, const unsigned char *p,
                                         ll_ofdmframesync_callback cb, void *ud)
{
    (void)taper;
    ll_ofdmframesync q = (ll_ofdmframesync)calloc(1, sizeof(*q));
    q->M = M; q->M2 = M / 2; q->cp = cp;
    q->p = (unsigned char *)malloc(M);
    if (p) memcpy(q->p, p, M); else ll_ofdmframe_init_default_sctype(M, q->p);
    ll_ofdmframe_validate_sctype(q->p, M, &q->M_null, &q->M_pilot, &q->M_data);
    q->Nen = q->M_pilot + q->M_data;
    q->X = (ll_cf *)calloc(M, sizeof(ll_cf)); q->x = (ll_cf *)calloc(M, sizeof(ll_cf));
    q->wcap = 8 * (M + cp); q->wi = 0;
    q->wbuf = (ll_cf *)calloc(q->wcap, sizeof(ll_cf));
    q->win = q->wbuf;
    q->S0 = (ll_cf *)calloc(M, sizeof(ll_cf)); q->s0 = (ll_cf *)calloc(M, sizeof(ll_cf));
    q->S1 = (ll_cf *)calloc(M, sizeof(ll_cf)); q->s1 = (ll_cf *)calloc(M, sizeof(ll_cf));
    ll_ofdmframe_init_S0(q->p, M, q->S0, q->s0, &q->M_S0);
    ll_ofdmframe_init_S1(q->p, M, q->S1, q->s1, &q->M_S1);
    q->g0 = 1.0f;
    q->G0a = (ll_cf *)calloc(M, sizeof(ll_cf)); q->G0b = (ll_cf *)calloc(M, sizeof(ll_cf));
    q->G = (ll_cf *)calloc(M, sizeof(ll_cf)); q->B = (ll_cf *)calloc(M, sizeof(ll_cf));
    q->R = (ll_cf *)calloc(M, sizeof(ll_cf));
    q->backoff = cp < 2 ? cp : 2;
    for (unsigned i = 0; i < M; i++) {
        double phi = (double)i * (double)q->backoff * 2.0 * M_PI / (double)M;
        q->B[i] = cf((float)cos(phi), (float)sin(phi));
    }
    q->Ssm = (float *)malloc(sizeof(float) * M * q->Nen);
    ll_ofdmframe_eq_smoother(q->p, M, 4, q->Ssm);
    q->Pfit = (float *)malloc(sizeof(float) * 2 * q->M_pilot);
    ll_ofdmframe_pilot_fit(q->p, M, q->Pfit);
    q->ytmp = (float *)malloc(sizeof(float) * M);
    q->atmp = (float *)malloc(sizeof(float) * M);
    ll_msequence_init_default(&q->ms_pilot, 8);
    q->cb = cb; q->ud = ud;
    ll_ofdmframesync_reset(q);
    return q;
}
void ll_ofdmframesync_destroy(ll_ofdmframesync q)
{
    if (!q) return;
    free(q->p); free(q->X); free(q->x); free(q->wbuf); free(q->S0); free(q->s0); free(q->S1); free(q->s1);
    free(q->G0a); free(q->G0b); free(q->G); free(q->B); free(q->R); free(q->Ssm); free(q->Pfit);
    free(q->ytmp); free(q->atmp); free(q);
}
void ll_ofdmframesync_reset(ll_ofdmframesync q)
{
    ll_nco_reset(&q->nco_rx);
    ll_msequence_reset(&q->ms_pilot);
    q->timer = 0; q->num_symbols = 0;
    q->s_hat_0 = cf(0, 0); q->s_hat_1 = cf(0, 0);
    q->phi_prime = 0; q->p1_prime = 0;
    q->plcp_detect_thresh = (q->M > 44) ? 0.35f : 0.35f + 0.01f * (float)(44 - q->M);
    q->plcp_sync_thresh   = (q->M > 44) ? 0.30f : 0.30f + 0.01f * (float)(44 - q->M);
    q->state = ST_SEEKPLCP;
}
float ll_ofdmframesync_get_rssi(ll_ofdmframesync q) { return -10.0f * log10f(q->g0); }
float ll_ofdmframesync_get_cfo(ll_ofdmframesync q)
{ return ll_nco_get_frequency(&q->nco_rx) / (2.0f * (float)M_PI); }
int ll_ofdmframesync_get_state(ll_ofdmframesync q) { return q->state; }

static void estimate_gain_S0(ll_ofdmframesync q, const ll_cf *x, ll_cf *G)
{
    ll_fft(q->M, x, q->X, 0);
    float gain = sqrtf((float)q->M_S0) / (float)q->M;
    for (unsigned i = 0; i < q->M; i++) {
        if (q->p[i] != LL_SCTYPE_NULL && (i % 2) == 0) {
            ll_cf t = cmulc(q->X[i], q->S0[i]);
            G[i] = cf(t.re * gain, t.im * gain);
        } else G[i] = cf(0, 0);
    }
}
static ll_cf S0_metrics(ll_ofdmframesync q, const ll_cf *G)
{
    ll_cf s = cf(0, 0);
    for (unsigned i = 0; i < q->M; i += 2) {
        ll_cf t = cmulc(G[(i + 2) % q->M], G[i]);
        s.re += t.re; s.im += t.im;
    }
    s.re /= (float)q->M_S0; s.im /= (float)q->M_S0;
    return s;
}
static void estimate_gain_S1(ll_ofdmframesync q, const ll_cf *x, ll_cf *G)
{
    ll_fft(q->M, x, q->X, 0);
    float gain = sqrtf((float)q->M_S1) / (float)q->M;
    for (unsigned i = 0; i < q->M; i++) {
        if (q->p[i] != LL_SCTYPE_NULL) {
            ll_cf t = cmulc(q->X[i], q->S1[i]);
            G[i] = cf(t.re * gain, t.im * gain);
        } else G[i] = cf(0, 0);
    }
}

/* order-4 polynomial smoothing of |G| and unwrapped arg(G) across frequency */
static void estimate_eqgain_poly(ll_ofdmframesync q)
{
    unsigned M = q->M, M2 = q->M2, Nen = q->Nen, n = 0;
    float *y_abs = q->ytmp, *y_arg = q->atmp;
    for (unsigned i = 0; i < M; i++) {
        unsigned k = (i + M2) % M;
        if (q->p[k] != LL_SCTYPE_NULL) { y_abs[n] = cabs_(q->G[k]); y_arg[n] = carg_(q->G[k]); n++; }
    }
    for (unsigned i = 1; i < Nen; i++) {
        while ((y_arg[i] - y_arg[i - 1]) >  (float)M_PI) y_arg[i] -= 2.0f * (float)M_PI;
        while ((y_arg[i] - y_arg[i - 1]) < -(float)M_PI) y_arg[i] += 2.0f * (float)M_PI;
    }
    for (unsigned i = 0; i < M; i++) {
        if (q->p[i] == LL_SCTYPE_NULL) { q->G[i] = cf(0, 0); continue; }
        float A = 0, th = 0;
        const float *row = q->Ssm + (size_t)i * Nen;
        for (unsigned nn = 0; nn < Nen; nn++) { A += row[nn] * y_abs[nn]; th += row[nn] * y_arg[nn]; }
        q->G[i] = cf(A * cosf(th), A * sinf(th));
    }
}

static void execute_seekplcp(ll_ofdmframesync q)
{
    q->timer++;
    if (q->timer < q->M) return;
    q->timer = 0;
    const ll_cf *rc = q->win;
    float g = 0.0f;
    for (unsigned i = q->cp; i < q->M + q->cp; i++) g += rc[i].re * rc[i].re + rc[i].im * rc[i].im;
    g = (float)q->M / g;
    estimate_gain_S0(q, rc + q->cp, q->G0a);
    ll_cf s_hat = S0_metrics(q, q->G0a);
    s_hat.re *= g; s_hat.im *= g;
    float tau_hat = carg_(s_hat) * (float)q->M2 / (2.0f * (float)M_PI);
    q->g0 = g;
    DBG("seek: g=%g |s|=%g tau=%g\n", g, cabs_(s_hat), tau_hat);
    if (cabs_(s_hat) > q->plcp_detect_thresh) {
        int dt = (int)roundf(tau_hat);
        q->timer = (unsigned)((int)q->M + dt) % q->M2;
        q->timer += q->M;
        q->state = ST_S0A;
    }
}
static void execute_S0a(ll_ofdmframesync q)
{
    q->timer++;
    if (q->timer < q->M2) return;
    q->timer = 0;
    const ll_cf *rc = q->win;
    estimate_gain_S0(q, rc + q->cp, q->G0a);
    ll_cf s_hat = S0_metrics(q, q->G0a);
    q->s_hat_0 = cf(s_hat.re * q->g0, s_hat.im * q->g0);
    DBG("S0a: |s|=%g arg=%g\n", cabs_(q->s_hat_0), carg_(q->s_hat_0));
    q->state = ST_S0B;
}
static void execute_S0b(ll_ofdmframesync q)
{
    q->timer++;
    if (q->timer < q->M2) return;
    q->timer = q->M + q->cp - q->backoff;
    const ll_cf *rc = q->win;
    estimate_gain_S0(q, rc + q->cp, q->G0b);
    ll_cf s_hat = S0_metrics(q, q->G0b);
    q->s_hat_1 = cf(s_hat.re * q->g0, s_hat.im * q->g0);
    ll_cf ssum = cf(q->s_hat_0.re + q->s_hat_1.re, q->s_hat_0.im + q->s_hat_1.im);
    float tau_hat = carg_(ssum) * (float)q->M2 / (2.0f * (float)M_PI);
    q->timer -= (unsigned)(int)roundf(tau_hat);
    /* CFO: time-domain ML estimate over the two halves of the oldest M samples */
    ll_cf t0 = cf(0, 0);
    for (unsigned i = 0; i < q->M2; i++) {
        ll_cf a = cmulc(q->s0[i], rc[i]);                       /* conj(rc[i]) * s0[i] */
        ll_cf b = cmulc(rc[i + q->M2], q->s0[i + q->M2]);       /* rc[i+M2] * conj(s0[i+M2]) */
        ll_cf t = cmul(a, b);
        t0.re += t.re; t0.im += t.im;
    }
    float nu_hat = carg_(t0) / (float)q->M2;
    ll_nco_set_frequency(&q->nco_rx, nu_hat);
    DBG("S0b: |s|=%g tau=%g nu=%g timer=%u\n", cabs_(q->s_hat_1), tau_hat, nu_hat, q->timer);
    q->state = ST_S1;
}
static void execute_S1(ll_ofdmframesync q)
{
    q->timer--;
    if (q->timer > 0) return;
    q->num_symbols++;
    const ll_cf *rc = q->win;
    estimate_gain_S1(q, rc + q->cp, q->G);
    ll_cf g_hat = cf(0, 0);
    for (unsigned i = 0; i < q->M; i++) {
        ll_cf t = cmulc(q->G[(i + 1) % q->M], q->G[i]);
        g_hat.re += t.re; g_hat.im += t.im;
    }
    g_hat.re /= (float)q->M_S1; g_hat.im /= (float)q->M_S1;
#if LL_S1_METRIC_G0_NORMALISED
    g_hat.re *= q->g0; g_hat.im *= q->g0;   /* level normalisation, as for the S0 metric (D7) */
#endif
    g_hat = cmul(g_hat, q->B[1]);      /* e^{j 2 pi backoff / M} */
    DBG("S1: |g|=%g arg=%g n=%u\n", cabs_(g_hat), carg_(g_hat), q->num_symbols);
    if (cabs_(g_hat) > q->plcp_sync_thresh && fabsf(carg_(g_hat)) < 0.1f * (float)M_PI) {
        q->state = ST_RXSYMBOLS;
        q->timer = q->M + q->cp + q->backoff;
        q->num_symbols = 0;
        float g = (float)q->M / sqrtf((float)(q->M_pilot + q->M_data));
        for (unsigned i = 0; i < q->M; i++) {
            ll_cf t = cf(q->G[i].re * g, q->G[i].im * g);
            /* The S1 window and every data window sit `backoff` samples inside the
             * cyclic prefix (S1 event at symbol end - backoff, data read at rc[cp-backoff]),
             * so G already carries the data windows' phase ramp and 1/G removes it.  liquid
             * lists a "timing backoff correction" G *= B here; with identical alignments it
             * would re-introduce the ramp, so it is not applied. */
#if LL_S1_BACKOFF_CORRECTION
            t = cmul(t, q->B[i]);           /* D6 */
#endif
            q->G[i] = t;
        }
        estimate_eqgain_poly(q);
        for (unsigned i = 0; i < q->M; i++) {
            if (q->p[i] == LL_SCTYPE_NULL) { q->R[i] = cf(0, 0); continue; }
            float d = q->G[i].re * q->G[i].re + q->G[i].im * q->G[i].im;
            q->R[i] = cf(q->G[i].re / d, -q->G[i].im / d);
        }
        return;
    }
    if (q->num_symbols == 16) ll_ofdmframesync_reset(q);
    q->timer = q->M2;       /* wait another half symbol */
}

static void rxsymbol(ll_ofdmframesync q)
{
    unsigned M = q->M, M2 = q->M2;
    for (unsigned i = 0; i < M; i++) q->X[i] = cmul(q->X[i], q->R[i]);
    float *y_phase = q->ytmp;
    unsigned n = 0;
    for (unsigned i = 0; i < M; i++) {
        unsigned k = (i + M2) % M;
        if (q->p[k] == LL_SCTYPE_PILOT) {
            float pilot = ll_msequence_advance(&q->ms_pilot) ? 1.0f : -1.0f;
            y_phase[n++] = atan2f(q->X[k].im * pilot, q->X[k].re * pilot);
        }
    }
    for (unsigned i = 1; i < q->M_pilot; i++) {
        while ((y_phase[i] - y_phase[i - 1]) >  (float)M_PI) y_phase[i] -= 2.0f * (float)M_PI;
        while ((y_phase[i] - y_phase[i - 1]) < -(float)M_PI) y_phase[i] += 2.0f * (float)M_PI;
    }
    float p0 = 0, p1 = 0;
    for (unsigned i = 0; i < q->M_pilot; i++) {
        p0 += q->Pfit[i] * y_phase[i];
        p1 += q->Pfit[q->M_pilot + i] * y_phase[i];
    }
#ifndef LL_P1_ALPHA
#define LL_P1_ALPHA 0.3f     /* slope smoothing of the pilot fit (experiments only: scratch/r6/d6_probe.py) */
#endif
    const float alpha = LL_P1_ALPHA;
    p1 = alpha * p1 + (1.0f - alpha) * q->p1_prime;
    q->p1_prime = p1;
    for (unsigned i = 0; i < M; i++) {
        if (q->p[i] == LL_SCTYPE_NULL) { q->X[i] = cf(0, 0); continue; }
        float fx = (i > M2) ? (float)i - (float)M : (float)i;
        float theta = p0 + p1 * fx;
        q->X[i] = cmul(q->X[i], cf(cosf(theta), -sinf(theta)));
    }
    if (q->num_symbols > 0) {
        float dphi = p0 - q->phi_prime;
        while (dphi >  (float)M_PI) dphi -= 2.0f * (float)M_PI;
        while (dphi < -(float)M_PI) dphi += 2.0f * (float)M_PI;
        ll_nco_adjust_frequency(&q->nco_rx, 1e-3f * dphi);
    }
    q->phi_prime = p0;
    q->num_symbols++;
}
static void execute_rxsymbols(ll_ofdmframesync q)
{
    q->timer--;
    if (q->timer != 0) return;
    memcpy(q->x, q->win + q->cp - q->backoff, sizeof(ll_cf) * q->M);
    ll_fft(q->M, q->x, q->X, 0);
    rxsymbol(q);
    if (q->cb) {
        int rv = q->cb(q->X, q->p, q->M, q->ud);
        if (rv != 0) ll_ofdmframesync_reset(q);
    }
    q->timer = q->M + q->cp;
}

void ll_ofdmframesync_execute(ll_ofdmframesync q, const ll_cf *xin, unsigned n)
{
    unsigned L = q->M + q->cp;
    for (unsigned i = 0; i < n; i++) {
        ll_cf x = xin[i];
        if (q->state != ST_SEEKPLCP) {
            x = ll_nco_mix_down(&q->nco_rx, x);
            ll_nco_step(&q->nco_rx);
        }
        q->wi++;
        if (q->wi + L > q->wcap) { memmove(q->wbuf, q->wbuf + q->wi, sizeof(ll_cf) * (L - 1)); q->wi = 0; }
        q->win = q->wbuf + q->wi;
        q->win[L - 1] = x;
        switch (q->state) {
        case ST_SEEKPLCP:  execute_seekplcp(q); break;
        case ST_S0A:       execute_S0a(q); break;
        case ST_S0B:       execute_S0b(q); break;
        case ST_S1:        execute_S1(q); break;
        case ST_RXSYMBOLS: execute_rxsymbols(q); break;
        }
    }
}
