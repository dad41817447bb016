// design.hpp -- host-side design data for the HIP receiver/transmitter (product code).
//
// Everything a liquid-dsp object computes once at create time, recomputed here from the
// published formulas so the kernels only see tables: Kaiser prototype of the polyphase
// bank (reference call: lib/multichannelrx.cc:89-91, lib/multichanneltx.cc:85-87), default
// subcarrier allocation, S0/S1 training symbols, pilot m-sequence, least-squares
// projection matrices of the equaliser / pilot-phase fits, modem soft-demod neighbour
// tables, Hamming(12,8) tables, Golay P matrix, CRC-32 tables (byte table and
// zero-advance tables for the lane-parallel CRC).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace mcrx {

struct cf { float re, im; };

enum { SC_NULL = 0, SC_PILOT = 1, SC_DATA = 2 };
enum { CRC_UNKNOWN = 0, CRC_NONE = 1, CRC_32 = 6 };
enum { FEC_UNKNOWN = 0, FEC_NONE = 1, FEC_REP3 = 2, FEC_REP5 = 3, FEC_HAMMING74 = 4, FEC_HAMMING84 = 5, FEC_HAMMING128 = 6, FEC_GOLAY2412 = 7,
       FEC_CONV_V27 = 11 };
// the schemes this path encodes and decodes as themselves; everything else is refused (MCRX_EUNSUPP), never sent under a false id
inline bool fec_supported(int fs) { return (fs >= FEC_NONE && fs <= FEC_GOLAY2412) || fs == FEC_CONV_V27; }
enum { MOD_UNKNOWN = 0, MOD_QAM16 = 27, MOD_QAM64 = 29, MOD_BPSK = 39, MOD_QPSK = 40 };

// ---------------------------------------------------------------- Kaiser prototype
inline double besseli0(double z)
{
    double y = 1.0, t = 1.0;
    for (int k = 1; k < 64; k++) { t *= (0.5 * z) / k; y += t * t; if (t * t < 1e-18 * y) break; }
    return y;
}
inline float kaiser_beta(float As)
{
    As = std::fabs(As);
    if (As > 50.0f) return 0.1102f * (As - 8.7f);
    if (As > 21.0f) return 0.5842f * std::pow(As - 21.0f, 0.4f) + 0.07886f * (As - 21.0f);
    return 0.0f;
}
// h[i] = sinc(2 fc t) * I0(beta sqrt(1 - (2t/n)^2)) / I0(beta),  t = i - (n-1)/2
inline std::vector<float> firdes_kaiser(unsigned n, float fc, float As)
{
    std::vector<float> h(n);
    double beta = kaiser_beta(As), ib = besseli0(beta);
    for (unsigned i = 0; i < n; i++) {
        double t = (double)i - (double)(n - 1) / 2.0;
        double x = 2.0 * (double)fc * t;
        double s = (std::fabs(x) < 1e-9) ? 1.0 : std::sin(M_PI * x) / (M_PI * x);
        double r = 2.0 * t / (double)n, a = 1.0 - r * r;
        h[i] = (float)(s * besseli0(beta * std::sqrt(a < 0 ? 0 : a)) / ib);
    }
    return h;
}
// polyphase prototype of firpfbch_crcf_create_kaiser(K, m, As): first 2*m*K taps of a (2Km+1)-tap design
inline std::vector<float> pfb_prototype(unsigned K, unsigned m, float As)
{
    std::vector<float> h = firdes_kaiser(2 * K * m + 1, 0.5f / (float)K, As);
    h.resize((size_t)2 * m * K);
    return h;
}

// prototype of firpfbch2_crcf_create_kaiser(LIQUID_ANALYZER, M, m, As): (2Mm+1)-tap Kaiser design at fc = 1/M,
// scaled so that the taps sum to M (float accumulation in index order, like the oracle), first 2*m*M taps used
inline std::vector<float> pfb2_prototype(unsigned M, unsigned m, float As)
{
    std::vector<float> h = firdes_kaiser(2 * M * m + 1, 1.0f / (float)M, As);
    float sum = 0.0f;
    for (float v : h) sum += v;
    for (float &v : h) v = v * (float)M / sum;
    h.resize((size_t)2 * m * M);
    return h;
}

// ---- column tap tables of channelizer.hip: tap[j][n] = column n's tap on the block j back (j = 0: the newest block)
// the reference's bank (firpfbch analyzer, lib/multichannelrx.cc:89-91): V_b[n] = sum_j h[K-1-n + j K] u[(b-j) K + n]
inline std::vector<float> pfb_column_taps(const std::vector<float> &h, unsigned K)
{
    const unsigned P = (unsigned)(h.size() / K);
    std::vector<float> t((size_t)P * K);
    for (unsigned j = 0; j < P; j++)
        for (unsigned n = 0; n < K; n++) t[(size_t)j * K + n] = h[(size_t)(K - 1 - n) + (size_t)j * K];
    return t;
}
// half-band branch filter of liquid's resamp2_crcf (m = 7): the odd taps of a 29-tap Kaiser design, reversed
inline std::vector<float> halfband_branch_taps(unsigned m, float As)
{
    const unsigned n = 4 * m + 1;
    std::vector<float> hh = firdes_kaiser(n, 0.25f, As), h1(2 * m);
    unsigned j = 0;
    for (unsigned i = 1; i < n; i += 2) h1[j++] = hh[n - i - 1];
    return h1;
}
// The oversampled front end as ONE critically sampled bank (cfg.front_end = 1; channelizer.hip's header).  With M = 2N channels,
// h = the firpfbch2 prototype (14 M taps), h1 = the half-band branch filter, the chain
//     Z_s[r] = sum_{j<14} h[r + j M] u[(s+1) M/2 - 1 - r - j M],   Y_s[c] = (-1)^(c s) / M  sum_r Z_s[r] e^{+j 2 pi r c / M}
//     out_k[c] = 0.5 ( Y_{2k-13}[c] + sum_{i<14} h1[i] Y_{2(k-13+i)}[c] )                                   (c < N)
// is, because the half-band filter acts along time and the transform along r and (-1)^c e^{j 2 pi r c / M} = e^{j 2 pi (r + M/2) c / M},
//     out_k[c] = sum_r W_k[r] e^{+j 2 pi r c / M},   W_k[r] = 0.5 / M ( sum_i h1[i] Z_{2(k-13+i)}[r] + Z_{2k-13}[(r - M/2) mod M] ).
// Both terms of W_k[r] read column n = (M/2 - 1 - r) mod M of the blocks k, k-1, ..., k-27:  W_k[r] = sum_{d<28} G[n][d] u[(k-d) M + n],
// and e^{+j 2 pi r c / M} = e^{-j 2 pi (n + M/2 + 1) c / M}: a forward transform of the columns rotated by M/2 + 1.
// Taps are formed in double from the float prototypes the stage-by-stage chain uses and rounded once; 0.5 / M is a power of two.
// Returns tap[d][n], 28 * M floats.  (scratch/r6/composite_check.py holds the same algebra against the oracle's chain.)
inline std::vector<float> pfb2_composite_taps(const std::vector<float> &h, const std::vector<float> &h1, unsigned M)
{
    const unsigned P = 14, PC = 28;
    std::vector<double> G((size_t)PC * M, 0.0);
    for (unsigned r = 0; r < M; r++) {
        const unsigned n = (M / 2 - 1 - r + M) % M, up = r >= M / 2 ? 1u : 0u;
        for (unsigned i = 0; i < P; i++)
            for (unsigned j = 0; j < P; j++)
                G[(size_t)(13 - i + j + up) * M + n] += (double)h1[i] * (double)h[r + (size_t)j * M];
        const unsigned rp = (r + M - M / 2) % M;
        for (unsigned j = 0; j < P; j++) G[(size_t)(7 + j) * M + n] += (double)h[rp + (size_t)j * M];
    }
    std::vector<float> t((size_t)PC * M);
    const double g = 0.5 / (double)M;
    for (size_t i = 0; i < t.size(); i++) t[i] = (float)(G[i] * g);
    return t;
}

// ---------------------------------------------------------------- NCO
inline uint32_t rad2u32(float rad)
{
    double p = (double)rad * (1.0 / (2.0 * M_PI));
    p -= std::floor(p);
    return (uint32_t)(uint64_t)std::llrint(p * 4294967296.0);
}
inline uint32_t channel_center_step(unsigned N)   // lib/multichannelrx.cc:98
{
    float f = -0.5f * (float)(N - 1) / (float)N;
    return rad2u32((float)((double)f * M_PI));
}

// ---------------------------------------------------------------- m-sequence
struct MSeq {
    unsigned m, g, n, v;
    explicit MSeq(unsigned m_) {
        static const unsigned poly[16] = { 0, 0, 0x7, 0xB, 0x13, 0x25, 0x43, 0x89, 0x11D, 0x211,
                                           0x409, 0x805, 0x1053, 0x201b, 0x402b, 0x8003 };
        m = m_; g = poly[m] >> 1; n = (1u << m) - 1; v = 1u << (m - 1);
    }
    unsigned advance() { unsigned b = (unsigned)__builtin_parity(v & g); v = ((v << 1) | b) & n; return b; }
    unsigned symbol(unsigned bps) { unsigned s = 0; for (unsigned i = 0; i < bps; i++) s = (s << 1) | advance(); return s; }
};

// ---------------------------------------------------------------- OFDM frame design
struct OfdmDesign {
    unsigned M = 0, M2 = 0, cp = 0, taper = 0, backoff = 0;
    unsigned M_null = 0, M_pilot = 0, M_data = 0, M_S0 = 0, M_S1 = 0, Nen = 0;
    std::vector<uint8_t> p;            // subcarrier types
    std::vector<float> S0, S1;         // +-1 / 0 per bin
    std::vector<cf> s0, s1;            // time-domain training symbols (unit power)
    std::vector<float> Ssm;            // [M][Nen] equaliser smoother (order-4 LSQ projection)
    // the same projection factored through an orthonormal basis of the fit: Ssm = smk * smn^T
    // (rank = order + 1 <= 5, unused columns zero), so the GPU needs 5 wave sums instead of Nen-term rows
    std::vector<float> smk;            // [M][5]   basis evaluated at every bin
    std::vector<float> smn;            // [Nen][5] orthonormal basis on the enabled bins
    std::vector<float> Pfit;           // [2][M_pilot] pilot phase line fit
    std::vector<int> data_rank;        // rank of bin among data bins (ascending bin), -1 otherwise
    std::vector<int> pilot_rank;       // rank of bin among pilots in fft-shifted order, -1 otherwise
    std::vector<int> en_rank;          // rank among enabled bins in fft-shifted order, -1 otherwise
    std::vector<float> taperwin;       // raised-cosine ramp, taper samples
    uint8_t pilot_seq[255];            // order-8 m-sequence bits
    float detect_thresh = 0.35f, sync_thresh = 0.30f;
    bool ok = false;

    static void lsq(const std::vector<double> &x, unsigned k, std::vector<double> &C)
    {
        unsigned n = (unsigned)x.size();
        std::vector<double> A((size_t)k * 2 * k);
        for (unsigned a = 0; a < k; a++)
            for (unsigned b = 0; b < 2 * k; b++) {
                double s = 0;
                if (b < k) for (unsigned i = 0; i < n; i++) s += std::pow(x[i], (double)(a + b));
                else s = ((b - k) == a) ? 1.0 : 0.0;
                A[a * 2 * k + b] = s;
            }
        for (unsigned c = 0; c < k; c++) {
            unsigned piv = c;
            for (unsigned r = c + 1; r < k; r++) if (std::fabs(A[r * 2 * k + c]) > std::fabs(A[piv * 2 * k + c])) piv = r;
            if (piv != c) for (unsigned b = 0; b < 2 * k; b++) std::swap(A[c * 2 * k + b], A[piv * 2 * k + b]);
            double d = A[c * 2 * k + c];
            for (unsigned b = 0; b < 2 * k; b++) A[c * 2 * k + b] /= d;
            for (unsigned r = 0; r < k; r++) if (r != c) {
                double f = A[r * 2 * k + c];
                if (f != 0.0) for (unsigned b = 0; b < 2 * k; b++) A[r * 2 * k + b] -= f * A[c * 2 * k + b];
            }
        }
        C.assign((size_t)k * n, 0.0);
        for (unsigned a = 0; a < k; a++)
            for (unsigned i = 0; i < n; i++) {
                double s = 0;
                for (unsigned b = 0; b < k; b++) s += A[a * 2 * k + k + b] * std::pow(x[i], (double)b);
                C[(size_t)a * n + i] = s;
            }
    }

    void training(bool longseq, std::vector<float> &S, std::vector<cf> &s, unsigned &cnt)
    {
        unsigned m = 0, x = M - 1; while (x) { x >>= 1; m++; }
        if (m < 4) m = 4; else if (m > 8) m = 8;
        if (longseq) m++;
        MSeq ms(m);
        S.assign(M, 0.0f); cnt = 0;
        for (unsigned i = 0; i < M; i++) {
            unsigned b = ms.symbol(3) & 1;
            if (p[i] != SC_NULL && (longseq || (i % 2) == 0)) { S[i] = b ? 1.0f : -1.0f; cnt++; }
        }
        s.resize(M);
        double g = 1.0 / std::sqrt((double)cnt);
        for (unsigned n = 0; n < M; n++) {
            double re = 0, im = 0;
            for (unsigned k = 0; k < M; k++) if (S[k] != 0.0f) {
                double a = 2.0 * M_PI * (double)((uint64_t)k * n % M) / (double)M;
                re += S[k] * std::cos(a); im += S[k] * std::sin(a);
            }
            s[n].re = (float)(re * g); s[n].im = (float)(im * g);
        }
    }

    int init(unsigned M_, unsigned cp_, unsigned taper_, const unsigned char *p_)
    {
        M = M_; M2 = M / 2; cp = cp_; taper = taper_; backoff = cp < 2 ? cp : 2;
        p.assign(M, SC_NULL);
        if (p_) std::memcpy(p.data(), p_, M);
        else {
            unsigned G = M / 10; if (G < 2) G = 2;
            unsigned P = (M > 34) ? 8 : 4, P2 = P / 2;
            for (unsigned i = 1; i < M2 - G; i++) {
                uint8_t t = (((i + P2) % P) == 0) ? SC_PILOT : SC_DATA;
                p[i] = t; p[M - i] = t;
            }
        }
        M_null = M_pilot = M_data = 0;
        for (unsigned i = 0; i < M; i++) {
            if (p[i] == SC_NULL) M_null++; else if (p[i] == SC_PILOT) M_pilot++;
            else if (p[i] == SC_DATA) M_data++; else return -1;
        }
        if (M_data == 0 || M_pilot < 2) return -1;
        Nen = M_pilot + M_data;
        training(false, S0, s0, M_S0);
        training(true, S1, s1, M_S1);
        { MSeq ms(8); for (int i = 0; i < 255; i++) pilot_seq[i] = (uint8_t)ms.advance(); }
        data_rank.assign(M, -1); pilot_rank.assign(M, -1); en_rank.assign(M, -1);
        int d = 0; for (unsigned i = 0; i < M; i++) if (p[i] == SC_DATA) data_rank[i] = d++;
        std::vector<double> xe, xp;
        int np = 0, ne = 0;
        for (unsigned i = 0; i < M; i++) {
            unsigned k = (i + M2) % M;
            double f = (k > M2) ? (double)k - (double)M : (double)k;
            if (p[k] != SC_NULL) { en_rank[k] = ne++; xe.push_back(f / (double)M); }
            if (p[k] == SC_PILOT) { pilot_rank[k] = np++; xp.push_back(f); }
        }
        unsigned order = 4;
        if (order > Nen - 1) order = Nen - 1;
        std::vector<double> C;
        lsq(xe, order + 1, C);
        Ssm.assign((size_t)M * Nen, 0.0f);
        for (unsigned i = 0; i < M; i++) {
            if (p[i] == SC_NULL) continue;
            double f = ((i > M2) ? (double)i - (double)M : (double)i) / (double)M;
            for (unsigned n = 0; n < Nen; n++) {
                double s = 0;
                for (unsigned a = 0; a <= order; a++) s += std::pow(f, (double)a) * C[(size_t)a * Nen + n];
                Ssm[(size_t)i * Nen + n] = (float)s;
            }
        }
        {   // thin QR (modified Gram-Schmidt, double) of the Vandermonde matrix on the enabled bins:
            // V = Q R, projection = Phi R^-1 Q^T
            const unsigned ks = order + 1;
            std::vector<double> Q((size_t)Nen * ks), Rm((size_t)ks * ks, 0.0);
            for (unsigned n = 0; n < Nen; n++) for (unsigned a = 0; a < ks; a++) Q[(size_t)n * ks + a] = std::pow(xe[n], (double)a);
            for (unsigned a = 0; a < ks; a++) {
                for (unsigned b = 0; b < a; b++) {
                    double d = 0; for (unsigned n = 0; n < Nen; n++) d += Q[(size_t)n * ks + b] * Q[(size_t)n * ks + a];
                    Rm[b * ks + a] = d;
                    for (unsigned n = 0; n < Nen; n++) Q[(size_t)n * ks + a] -= d * Q[(size_t)n * ks + b];
                }
                double nr = 0; for (unsigned n = 0; n < Nen; n++) nr += Q[(size_t)n * ks + a] * Q[(size_t)n * ks + a];
                nr = std::sqrt(nr); Rm[a * ks + a] = nr;
                for (unsigned n = 0; n < Nen; n++) Q[(size_t)n * ks + a] /= nr;
            }
            smk.assign((size_t)M * 5, 0.0f); smn.assign((size_t)Nen * 5, 0.0f);
            for (unsigned n = 0; n < Nen; n++) for (unsigned a = 0; a < ks; a++) smn[(size_t)n * 5 + a] = (float)Q[(size_t)n * ks + a];
            for (unsigned i = 0; i < M; i++) {
                if (p[i] == SC_NULL) continue;
                const double f = ((i > M2) ? (double)i - (double)M : (double)i) / (double)M;
                double row[5];
                for (unsigned d = 0; d < ks; d++) {                 // row = phi R^-1 by forward substitution (R upper triangular)
                    double v = std::pow(f, (double)d);
                    for (unsigned b = 0; b < d; b++) v -= row[b] * Rm[b * ks + d];
                    row[d] = v / Rm[d * ks + d];
                }
                for (unsigned d = 0; d < ks; d++) smk[(size_t)i * 5 + d] = (float)row[d];
            }
        }
        lsq(xp, 2, C);
        Pfit.resize((size_t)2 * M_pilot);
        for (unsigned i = 0; i < 2 * M_pilot; i++) Pfit[i] = (float)C[i];
        taperwin.resize(taper ? taper : 1);
        for (unsigned i = 0; i < taper; i++) {
            double t = ((double)i + 0.5) / (double)taper, g = std::sin(M_PI_2 * t);
            taperwin[i] = (float)(g * g);
        }
        detect_thresh = (M > 44) ? 0.35f : 0.35f + 0.01f * (float)(44 - M);
        sync_thresh   = (M > 44) ? 0.30f : 0.30f + 0.01f * (float)(44 - M);
        ok = true;
        return 0;
    }
};

// ---------------------------------------------------------------- coding tables
inline unsigned hamming128_encode(unsigned s)
{
    auto par = [](unsigned v) { return (unsigned)__builtin_parity(v); };
    unsigned p1 = par(s & 0xda), p2 = par(s & 0xb6), p4 = par(s & 0x71), p8 = par(s & 0x0f);
    return (s & 0x0f) | ((s & 0x70) << 1) | ((s & 0x80) << 2) | (p1 << 11) | (p2 << 10) | (p4 << 8) | (p8 << 4);
}
static const unsigned golay_P[12] = { 0x08ed, 0x01db, 0x03b5, 0x0769, 0x0ed1, 0x0da3,
                                      0x0b47, 0x068f, 0x0d1d, 0x0a3b, 0x0477, 0x0ffe };

struct CodingTables {
    uint32_t crc_byte[256];            // reflected CRC-32 byte table
    uint32_t crc_zadv[16][4][256];     // advance the CRC state through 2^k zero bytes (k = 0..15), by state byte
    uint8_t  qam16_nb[16][4];          // soft-demod nearest neighbours
    uint8_t  qam64_nb[64][4];

    static unsigned gray_decode(unsigned x) { unsigned y = x; while (x >>= 1) y ^= x; return y; }
    static void qam_nb(unsigned bps, uint8_t *nb)
    {
        unsigned Mq = 1u << bps, mq = bps / 2, L = 1u << mq;
        auto grid = [&](unsigned s, int &gi, int &gq) {
            gi = 2 * (int)gray_decode(s >> mq) - (int)L + 1;
            gq = 2 * (int)gray_decode(s & (L - 1)) - (int)L + 1;
        };
        for (unsigned i = 0; i < Mq; i++) {
            int ai, aq; grid(i, ai, aq);
            for (unsigned k = 0; k < 4; k++) {
                long dmin = 1L << 40; unsigned best = Mq;
                for (unsigned j = 0; j < Mq; j++) {
                    bool ok = (j != i);
                    for (unsigned l = 0; l < k; l++) if (nb[i * 4 + l] == j) ok = false;
                    if (!ok) continue;
                    int bi, bq; grid(j, bi, bq);
                    long d = (long)(ai - bi) * (ai - bi) + (long)(aq - bq) * (aq - bq);
                    if (d < dmin) { dmin = d; best = j; }
                }
                nb[i * 4 + k] = (uint8_t)best;
            }
        }
    }
    CodingTables()
    {
        for (unsigned b = 0; b < 256; b++) {
            uint32_t c = b;
            for (int j = 0; j < 8; j++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1)));
            crc_byte[b] = c;
        }
        // zero-advance operators: A_0 = one zero byte; A_{k+1} = A_k applied twice
        auto adv1 = [&](uint32_t s) { return (s >> 8) ^ crc_byte[s & 0xff]; };
        uint32_t col[32];
        for (int b = 0; b < 32; b++) col[b] = adv1(1u << b);
        for (int k = 0; k < 16; k++) {
            for (int byte = 0; byte < 4; byte++)
                for (unsigned v = 0; v < 256; v++) {
                    uint32_t r = 0;
                    for (int b = 0; b < 8; b++) if (v & (1u << b)) r ^= col[8 * byte + b];
                    crc_zadv[k][byte][v] = r;
                }
            uint32_t nxt[32];
            for (int b = 0; b < 32; b++) {
                uint32_t s = col[b], r = 0;
                for (int c = 0; c < 32; c++) if (s & (1u << c)) r ^= col[c];
                nxt[b] = r;
            }
            std::memcpy(col, nxt, sizeof(col));
        }
        qam_nb(4, &qam16_nb[0][0]);
        qam_nb(6, &qam64_nb[0][0]);
    }
};

inline unsigned fec_enc_len(int fs, unsigned n)
{
    if (fs == FEC_HAMMING128) return (n / 2) * 3 + (n % 2) * 2;
    if (fs == FEC_GOLAY2412) return (n / 3) * 6 + (n % 3) * 3;
    if (fs == FEC_CONV_V27) return 2 * n + 2;               // r = 1/2, K = 7: 2 (8 n + 6) bits
    if (fs == FEC_REP3) return 3 * n;
    if (fs == FEC_REP5) return 5 * n;
    if (fs == FEC_HAMMING74) return (14 * n + 7) / 8;        // liquid fec_block_get_enc_msg_len(n, 4, 7)
    if (fs == FEC_HAMMING84) return 2 * n;
    return n;
}
// liquid packetizer.c: every stage but an uncoded one is followed by the four-pass interleaver
inline unsigned fec_il_depth(int fs) { return (fs == FEC_NONE || fs == FEC_UNKNOWN) ? 0u : 4u; }
inline unsigned packet_enc_len(unsigned n, int crc, int fec0, int fec1)
{ return fec_enc_len(fec1, fec_enc_len(fec0, n + (crc == CRC_32 ? 4 : 0))); }
inline unsigned mod_bps(int mod)
{
    switch (mod) { case MOD_BPSK: return 1; case MOD_QPSK: return 2; case MOD_QAM16: return 4; case MOD_QAM64: return 6; default: return 0; }
}

}  // namespace mcrx
