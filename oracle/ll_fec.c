/*
 * ll_fec.c -- CPU ORACLE (test infrastructure): CRC, block FEC, interleaver, scrambler,
 * bit repacking and the packetizer (CRC -> fec0 -> interleave -> fec1 -> interleave).
 *
 * Restates liquid-dsp src/fec/src/{crc.c, fec_hamming128.c, fec_golay2412.c, fec.c,
 * interleaver.c, packetizer.c}, src/framing/... scramble.c and src/utility pack_bytes.c.
 * The reference selects these through ofdmflexframegenprops_s {check, fec0, fec1}
 * (/root/reference/lib/multichanneltx.cc:72-75,184; src/multichannel_txrx.cc:131-132):
 * CRC-32, fec0 = none, fec1 = Hamming(12,8) or Golay(24,12); the applications' -c / -k options hand any
 * liquid scheme name to the library (src/multichannel_tx.cc:46-52,92-94), so the short block codes those
 * options list first -- rep3, rep5, Hamming(7,4), Hamming(8,4): liquid fec_rep3.c, fec_rep5.c, fec_hamming74.c,
 * fec_hamming84.c -- are restated too.
 *
 *  CRC-32      : reflected 0xEDB88320, init/xorout 0xFFFFFFFF, appended big-endian.
 *  Hamming128  : 12-bit symbol  p1 p2 d1 p4 d2 d3 d4 p8 d5 d6 d7 d8 (MSB first);
 *                two symbols packed in three bytes.  Soft decode: hard decision,
 *                then the re-encoded estimate and its distance-3 neighbour codewords
 *                are compared by soft distance.
 *  Golay2412   : systematic extended Golay, codeword = (parity << 12) | message,
 *                parity = P * m with the Lin/Costello P matrix; arithmetic decoder
 *                (syndrome weight tests), corrects <= 3 errors.  No soft decoder
 *                (soft input is sliced at 127 and hard decoded).
 *  rep3 / rep5 : the message three / five times in a row; hard decode = bitwise majority; soft decode = mean of
 *                the copies' soft bits (integer division) against the erasure level 127.
 *  Hamming74/84: 4-bit symbol -> p1 p2 d1 p4 d2 d3 d4 (MSB first) [+ overall parity as the LSB]; a byte is two
 *                symbols, high nibble first; (7,4) symbols are bit-packed back to back, (8,4) symbols are bytes.
 *                Soft decode: the codeword of least soft distance among all 16, ascending, strict <.  Hard decode:
 *                nearest codeword the same way ((7,4) is perfect: unique; a double error in (8,4) goes to the
 *                lowest symbol at distance 2 -- liquid's table for that case is not visible from the reference).
 *  interleaver : byte swaps x[2i] <-> x[2j+1] over an M x N column walk, then three
 *                masked passes (N+2,0x0f) (N+4,0x55) (N+8,0x33).
 *  scrambler   : XOR with repeating {0xb4, 0x6a, 0x8b, 0xc5}.
 */
#include "liquidlite.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ CRC */
unsigned ll_crc_length(int scheme)
{
    switch (scheme) {
    case LL_CRC_NONE: return 0;
    case LL_CRC_CHECKSUM: case LL_CRC_8: return 1;
    case LL_CRC_16: return 2;
    case LL_CRC_24: return 3;
    case LL_CRC_32: return 4;
    default: return 0;
    }
}
static unsigned reverse_bits(unsigned v, unsigned n)
{ unsigned r = 0; for (unsigned i = 0; i < n; i++) if (v & (1u << i)) r |= 1u << (n - 1 - i); return r; }

static unsigned crc_reflected(const unsigned char *msg, unsigned n, unsigned poly, unsigned bits)
{
    unsigned mask = (bits == 32) ? 0xffffffffu : ((1u << bits) - 1);
    unsigned rpoly = reverse_bits(poly, bits);
    unsigned key = mask;
    for (unsigned i = 0; i < n; i++) {
        key ^= msg[i];
        for (unsigned j = 0; j < 8; j++) key = (key >> 1) ^ (rpoly & (0u - (key & 1)));
    }
    return (~key) & mask;
}
unsigned ll_crc_generate_key(int scheme, const unsigned char *msg, unsigned n)
{
    switch (scheme) {
    case LL_CRC_CHECKSUM: {
        unsigned sum = 0;
        for (unsigned i = 0; i < n; i++) sum += msg[i];
        return (unsigned)((~(sum & 0xff) + 1) & 0xff);
    }
    case LL_CRC_8:  return crc_reflected(msg, n, 0x07, 8);
    case LL_CRC_16: return crc_reflected(msg, n, 0x8005, 16);
    case LL_CRC_24: return crc_reflected(msg, n, 0x5D6DCB, 24);
    case LL_CRC_32: return crc_reflected(msg, n, 0x04C11DB7, 32);
    default: return 0;
    }
}

/* ------------------------------------------------------------------ Hamming(12,8) */
#define H128_M1 0x00da
#define H128_M2 0x00b6
#define H128_M4 0x0071
#define H128_M8 0x000f
#define H128_S1 0x0aaa
#define H128_S2 0x0666
#define H128_S4 0x01e1
#define H128_S8 0x001f
static unsigned par(unsigned v) { return (unsigned)__builtin_parity(v); }

unsigned ll_hamming128_encode_symbol(unsigned s)
{
    unsigned p1 = par(s & H128_M1), p2 = par(s & H128_M2), p4 = par(s & H128_M4), p8 = par(s & H128_M8);
    return (s & 0x000f) | ((s & 0x0070) << 1) | ((s & 0x0080) << 2) |
           (p1 << 11) | (p2 << 10) | (p4 << 8) | (p8 << 4);
}
unsigned ll_hamming128_decode_symbol(unsigned c)
{
    unsigned z = (par(c & H128_S8) << 3) | (par(c & H128_S4) << 2) | (par(c & H128_S2) << 1) | par(c & H128_S1);
    if (z && z <= 12) c ^= 1u << (12 - z);
    return (c & 0x000f) | ((c & 0x00e0) >> 1) | ((c & 0x0200) >> 2);
}

#define H128_MAXNB 32
static int h128_init = 0;
static unsigned short h128_enc[256];
static unsigned char h128_nb[256][H128_MAXNB];
static unsigned char h128_nnb[256];
static void h128_tables(void)
{
    if (h128_init) return;
    for (unsigned s = 0; s < 256; s++) h128_enc[s] = (unsigned short)ll_hamming128_encode_symbol(s);
    for (unsigned s = 0; s < 256; s++) {
        unsigned n = 0;
        for (unsigned t = 0; t < 256; t++)
            if (t != s && __builtin_popcount(h128_enc[s] ^ h128_enc[t]) == 3 && n < H128_MAXNB)
                h128_nb[s][n++] = (unsigned char)t;
        h128_nnb[s] = (unsigned char)n;
    }
    h128_init = 1;
}
static unsigned h128_soft_dist(unsigned c, const unsigned char *soft)
{
    unsigned d = 0;
    for (unsigned k = 0; k < 12; k++) d += ((c >> (11 - k)) & 1) ? 255u - soft[k] : soft[k];
    return d;
}
static unsigned h128_decode_soft_symbol(const unsigned char *soft)
{
    h128_tables();
    unsigned c = 0;
    for (unsigned k = 0; k < 12; k++) c = (c << 1) | (soft[k] > 127 ? 1u : 0u);
    unsigned s0 = ll_hamming128_decode_symbol(c);
    unsigned s_hat = s0;
    unsigned dmin = h128_soft_dist(h128_enc[s0], soft);
    for (unsigned i = 0; i < h128_nnb[s0]; i++) {
        unsigned t = h128_nb[s0][i];
        unsigned d = h128_soft_dist(h128_enc[t], soft);
        if (d < dmin) { dmin = d; s_hat = t; }
    }
    return s_hat;
}

/* ------------------------------------------------------------------ Golay(24,12) */
static const unsigned golay_P[12] = {
    0x08ed, 0x01db, 0x03b5, 0x0769, 0x0ed1, 0x0da3,
    0x0b47, 0x068f, 0x0d1d, 0x0a3b, 0x0477, 0x0ffe };

/* y = v * P  (row vector times matrix; bit 11 of v selects row 0) */
static unsigned golay_mulP(unsigned v)
{
    unsigned y = 0;
    for (unsigned i = 0; i < 12; i++) if (v & (1u << (11 - i))) y ^= golay_P[i];
    return y;
}
unsigned ll_golay2412_encode_symbol(unsigned s)
{
    s &= 0xfff;
    return (golay_mulP(s) << 12) | s;    /* P is symmetric: m*P == P*m */
}
unsigned ll_golay2412_decode_symbol(unsigned r)
{
    /* r = (parity', message'); codeword = (m*P, m).  H = [I | P]:  s = parity' + message'*P */
    unsigned rp = (r >> 12) & 0xfff, rm = r & 0xfff;
    unsigned s = rp ^ golay_mulP(rm);
    unsigned ep = 0, em = 0;    /* error estimates in parity / message halves */
    int found = 0;
    if (__builtin_popcount(s) <= 3) { ep = s; em = 0; found = 1; }
    if (!found) {
        for (unsigned i = 0; i < 12; i++) {
            unsigned t = s ^ golay_P[i];
            if (__builtin_popcount(t) <= 2) { ep = t; em = 1u << (11 - i); found = 1; break; }
        }
    }
    if (!found) {
        unsigned sP = golay_mulP(s);
        if (__builtin_popcount(sP) <= 3) { ep = 0; em = sP; found = 1; }
        else {
            for (unsigned i = 0; i < 12; i++) {
                unsigned t = sP ^ golay_P[i];
                if (__builtin_popcount(t) <= 2) { ep = 1u << (11 - i); em = t; found = 1; break; }
            }
        }
    }
    (void)ep;
    return (rm ^ (found ? em : 0)) & 0xfff;
}

/* ------------------------------------------------------------------ r = 1/2, K = 7 convolutional code
 * liquid fec_conv.c (LIQUID_FEC_CONV_V27) over libfec's viterbi27: generator polynomials 0x6d, 0x4f on the shift register
 * sr = (sr << 1) | bit, message bits MSB first, K-1 zero tail bits, output bits MSB first, last byte zero padded:
 * 2 (8 n + 6) bits = 2 n + 2 bytes.  The decoder is the maximum-likelihood Viterbi decoder over 8-bit soft symbols
 * (0 = certainly 0, 255 = certainly 1; hard decisions are 0 / 255), branch metric = sum of |symbol - expected|, path metrics
 * in 32-bit integers (no renormalisation needed below 2^31 / 510 steps), ties to the predecessor with the older bit 0,
 * full traceback from state 0.  (libfec keeps 64-bit decision words per step and traces back the same way; its
 * tie-breaking is not visible from the reference -- parity unpinned like the rest of this file.) */
static unsigned conv27_parity(unsigned v) { v ^= v >> 4; v ^= v >> 2; v ^= v >> 1; return v & 1u; }
static void conv27_encode(unsigned n, const unsigned char *dec, unsigned char *enc)
{
    unsigned sr = 0, acc = 0, nb = 0, j = 0;
    for (unsigned i = 0; i < 8 * n + 6; i++) {
        unsigned bit = i < 8 * n ? (dec[i >> 3] >> (7 - (i & 7))) & 1u : 0u;
        sr = ((sr << 1) | bit) & 0x7f;
        acc = (acc << 1) | conv27_parity(sr & 0x6d); acc = (acc << 1) | conv27_parity(sr & 0x4f); nb += 2;
        if (nb == 8) { enc[j++] = (unsigned char)acc; acc = 0; nb = 0; }
    }
    if (nb) enc[j++] = (unsigned char)(acc << (8 - nb));
}
/* sym: 2 soft symbols per step */
static void conv27_viterbi(unsigned n, const unsigned char *sym, unsigned char *dec)
{
    const unsigned T = 8 * n + 6;
    unsigned long long *d = (unsigned long long *)malloc(sizeof(unsigned long long) * (T ? T : 1));
    int pm[64], nm[64];
    for (int s = 0; s < 64; s++) pm[s] = s ? (1 << 28) : 0;               /* the encoder starts in state 0 */
    for (unsigned t = 0; t < T; t++) {
        const int sa = sym[2 * t], sb = sym[2 * t + 1];
        unsigned long long w = 0;
        for (int s = 0; s < 64; s++) {                                      /* new state s = (prev << 1 | bit) & 63 */
            const int p0 = s >> 1, p1 = (s >> 1) | 32;
            const int bm0 = (conv27_parity((unsigned)s & 0x6d) ? 255 - sa : sa) + (conv27_parity((unsigned)s & 0x4f) ? 255 - sb : sb);
            const int m0 = pm[p0] + bm0, m1 = pm[p1] + (510 - bm0);       /* both polynomials tap the oldest bit */
            const int take1 = m1 < m0;
            nm[s] = take1 ? m1 : m0;
            w |= (unsigned long long)take1 << s;
        }
        d[t] = w;
        memcpy(pm, nm, sizeof(pm));
    }
    memset(dec, 0, n);
    unsigned state = 0;
    for (unsigned t = T; t-- > 0;) {
        if (t < 8 * n) dec[t >> 3] |= (unsigned char)((state & 1u) << (7 - (t & 7)));
        state = (state >> 1) | ((unsigned)((d[t] >> state) & 1ull) << 5);
    }
    free(d);
}

/* ------------------------------------------------------------------ rep3 / rep5 / Hamming(7,4) / Hamming(8,4) */
unsigned ll_hamming74_encode_symbol(unsigned s)
{
    unsigned d1 = (s >> 3) & 1, d2 = (s >> 2) & 1, d3 = (s >> 1) & 1, d4 = s & 1;
    unsigned p1 = d1 ^ d2 ^ d4, p2 = d1 ^ d3 ^ d4, p4 = d2 ^ d3 ^ d4;
    return (p1 << 6) | (p2 << 5) | (d1 << 4) | (p4 << 3) | (d2 << 2) | (d3 << 1) | d4;
}
unsigned ll_hamming84_encode_symbol(unsigned s)
{
    unsigned c = ll_hamming74_encode_symbol(s);
    return (c << 1) | par(c);
}
static unsigned hsmall_encode(unsigned s, unsigned nb) { return nb == 7 ? ll_hamming74_encode_symbol(s) : ll_hamming84_encode_symbol(s); }
/* nearest codeword of an nb-bit word, symbols ascending, strict < */
static unsigned hsmall_decode(unsigned w, unsigned nb)
{
    unsigned best = 0, dmin = 99;
    for (unsigned s = 0; s < 16; s++) {
        unsigned d = (unsigned)__builtin_popcount(w ^ hsmall_encode(s, nb));
        if (d < dmin) { dmin = d; best = s; }
    }
    return best;
}
static unsigned hsmall_decode_soft(const unsigned char *soft, unsigned nb)
{
    unsigned best = 0, dmin = 0;
    for (unsigned s = 0; s < 16; s++) {
        unsigned c = hsmall_encode(s, nb), d = 0;
        for (unsigned k = 0; k < nb; k++) d += ((c >> (nb - 1 - k)) & 1) ? 255u - soft[k] : soft[k];
        if (s == 0 || d < dmin) { dmin = d; best = s; }
    }
    return best;
}
static unsigned bits_get(const unsigned char *x, unsigned k, unsigned nb)      /* nb bits from bit index k, MSB first */
{ unsigned v = 0; for (unsigned i = 0; i < nb; i++) v = (v << 1) | ((x[(k + i) >> 3] >> (7 - ((k + i) & 7))) & 1u); return v; }
static void bits_put(unsigned char *x, unsigned k, unsigned nb, unsigned v)
{ for (unsigned i = 0; i < nb; i++) if ((v >> (nb - 1 - i)) & 1u) x[(k + i) >> 3] |= (unsigned char)(0x80u >> ((k + i) & 7)); }
static unsigned rep_copies(int scheme) { return scheme == LL_FEC_REP3 ? 3u : 5u; }

/* ------------------------------------------------------------------ FEC block codecs */
int ll_fec_supported(int scheme)
{ return (scheme >= LL_FEC_NONE && scheme <= LL_FEC_GOLAY2412) || scheme == LL_FEC_CONV_V27; }

unsigned ll_fec_enc_len(int scheme, unsigned n)
{
    switch (scheme) {
    case LL_FEC_CONV_V27:   return 2 * n + 2;
    case LL_FEC_HAMMING128: return (n / 2) * 3 + (n % 2) * 2;
    case LL_FEC_GOLAY2412:  return (n / 3) * 6 + (n % 3) * 3;
    case LL_FEC_REP3:       return 3 * n;
    case LL_FEC_REP5:       return 5 * n;
    case LL_FEC_HAMMING74:  return (14 * n + 7) / 8;        /* liquid fec_block_get_enc_msg_len(n, 4, 7) */
    case LL_FEC_HAMMING84:  return 2 * n;
    default: return n;
    }
}

void ll_fec_encode(int scheme, unsigned n, const unsigned char *dec, unsigned char *enc)
{
    unsigned i, j = 0;
    switch (scheme) {
    case LL_FEC_HAMMING128: {
        unsigned r = n % 2;
        for (i = 0; i < n - r; i += 2) {
            unsigned m0 = ll_hamming128_encode_symbol(dec[i]);
            unsigned m1 = ll_hamming128_encode_symbol(dec[i + 1]);
            enc[j + 0] = (unsigned char)((m0 >> 4) & 0xff);
            enc[j + 1] = (unsigned char)(((m0 << 4) & 0xf0) | ((m1 >> 8) & 0x0f));
            enc[j + 2] = (unsigned char)(m1 & 0xff);
            j += 3;
        }
        if (r) {
            unsigned m0 = ll_hamming128_encode_symbol(dec[n - 1]);
            enc[j + 0] = (unsigned char)((m0 >> 4) & 0xff);
            enc[j + 1] = (unsigned char)((m0 << 4) & 0xf0);
        }
    } break;
    case LL_FEC_GOLAY2412: {
        unsigned r = n % 3;
        for (i = 0; i < n - r; i += 3) {
            unsigned s0 = ((unsigned)dec[i] << 4) | ((unsigned)dec[i + 1] >> 4);
            unsigned s1 = (((unsigned)dec[i + 1] & 0x0f) << 8) | (unsigned)dec[i + 2];
            unsigned m0 = ll_golay2412_encode_symbol(s0), m1 = ll_golay2412_encode_symbol(s1);
            enc[j + 0] = (unsigned char)(m0 >> 16); enc[j + 1] = (unsigned char)(m0 >> 8); enc[j + 2] = (unsigned char)m0;
            enc[j + 3] = (unsigned char)(m1 >> 16); enc[j + 4] = (unsigned char)(m1 >> 8); enc[j + 5] = (unsigned char)m1;
            j += 6;
        }
        for (i = n - r; i < n; i++) {
            unsigned m0 = ll_golay2412_encode_symbol(dec[i]);
            enc[j + 0] = (unsigned char)(m0 >> 16); enc[j + 1] = (unsigned char)(m0 >> 8); enc[j + 2] = (unsigned char)m0;
            j += 3;
        }
    } break;
    case LL_FEC_CONV_V27: conv27_encode(n, dec, enc); break;
    case LL_FEC_REP3: case LL_FEC_REP5:
        for (i = 0; i < rep_copies(scheme); i++) memmove(enc + i * n, dec, n);
        break;
    case LL_FEC_HAMMING74:
        memset(enc, 0, ll_fec_enc_len(scheme, n));
        for (i = 0; i < n; i++) {
            bits_put(enc, 14 * i,     7, ll_hamming74_encode_symbol(dec[i] >> 4));
            bits_put(enc, 14 * i + 7, 7, ll_hamming74_encode_symbol(dec[i] & 0x0f));
        }
        break;
    case LL_FEC_HAMMING84:
        for (i = 0; i < n; i++) {
            enc[2 * i]     = (unsigned char)ll_hamming84_encode_symbol(dec[i] >> 4);
            enc[2 * i + 1] = (unsigned char)ll_hamming84_encode_symbol(dec[i] & 0x0f);
        }
        break;
    default: memmove(enc, dec, n);
    }
}

void ll_fec_decode(int scheme, unsigned n, const unsigned char *enc, unsigned char *dec)
{
    unsigned i, j = 0;
    switch (scheme) {
    case LL_FEC_CONV_V27: {
        const unsigned T = 8 * n + 6;
        unsigned char *sym = (unsigned char *)malloc(2 * T);
        for (i = 0; i < 2 * T; i++) sym[i] = ((enc[i >> 3] >> (7 - (i & 7))) & 1u) ? 255 : 0;
        conv27_viterbi(n, sym, dec);
        free(sym);
    } break;
    case LL_FEC_HAMMING128: {
        unsigned r = n % 2;
        for (i = 0; i < n - r; i += 2) {
            unsigned m0 = ((unsigned)enc[j] << 4) | ((unsigned)enc[j + 1] >> 4);
            unsigned m1 = (((unsigned)enc[j + 1] & 0x0f) << 8) | (unsigned)enc[j + 2];
            dec[i]     = (unsigned char)ll_hamming128_decode_symbol(m0);
            dec[i + 1] = (unsigned char)ll_hamming128_decode_symbol(m1);
            j += 3;
        }
        if (r) {
            unsigned m0 = ((unsigned)enc[j] << 4) | ((unsigned)enc[j + 1] >> 4);
            dec[n - 1] = (unsigned char)ll_hamming128_decode_symbol(m0);
        }
    } break;
    case LL_FEC_GOLAY2412: {
        unsigned r = n % 3;
        for (i = 0; i < n - r; i += 3) {
            unsigned m0 = ((unsigned)enc[j] << 16) | ((unsigned)enc[j + 1] << 8) | enc[j + 2];
            unsigned m1 = ((unsigned)enc[j + 3] << 16) | ((unsigned)enc[j + 4] << 8) | enc[j + 5];
            unsigned s0 = ll_golay2412_decode_symbol(m0), s1 = ll_golay2412_decode_symbol(m1);
            dec[i]     = (unsigned char)((s0 >> 4) & 0xff);
            dec[i + 1] = (unsigned char)(((s0 << 4) & 0xf0) | ((s1 >> 8) & 0x0f));
            dec[i + 2] = (unsigned char)(s1 & 0xff);
            j += 6;
        }
        for (i = n - r; i < n; i++) {
            unsigned m0 = ((unsigned)enc[j] << 16) | ((unsigned)enc[j + 1] << 8) | enc[j + 2];
            dec[i] = (unsigned char)(ll_golay2412_decode_symbol(m0) & 0xff);
            j += 3;
        }
    } break;
    case LL_FEC_REP3:
        for (i = 0; i < n; i++) {
            unsigned s0 = enc[i], s1 = enc[i + n], s2 = enc[i + 2 * n];
            dec[i] = (unsigned char)((s0 & s1) | (s0 & s2) | (s1 & s2));
        }
        break;
    case LL_FEC_REP5:
        for (i = 0; i < n; i++) {
            unsigned b = 0;
            for (unsigned k = 0; k < 8; k++) {
                unsigned cnt = 0;
                for (unsigned r = 0; r < 5; r++) cnt += (enc[i + r * n] >> k) & 1u;
                b |= (cnt >= 3 ? 1u : 0u) << k;
            }
            dec[i] = (unsigned char)b;
        }
        break;
    case LL_FEC_HAMMING74:
        for (i = 0; i < n; i++)
            dec[i] = (unsigned char)((hsmall_decode(bits_get(enc, 14 * i, 7), 7) << 4) | hsmall_decode(bits_get(enc, 14 * i + 7, 7), 7));
        break;
    case LL_FEC_HAMMING84:
        for (i = 0; i < n; i++)
            dec[i] = (unsigned char)((hsmall_decode(enc[2 * i], 8) << 4) | hsmall_decode(enc[2 * i + 1], 8));
        break;
    default: memmove(dec, enc, n);
    }
}

void ll_fec_decode_soft(int scheme, unsigned n, const unsigned char *soft, unsigned char *dec)
{
    if (scheme == LL_FEC_HAMMING128) {
        unsigned r = n % 2, k = 0;       /* k = soft-bit index, 12 per symbol */
        for (unsigned i = 0; i < n - r; i += 2) {
            dec[i]     = (unsigned char)h128_decode_soft_symbol(soft + k);
            dec[i + 1] = (unsigned char)h128_decode_soft_symbol(soft + k + 12);
            k += 24;
        }
        if (r) dec[n - 1] = (unsigned char)h128_decode_soft_symbol(soft + k);
        return;
    }
    if (scheme == LL_FEC_CONV_V27) { conv27_viterbi(n, soft, dec); return; }
    if (scheme == LL_FEC_REP3 || scheme == LL_FEC_REP5) {
        const unsigned R = rep_copies(scheme);
        for (unsigned i = 0; i < n; i++) {
            unsigned b = 0;
            for (unsigned k = 0; k < 8; k++) {
                unsigned sum = 0;
                for (unsigned r = 0; r < R; r++) sum += soft[8 * (i + r * n) + k];
                b = (b << 1) | ((sum / R) > 127 ? 1u : 0u);
            }
            dec[i] = (unsigned char)b;
        }
        return;
    }
    if (scheme == LL_FEC_HAMMING74 || scheme == LL_FEC_HAMMING84) {
        const unsigned nb = scheme == LL_FEC_HAMMING74 ? 7u : 8u;
        for (unsigned i = 0; i < n; i++)
            dec[i] = (unsigned char)((hsmall_decode_soft(soft + 2 * nb * i, nb) << 4) | hsmall_decode_soft(soft + 2 * nb * i + nb, nb));
        return;
    }
    /* no soft decoder: slice at 127, pack MSB first, hard decode */
    unsigned enc_len = ll_fec_enc_len(scheme, n);
    unsigned char *hard = (unsigned char *)malloc(enc_len ? enc_len : 1);
    for (unsigned i = 0; i < enc_len; i++) {
        unsigned b = 0;
        for (unsigned k = 0; k < 8; k++) b = (b << 1) | (soft[8 * i + k] > 127 ? 1u : 0u);
        hard[i] = (unsigned char)b;
    }
    ll_fec_decode(scheme, n, hard, dec);
    free(hard);
}

/* ------------------------------------------------------------------ interleaver */
void ll_interleaver_dims(unsigned n, unsigned *M, unsigned *N)
{
    unsigned m = 1 + (unsigned)floorf(sqrtf((float)n));
    unsigned nn = n / m;
    while (n >= m * nn) nn++;
    *M = m; *N = nn;
}

/* j(i) of one permutation pass: column walk over an M x N grid, skipping j >= n/2 */
static void il_walk(unsigned n, unsigned M, unsigned N, unsigned *jidx)
{
    unsigned n2 = n / 2, m = 0, c = n / 3, j;
    for (unsigned i = 0; i < n2; i++) {
        do {
            j = m * N + c;
            m++;
            if (m == M) { c = (c + 1) % N; m = 0; }
        } while (j >= n2);
        jidx[i] = j;
    }
}
static void il_permute(unsigned char *x, unsigned n, unsigned M, unsigned N, unsigned mask, unsigned *jidx)
{
    il_walk(n, M, N, jidx);
    for (unsigned i = 0; i < n / 2; i++) {
        unsigned a = x[2 * i], b = x[2 * jidx[i] + 1];
        x[2 * i]           = (unsigned char)((a & ~mask) | (b & mask));
        x[2 * jidx[i] + 1] = (unsigned char)((a & mask) | (b & ~mask));
    }
}
static void il_permute_soft(unsigned char *x, unsigned n, unsigned M, unsigned N, unsigned mask, unsigned *jidx)
{
    il_walk(n, M, N, jidx);
    for (unsigned i = 0; i < n / 2; i++) {
        unsigned char *a = x + 8 * (2 * i), *b = x + 8 * (2 * jidx[i] + 1);
        for (unsigned k = 0; k < 8; k++)
            if ((mask >> (7 - k)) & 1) { unsigned char t = a[k]; a[k] = b[k]; b[k] = t; }
    }
}
void ll_interleaver_encode(unsigned n, unsigned depth, const unsigned char *in, unsigned char *out)
{
    unsigned M, N; ll_interleaver_dims(n, &M, &N);
    unsigned *jidx = (unsigned *)malloc(sizeof(unsigned) * (n / 2 + 1));
    memmove(out, in, n);
    if (depth > 0) il_permute(out, n, M, N, 0xff, jidx);
    if (depth > 1) il_permute(out, n, M, N + 2, 0x0f, jidx);
    if (depth > 2) il_permute(out, n, M, N + 4, 0x55, jidx);
    if (depth > 3) il_permute(out, n, M, N + 8, 0x33, jidx);
    free(jidx);
}
void ll_interleaver_decode(unsigned n, unsigned depth, const unsigned char *in, unsigned char *out)
{
    unsigned M, N; ll_interleaver_dims(n, &M, &N);
    unsigned *jidx = (unsigned *)malloc(sizeof(unsigned) * (n / 2 + 1));
    memmove(out, in, n);
    if (depth > 3) il_permute(out, n, M, N + 8, 0x33, jidx);
    if (depth > 2) il_permute(out, n, M, N + 4, 0x55, jidx);
    if (depth > 1) il_permute(out, n, M, N + 2, 0x0f, jidx);
    if (depth > 0) il_permute(out, n, M, N, 0xff, jidx);
    free(jidx);
}
void ll_interleaver_decode_soft(unsigned n, unsigned depth, const unsigned char *in, unsigned char *out)
{
    unsigned M, N; ll_interleaver_dims(n, &M, &N);
    unsigned *jidx = (unsigned *)malloc(sizeof(unsigned) * (n / 2 + 1));
    memmove(out, in, 8 * (size_t)n);
    if (depth > 3) il_permute_soft(out, n, M, N + 8, 0x33, jidx);
    if (depth > 2) il_permute_soft(out, n, M, N + 4, 0x55, jidx);
    if (depth > 1) il_permute_soft(out, n, M, N + 2, 0x0f, jidx);
    if (depth > 0) il_permute_soft(out, n, M, N, 0xff, jidx);
    free(jidx);
}

/* ------------------------------------------------------------------ scrambler / repack */
void ll_scramble(unsigned char *x, unsigned n)
{
    static const unsigned char mask[4] = { 0xb4, 0x6a, 0x8b, 0xc5 };
    for (unsigned i = 0; i < n; i++) x[i] ^= mask[i & 3];
}

void ll_repack_bytes(const unsigned char *in, unsigned in_bps, unsigned in_len,
                     unsigned char *out, unsigned out_bps, unsigned out_len, unsigned *written)
{
    /* MSB-first bit stream; the last output symbol is zero padded */
    unsigned total = in_len * in_bps;
    unsigned need = total / out_bps + ((total % out_bps) ? 1 : 0);
    if (need > out_len) need = out_len;
    unsigned bit = 0;
    for (unsigned o = 0; o < need; o++) {
        unsigned v = 0;
        for (unsigned k = 0; k < out_bps; k++, bit++) {
            unsigned b = 0;
            if (bit < total) b = (in[bit / in_bps] >> (in_bps - 1 - (bit % in_bps))) & 1;
            v = (v << 1) | b;
        }
        out[o] = (unsigned char)v;
    }
    if (written) *written = need;
}

/* ------------------------------------------------------------------ packetizer */
struct ll_packetizer_s {
    unsigned msg_len, packet_len, crc_len;
    int check;
    struct { int fs; unsigned dec_len, enc_len, depth; } plan[2];
    unsigned char *b0, *b1;
};

unsigned ll_packetizer_compute_enc_len(unsigned n, int crc, int fec0, int fec1)
{ return ll_fec_enc_len(fec1, ll_fec_enc_len(fec0, n + ll_crc_length(crc))); }

ll_packetizer ll_packetizer_create(unsigned n, int crc, int fec0, int fec1)
{
    ll_packetizer p = (ll_packetizer)calloc(1, sizeof(*p));
    p->msg_len = n; p->check = crc; p->crc_len = ll_crc_length(crc);
    unsigned n0 = n + p->crc_len;
    for (int i = 0; i < 2; i++) {
        int fs = i ? fec1 : fec0;
        p->plan[i].fs = fs;
        p->plan[i].dec_len = n0;
        p->plan[i].enc_len = ll_fec_enc_len(fs, n0);
        p->plan[i].depth = (fs == LL_FEC_NONE || fs == LL_FEC_UNKNOWN) ? 0 : 4;
        n0 = p->plan[i].enc_len;
    }
    p->packet_len = n0;
    p->b0 = (unsigned char *)malloc(8 * (size_t)n0 + 16);
    p->b1 = (unsigned char *)malloc(8 * (size_t)n0 + 16);
    return p;
}
void ll_packetizer_destroy(ll_packetizer p) { if (!p) return; free(p->b0); free(p->b1); free(p); }
unsigned ll_packetizer_enc_len(ll_packetizer p) { return p->packet_len; }

void ll_packetizer_encode(ll_packetizer p, const unsigned char *msg, unsigned char *pkt)
{
    memmove(p->b0, msg, p->msg_len);
    unsigned key = ll_crc_generate_key(p->check, p->b0, p->msg_len);
    for (unsigned i = 0; i < p->crc_len; i++) { p->b0[p->msg_len + p->crc_len - i - 1] = key & 0xff; key >>= 8; }
    for (int i = 0; i < 2; i++) {
        ll_fec_encode(p->plan[i].fs, p->plan[i].dec_len, p->b0, p->b1);
        ll_interleaver_encode(p->plan[i].enc_len, p->plan[i].depth, p->b1, p->b0);
    }
    memmove(pkt, p->b0, p->packet_len);
}

static int pk_check(ll_packetizer p, unsigned char *msg)
{
    unsigned key = 0;
    for (unsigned i = 0; i < p->crc_len; i++) key = (key << 8) | p->b0[p->msg_len + i];
    memmove(msg, p->b0, p->msg_len);
    if (p->crc_len == 0) return 1;
    return ll_crc_generate_key(p->check, p->b0, p->msg_len) == key;
}
int ll_packetizer_decode(ll_packetizer p, const unsigned char *pkt, unsigned char *msg)
{
    memmove(p->b0, pkt, p->packet_len);
    for (int i = 1; i >= 0; i--) {
        ll_interleaver_decode(p->plan[i].enc_len, p->plan[i].depth, p->b0, p->b1);
        ll_fec_decode(p->plan[i].fs, p->plan[i].dec_len, p->b1, p->b0);
    }
    return pk_check(p, msg);
}
int ll_packetizer_decode_soft(ll_packetizer p, const unsigned char *pkt_soft, unsigned char *msg)
{
    /* outer code (plan 1) soft, inner code (plan 0) hard */
    memmove(p->b0, pkt_soft, 8 * (size_t)p->packet_len);
    ll_interleaver_decode_soft(p->plan[1].enc_len, p->plan[1].depth, p->b0, p->b1);
    ll_fec_decode_soft(p->plan[1].fs, p->plan[1].dec_len, p->b1, p->b0);
    ll_interleaver_decode(p->plan[0].enc_len, p->plan[0].depth, p->b0, p->b1);
    ll_fec_decode(p->plan[0].fs, p->plan[0].dec_len, p->b1, p->b0);
    return pk_check(p, msg);
}
