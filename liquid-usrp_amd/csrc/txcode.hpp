// txcode.hpp -- host-side bit-level frame assembly for the GPU transmitter (product code).
//
// What ofdmflexframegen_assemble() does before any sample exists (reference call sites:
// lib/multichanneltx.cc:184-188; traffic recipe src/multichannel_tx.cc:166-190):
//   header  : 8 user bytes + [104, len_hi, len_lo, mod, (check&7)<<5 | fec0, fec1]
//             -> CRC-32 -> Golay(24,12) -> interleave -> scramble -> 288 BPSK symbols
//   payload : bytes -> CRC-32 -> fec0 -> interleave -> fec1 -> interleave -> bps-bit symbols
// plus the filler symbols of the last header / payload OFDM symbol (fixed LCG).  A frame is a
// few kilobytes, so this runs on the host; everything per sample runs in txgen.hip.
#pragma once
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include "design.hpp"

namespace mcrx {

inline uint32_t crc32_bytes(const uint8_t *p, size_t n)
{
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) {
        c ^= p[i];
        for (int j = 0; j < 8; j++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1)));
    }
    return ~c;
}
inline unsigned golay_encode(unsigned m)
{
    m &= 0xfff;
    unsigned par = 0;
    for (unsigned i = 0; i < 12; i++) if (m & (1u << (11 - i))) par ^= golay_P[i];
    return (par << 12) | m;
}
// liquid fec_hamming74.c / fec_hamming84.c: p1 p2 d1 p4 d2 d3 d4 (MSB first) [+ overall parity as the LSB]
inline unsigned hamming74_encode(unsigned s)
{
    const unsigned d1 = (s >> 3) & 1, d2 = (s >> 2) & 1, d3 = (s >> 1) & 1, d4 = s & 1;
    return ((d1 ^ d2 ^ d4) << 6) | ((d1 ^ d3 ^ d4) << 5) | (d1 << 4) | ((d2 ^ d3 ^ d4) << 3) | (d2 << 2) | (d3 << 1) | d4;
}
inline unsigned hamming84_encode(unsigned s)
{ const unsigned c = hamming74_encode(s); unsigned p = c; p ^= p >> 4; p ^= p >> 2; p ^= p >> 1; return (c << 1) | (p & 1u); }
inline void fec_encode(int fs, const std::vector<uint8_t> &dec, std::vector<uint8_t> &enc)
{
    const size_t n = dec.size();
    enc.assign(fec_enc_len(fs, (unsigned)n), 0);
    size_t j = 0;
    if (fs == FEC_HAMMING128) {
        const size_t r = n % 2;
        for (size_t i = 0; i < n - r; i += 2) {
            unsigned m0 = hamming128_encode(dec[i]), m1 = hamming128_encode(dec[i + 1]);
            enc[j] = (uint8_t)(m0 >> 4); enc[j + 1] = (uint8_t)(((m0 << 4) & 0xf0) | ((m1 >> 8) & 0x0f)); enc[j + 2] = (uint8_t)m1;
            j += 3;
        }
        if (r) { unsigned m0 = hamming128_encode(dec[n - 1]); enc[j] = (uint8_t)(m0 >> 4); enc[j + 1] = (uint8_t)((m0 << 4) & 0xf0); }
    } else if (fs == FEC_GOLAY2412) {
        const size_t r = n % 3;
        for (size_t i = 0; i < n - r; i += 3) {
            unsigned s0 = ((unsigned)dec[i] << 4) | (dec[i + 1] >> 4);
            unsigned s1 = (((unsigned)dec[i + 1] & 0x0f) << 8) | dec[i + 2];
            unsigned m0 = golay_encode(s0), m1 = golay_encode(s1);
            enc[j] = (uint8_t)(m0 >> 16); enc[j + 1] = (uint8_t)(m0 >> 8); enc[j + 2] = (uint8_t)m0;
            enc[j + 3] = (uint8_t)(m1 >> 16); enc[j + 4] = (uint8_t)(m1 >> 8); enc[j + 5] = (uint8_t)m1;
            j += 6;
        }
        for (size_t i = n - r; i < n; i++) {
            unsigned m0 = golay_encode(dec[i]);
            enc[j] = (uint8_t)(m0 >> 16); enc[j + 1] = (uint8_t)(m0 >> 8); enc[j + 2] = (uint8_t)m0;
            j += 3;
        }
    } else if (fs == FEC_CONV_V27) {
        // liquid fec_conv.c / libfec: polynomials 0x6d, 0x4f on sr = (sr << 1) | bit, message bits MSB first, six zero tail bits
        auto par = [](unsigned v) { v ^= v >> 4; v ^= v >> 2; v ^= v >> 1; return v & 1u; };
        unsigned sr = 0, acc = 0, nb = 0;
        for (size_t i = 0; i < 8 * n + 6; i++) {
            const unsigned bit = i < 8 * n ? (dec[i >> 3] >> (7 - (i & 7))) & 1u : 0u;
            sr = ((sr << 1) | bit) & 0x7f;
            acc = (acc << 1) | par(sr & 0x6d); acc = (acc << 1) | par(sr & 0x4f); nb += 2;
            if (nb == 8) { enc[j++] = (uint8_t)acc; acc = 0; nb = 0; }
        }
        if (nb) enc[j++] = (uint8_t)(acc << (8 - nb));
    } else if (fs == FEC_REP3 || fs == FEC_REP5) {
        // liquid fec_rep3.c / fec_rep5.c: the whole message, three / five times in a row
        for (size_t r = 0; r < (fs == FEC_REP3 ? 3u : 5u); r++) std::memcpy(enc.data() + r * n, dec.data(), n);
    } else if (fs == FEC_HAMMING74) {
        // two 7-bit symbols per byte (high nibble first), bit-packed back to back
        auto put = [&](size_t k, unsigned v) { for (unsigned i = 0; i < 7; i++) if ((v >> (6 - i)) & 1u) enc[(k + i) >> 3] |= (uint8_t)(0x80u >> ((k + i) & 7)); };
        for (size_t i = 0; i < n; i++) { put(14 * i, hamming74_encode(dec[i] >> 4)); put(14 * i + 7, hamming74_encode(dec[i] & 0x0f)); }
    } else if (fs == FEC_HAMMING84) {
        for (size_t i = 0; i < n; i++) { enc[2 * i] = (uint8_t)hamming84_encode(dec[i] >> 4); enc[2 * i + 1] = (uint8_t)hamming84_encode(dec[i] & 0x0f); }
    } else enc = dec;
}
// liquid's interleaver: pass = swap masked bits of x[2i] and x[2j+1], j(i) from the column walk
inline void il_pass_host(std::vector<uint8_t> &x, unsigned Mi, unsigned Ncol, unsigned mask)
{
    const unsigned n = (unsigned)x.size(), n2 = n / 2;
    unsigned m = 0, c = n / 3;
    for (unsigned i = 0; i < n2; i++) {
        unsigned j;
        do { j = m * Ncol + c; m++; if (m == Mi) { c = (c + 1) % Ncol; m = 0; } } while (j >= n2);
        const unsigned a = x[2 * i], b = x[2 * j + 1];
        x[2 * i] = (uint8_t)((a & ~mask) | (b & mask));
        x[2 * j + 1] = (uint8_t)((a & mask) | (b & ~mask));
    }
}
inline void interleave(std::vector<uint8_t> &x, unsigned depth)
{
    const unsigned n = (unsigned)x.size();
    if (n < 2 || depth == 0) return;
    unsigned Mi = 1 + (unsigned)std::floor(std::sqrt((float)n)), Ni = n / Mi;
    while (n >= Mi * Ni) Ni++;
    if (depth > 0) il_pass_host(x, Mi, Ni, 0xff);
    if (depth > 1) il_pass_host(x, Mi, Ni + 2, 0x0f);
    if (depth > 2) il_pass_host(x, Mi, Ni + 4, 0x55);
    if (depth > 3) il_pass_host(x, Mi, Ni + 8, 0x33);
}
inline void packet_encode(const std::vector<uint8_t> &msg, int crc, int fec0, int fec1, std::vector<uint8_t> &pkt)
{
    std::vector<uint8_t> b0 = msg, b1;
    if (crc == CRC_32) {
        uint32_t key = crc32_bytes(msg.data(), msg.size());
        b0.push_back((uint8_t)(key >> 24)); b0.push_back((uint8_t)(key >> 16)); b0.push_back((uint8_t)(key >> 8)); b0.push_back((uint8_t)key);
    }
    fec_encode(fec0, b0, b1); interleave(b1, fec_il_depth(fec0));
    fec_encode(fec1, b1, b0); interleave(b0, fec_il_depth(fec1));
    pkt.swap(b0);
}

// modem symbols of one frame, per OFDM data symbol: hdr [S_hdr * M_data] (1 bit), pay [S_pay * M_data] (bps bits)
struct FrameSymbols { std::vector<uint8_t> hdr, pay; };

inline void assemble_frame(const uint8_t header8[8], const std::vector<uint8_t> &payload, int mod, int fec0, int fec1,
                           unsigned M_data, unsigned S_hdr, unsigned S_pay, FrameSymbols &out)
{
    const unsigned bps = mod_bps(mod);
    std::vector<uint8_t> h(14), henc;
    std::memcpy(h.data(), header8, 8);
    h[8] = 104; h[9] = (uint8_t)(payload.size() >> 8); h[10] = (uint8_t)payload.size();
    h[11] = (uint8_t)mod; h[12] = (uint8_t)(((CRC_32 & 7) << 5) | (fec0 & 0x1f)); h[13] = (uint8_t)(fec1 & 0x1f);
    packet_encode(h, CRC_32, FEC_GOLAY2412, FEC_NONE, henc);            // 36 bytes
    static const uint8_t smask[4] = { 0xb4, 0x6a, 0x8b, 0xc5 };
    for (size_t i = 0; i < henc.size(); i++) henc[i] ^= smask[i & 3];
    std::vector<uint8_t> penc;
    packet_encode(payload, CRC_32, fec0, fec1, penc);
    uint32_t lcg = 0x1234567u;                                          // filler generator, reset per frame
    auto filler = [&](unsigned b) { lcg = lcg * 1664525u + 1013904223u; return (uint8_t)((lcg >> 16) & ((1u << b) - 1)); };
    out.hdr.assign((size_t)S_hdr * M_data, 0);
    for (size_t i = 0; i < out.hdr.size(); i++)
        out.hdr[i] = (i < 288) ? (uint8_t)((henc[i / 8] >> (7 - (i % 8))) & 1) : filler(1);
    const size_t nbits = 8 * penc.size(), mod_len = nbits / bps + ((nbits % bps) ? 1 : 0);
    out.pay.assign((size_t)S_pay * M_data, 0);
    for (size_t i = 0; i < out.pay.size(); i++) {
        if (i < mod_len) {
            unsigned v = 0;
            for (unsigned k = 0; k < bps; k++) {
                const size_t bit = i * bps + k;
                v = (v << 1) | ((bit < nbits) ? ((penc[bit / 8] >> (7 - (bit % 8))) & 1u) : 0u);
            }
            out.pay[i] = (uint8_t)v;
        } else out.pay[i] = filler(bps);
    }
}

}  // namespace mcrx
