// survivor merge depth statistics of the K=7 r=1/2 Viterbi decoder (polys 0x6d, 0x4f), 8-bit soft symbols, tie -> predecessor x=0
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
#define RING 1024
static uint64_t rng = 88172645463325252ull;
static inline uint64_t xr(void){ rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; }
static double gauss(void){ double u = (xr() >> 11) * (1.0 / 9007199254740992.0), v = (xr() >> 11) * (1.0 / 9007199254740992.0); if (u < 1e-300) u = 1e-300; return sqrt(-2 * log(u)) * cos(6.283185307179586 * v); }
int main(int argc, char **argv)
{
    long steps = atol(argv[1]); double snr_db = atof(argv[2]);   // snr_db <= -100: pure noise (uniform random bytes)
    int noise_only = snr_db <= -100; double sigma = noise_only ? 0 : pow(10, -snr_db / 20) / sqrt(2.0);
    static uint64_t dec[RING]; int32_t pm[64], nm[64]; memset(pm, 0, sizeof pm);
    static long hist[RING + 1]; unsigned enc = 0;
    for (long t = 0; t < steps; t++) {
        unsigned sa, sb;
        if (noise_only) { sa = xr() & 255; sb = xr() & 255; }
        else {
            unsigned bit = xr() & 1; enc = ((enc << 1) | bit) & 127;
            int ca = __builtin_parity(enc & 0x6d), cb = __builtin_parity(enc & 0x4f);
            double ra = (ca ? 1 : -1) + sigma * gauss(), rb = (cb ? 1 : -1) + sigma * gauss();
            int qa = (int)lrint(127.5 + 127.5 * ra * 0.5), qb = (int)lrint(127.5 + 127.5 * rb * 0.5);       // (a soft demodulator's scaling: +-1 at 64 / 191)
            sa = qa < 0 ? 0 : qa > 255 ? 255 : qa; sb = qb < 0 ? 0 : qb > 255 ? 255 : qb;
        }
        uint64_t w = 0; int32_t mn = 1 << 30;
        for (unsigned n = 0; n < 64; n++) {
            unsigned p0 = n >> 1, p1 = (n >> 1) | 32;
            unsigned r0 = (p0 << 1) | (n & 1), r1 = (p1 << 1) | (n & 1);
            int e0a = __builtin_parity(r0 & 0x6d) ? 255 : 0, e0b = __builtin_parity(r0 & 0x4f) ? 255 : 0;
            int e1a = __builtin_parity(r1 & 0x6d) ? 255 : 0, e1b = __builtin_parity(r1 & 0x4f) ? 255 : 0;
            int32_t m0 = pm[p0] + abs((int)sa - e0a) + abs((int)sb - e0b), m1 = pm[p1] + abs((int)sa - e1a) + abs((int)sb - e1b);
            int tk = m1 < m0; nm[n] = tk ? m1 : m0; if (tk) w |= 1ull << n; if (nm[n] < mn) mn = nm[n];
        }
        for (int n = 0; n < 64; n++) pm[n] = nm[n] - mn;
        dec[t % RING] = w;
        if (t >= RING) {
            uint64_t S = ~0ull; int d = 0;
            while (__builtin_popcountll(S) > 1 && d < RING - 1) {
                uint64_t wd = dec[(t - d) % RING], P = 0;
                for (uint64_t m = S; m; m &= m - 1) { unsigned n = __builtin_ctzll(m); P |= 1ull << ((n >> 1) | (((wd >> n) & 1) << 5)); }
                S = P; d++;
            }
            hist[d]++;
        }
    }
    long tot = 0, cum = 0; int mx = 0; for (int d = 0; d <= RING; d++) { tot += hist[d]; if (hist[d]) mx = d; }
    printf("snr %g dB  steps %ld  max merge depth %d\n", snr_db, tot, mx);
    for (int d = RING; d >= 0; d--) { cum += hist[d]; if (d % 8 == 0 && d <= mx + 8 && d >= 16) printf("  P(depth >= %3d) = %.3e\n", d, (double)cum / tot); }
    return 0;
}
