// demod_pk.hpp -- the soft demodulator instantiated per modem, for the lean payload workers (payload_lean.hpp: 48 / 64 subcarriers;
// payload_wide.hpp: 128 / 256).  Included inside namespace mcrx behind demod_soft (ofdmsync.hip).
#pragma once
#include "lean_prims.hpp"
namespace lean {

// soft bits of one symbol, byte k = bit k (most significant first), and the hard symbol.  Same expressions as demod_soft
template <int MOD>
__device__ __forceinline__ uint64_t demod_pk(const uint8_t *nbt, v2f r, bool soft_mode)
{
    constexpr unsigned bps = MOD == 39 ? 1u : MOD == 40 ? 2u : MOD == 27 ? 4u : 6u;
    if constexpr (MOD == 39) {
        if (soft_mode) return soft_clamp((-2.0f * r.x * 4.0f) * 16.0f + 127.0f);
        return r.x > 0 ? 0u : 255u;
    } else if constexpr (MOD == 40) {
        if (soft_mode) return (uint32_t)soft_clamp((-2.0f * r.y * 5.8f) * 16.0f + 127.0f) | ((uint32_t)soft_clamp((-2.0f * r.x * 5.8f) * 16.0f + 127.0f) << 8);
        return (r.y > 0 ? 0u : 255u) | (r.x > 0 ? 0u : 0xff00u);
    } else {
        uint8_t sb[6];
        const unsigned hs = demod_soft(nbt, (unsigned)MOD, make_float2(r.x, r.y), sb);
        uint64_t w = 0;
#pragma unroll
        for (unsigned kb = 0; kb < bps; kb++) w |= (uint64_t)(soft_mode ? (unsigned)sb[kb] : (((hs >> (bps - 1 - kb)) & 1) ? 255u : 0u)) << (8 * kb);
        return w;
    }
}

template <int MOD> struct bits_per_symbol { static constexpr unsigned v = MOD == 39 ? 1u : MOD == 40 ? 2u : MOD == 27 ? 4u : 6u; };

}  // namespace lean
