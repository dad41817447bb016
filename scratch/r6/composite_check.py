"""Round 6: the oversampled front end (NCO -> firpfbch2 analysis, 2N channels -> half-band decimator per kept channel)
is ONE critically sampled polyphase bank: a 28-tap-per-column composite prototype, a forward FFT, and a rotation of the
columns by M/2 + 1.  Checks the algebra in float64 numpy against the oracle's stage-by-stage chain (float32)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as O
O.build()


def composite_taps(N):
    """G[n][d]: out_k[c] = FFT_fwd( V_k[(n - s) mod M] )[c],  V_k[n] = sum_d G[n][d] u[(k-d) M + n],  s = M/2 + 1"""
    M = 2 * N
    h = O.Channelizer2(M, 7).taps().astype(np.float64)           # firpfbch2 prototype, 14 M taps
    hh = O.firdes_kaiser(29, 0.25, 60.0)
    h1 = np.array([hh[29 - i - 1] for i in range(1, 29, 2)], np.float64)
    G = np.zeros((M, 28))
    for r in range(M):
        n = (M // 2 - 1 - r) % M
        up = 1 if r >= M // 2 else 0
        for i in range(14):
            for j in range(14):
                G[n, 13 - i + j + up] += h1[i] * h[r + j * M]      # even steps through the half-band branch filter
        rp = (r - M // 2) % M
        for j in range(14):
            G[n, 7 + j] += h[rp + j * M]                           # the odd step 2k - 13 through the delay branch
    return G * (0.5 / M)


def run(N, nblocks=96, seed=1):
    M = 2 * N
    rng = np.random.RandomState(seed)
    x = (rng.randn(nblocks * M) + 1j * rng.randn(nblocks * M)).astype(np.complex64)
    rx = O.MultiChannelRx(N, 64, 8, 4)
    want = rx.channelize_oversampled(x)                           # [nblocks][N]
    G = composite_taps(N)
    f = np.float32(-0.5 * np.float32(N - 1) / np.float32(N))
    p = (float(np.float32(float(f) * np.pi)) / (2 * np.pi)) % 1.0
    dth = int(np.rint(p * 2 ** 32)) & 0xFFFFFFFF
    t = np.arange(len(x), dtype=np.uint64)
    ph = ((t * np.uint64(dth)) & np.uint64(0xFFFFFFFF)).astype(np.float64) * (2 * np.pi / 2 ** 32)
    u = (x.astype(np.complex128) * np.exp(-1j * ph)).reshape(nblocks, M)
    up = np.concatenate([np.zeros((27, M), np.complex128), u])
    V = np.zeros((nblocks, M), np.complex128)
    for d in range(28):
        V += G[None, :, d] * up[27 - d:27 - d + nblocks]
    s = M // 2 + 1
    Vs = np.roll(V, s, axis=1)                                    # Vs[n] = V[(n - s) mod M]
    got = np.fft.fft(Vs, axis=1)[:, :N]
    err = np.max(np.abs(got - want)) / np.max(np.abs(want))
    print("N=%d: composite vs oracle chain, max rel err %.3g (zero taps: d=0 %d cols, d=27 %d cols)" %
          (N, err, int(np.sum(G[:, 0] == 0)), int(np.sum(G[:, 27] == 0))))
    return err


if __name__ == "__main__":
    for N in (1, 2, 4, 8, 32, 64):
        assert run(N) < 3e-6


def tap_profile(N=512):
    G = composite_taps(N)
    a = np.abs(G)
    tot = a.sum(axis=1).max()
    print("N=%d: per-d max |G| relative to the largest tap, and the worst-case column sum of the taps beyond d (relative to the column's tap sum)" % N)
    gmax = a.max()
    for d in range(28):
        print("  d=%2d  max %.3e   sum over columns' worst |tap| share %.3e" % (d, a[:, d].max() / gmax, (a[:, d] / a.sum(axis=1)).max()))
    for lo, hi in ((1, 26), (2, 25), (3, 24), (4, 23)):
        drop = np.concatenate([a[:, :lo], a[:, hi + 1:]], axis=1).sum(axis=1)
        print("  keep d in [%d, %d] (%d taps): dropped tap mass / kept, worst column %.3e" % (lo, hi, hi - lo + 1, (drop / a.sum(axis=1)).max()))
