"""Pins the CPU oracle's DSP pieces against independent float64 numpy/scipy models and
round-trip invariants."""
import ctypes as C

import numpy as np
import pytest
import scipy.signal
import scipy.special


@pytest.mark.parametrize("n", [2, 6, 16, 48, 64, 100, 128, 1024])
def test_fft_matches_numpy(oracle, n):
    rng = np.random.RandomState(n)
    x = (rng.randn(n) + 1j * rng.randn(n)).astype(np.complex64)
    ref = np.fft.fft(x.astype(np.complex128))
    assert np.max(np.abs(oracle.fft(x) - ref)) < 3e-6 * np.max(np.abs(ref))
    refb = np.fft.ifft(x.astype(np.complex128)) * n
    assert np.max(np.abs(oracle.fft(x, backward=True) - refb)) < 3e-6 * np.max(np.abs(refb))


@pytest.mark.parametrize("K,m", [(2, 7), (16, 7), (128, 7), (1024, 7), (16, 13)])
def test_kaiser_prototype_matches_float64_model(oracle, K, m):
    n = 2 * K * m + 1
    h = oracle.firdes_kaiser(n, 0.5 / K, 60.0)
    beta = 0.1102 * (60.0 - 8.7)
    t = np.arange(n) - (n - 1) / 2.0
    r = 2.0 * t / n                                     # liquid's kaiser(): N, not N-1
    w = scipy.special.i0(beta * np.sqrt(1 - r * r)) / scipy.special.i0(beta)
    ref = np.sinc(2 * (0.5 / K) * t) * w
    assert np.max(np.abs(h - ref)) < 2e-7
    assert abs(h[(n - 1) // 2] - 1.0) < 1e-6 and np.allclose(h, h[::-1], atol=1e-7)


@pytest.mark.parametrize("K", [2, 6, 16, 128])
def test_analysis_bank_equals_mix_filter_decimate(oracle, K):
    """Polyphase identity: bin k of block b == sum_t h[T-t] x[t] e^{-j 2 pi k t / K}, T = bK+K-1."""
    m = 7
    ch = oracle.Channelizer(oracle.ANALYZER, K, m)
    h = ch.taps().astype(np.float64)
    assert len(h) == 2 * m * K
    nb = 2 * m + 6
    rng = np.random.RandomState(K)
    x = (rng.randn(nb * K) + 1j * rng.randn(nb * K)).astype(np.complex64)
    y = ch.analyze(x)
    xd = x.astype(np.complex128)
    t = np.arange(len(x))
    for b in (0, 3, 2 * m - 1, nb - 1):
        T = b * K + K - 1
        for k in sorted({0, 1, K // 2, K - 1}):
            tt = np.arange(max(0, T - len(h) + 1), T + 1)
            ref = np.sum(h[T - tt] * xd[tt] * np.exp(-2j * np.pi * k * tt / K))
            assert abs(y[b, k] - ref) < 2e-5 * np.sqrt(K), (b, k)


def test_tone_lands_in_one_bin_and_synthesis_analysis_delay(oracle):
    K, N = 16, 8
    syn = oracle.Channelizer(oracle.SYNTHESIZER, K, 13)
    ana = oracle.Channelizer(oracle.ANALYZER, K, 7)
    nb = 80
    X = np.zeros((nb, K), np.complex64)
    X[:, 3] = 1.0                                        # constant on channel 3
    y = ana.analyze(syn.synthesize(X))
    tail = y[40:]
    p = np.mean(np.abs(tail) ** 2, axis=0)
    assert np.argmax(p) == 3 and p[3] > 1e4 * np.delete(p, 3).max()
    # impulse on a channel comes out (m_tx + m_rx) blocks later (minus one: both banks are causal)
    X = np.zeros((nb, K), np.complex64)
    X[5, 2] = 1.0
    y = ana.analyze(oracle.Channelizer(oracle.SYNTHESIZER, K, 13).synthesize(X))
    assert abs(int(np.argmax(np.abs(y[:, 2]))) - 5 - 20) <= 1


def test_nco_phase_accumulator(oracle):
    L = oracle.lib()
    assert L.ll_nco_rad2u32(0.0) == 0
    assert abs(int(L.ll_nco_rad2u32(np.float32(np.pi))) - 0x80000000) < 128      # float32(pi) > pi
    assert abs(int(L.ll_nco_rad2u32(np.float32(-np.pi / 2))) - 0xC0000000) < 64
    s, c = C.c_float(), C.c_float()
    for u in (0, 1 << 30, 1 << 31, 3 << 30, 12345678):
        L.ll_nco_sincos_u32(u, C.byref(s), C.byref(c))
        a = u * 2 * np.pi / 2 ** 32
        assert abs(s.value - np.sin(a)) < 1e-7 and abs(c.value - np.cos(a)) < 1e-7


@pytest.mark.parametrize("M", [48, 64, 256])
def test_subcarrier_allocation_and_training_symbols(oracle, M):
    p = oracle.default_sctype(M)
    counts = {64: (14, 6, 44), 256: (52, 26, 178), 48: (10, 4, 34)}[M]
    assert (int((p == 0).sum()), int((p == 1).sum()), int((p == 2).sum())) == counts
    assert p[0] == 0 and all(p[i] == p[M - i] for i in range(1, M // 2))
    S = oracle.init_S0S1(p)
    S0, s0, m0 = S["S0"]
    S1, s1, m1 = S["S1"]
    assert m0 == int(np.count_nonzero(S0)) and m1 == counts[1] + counts[2]
    assert np.all(S0[1::2] == 0) and set(np.unique(S0.real)) <= {-1.0, 0.0, 1.0}
    # s0 has two identical halves (only even bins), unit average power
    assert np.allclose(s0[:M // 2], s0[M // 2:], atol=1e-6)
    assert abs(np.mean(np.abs(s0) ** 2) * M / M - 1.0 * 1.0) < 1e-5 or abs(np.sum(np.abs(s0) ** 2) - M) < 1e-3
    assert abs(np.sum(np.abs(s1) ** 2) - M) < 1e-3
    assert np.max(np.abs(np.fft.ifft(S1.astype(np.complex128)) * M / np.sqrt(m1) - s1)) < 1e-5


@pytest.mark.parametrize("M", [48, 64, 256])
def test_lsq_projections_match_polyfit(oracle, M):
    p = oracle.default_sctype(M)
    S = oracle.eq_smoother(p, 4).astype(np.float64)
    k = (np.arange(M) + M // 2) % M
    en = k[p[k] != 0]
    x = np.where(en > M // 2, en - M, en) / M
    rng = np.random.RandomState(M)
    y = rng.randn(len(en))
    coef = np.polyfit(x, y, 4)
    f = np.where(np.arange(M) > M // 2, np.arange(M) - M, np.arange(M)) / M
    ref = np.where(p != 0, np.polyval(coef, f), 0.0)
    assert np.max(np.abs(S @ y - ref)) < 2e-5 * np.max(np.abs(ref))
    P = oracle.pilot_fit(p).astype(np.float64)
    pk = k[p[k] == 1]
    xp = np.where(pk > M // 2, pk - M, pk).astype(np.float64)
    yp = rng.randn(len(pk))
    c1, c0 = np.polyfit(xp, yp, 1)
    assert abs(P[0] @ yp - c0) < 1e-5 and abs(P[1] @ yp - c1) < 1e-6


@pytest.mark.parametrize("mod,fec1,plen", [(40, 6, 1200), (27, 7, 500), (39, 1, 37), (29, 6, 64), (40, 6, 0)])
def test_flexframe_loopback_with_offsets(oracle, mod, fec1, plen):
    M, cp, tp = 64, 8, 4
    fg = oracle.FlexFrameGen(M, cp, tp, fec1=fec1, mod=mod)
    rng = np.random.RandomState(plen + mod)
    hdr = bytes(rng.randint(0, 256, 8).astype(np.uint8))
    pl = bytes(rng.randint(0, 256, plen).astype(np.uint8))
    x = fg.frame(hdr, pl)
    n = np.arange(len(x) + 700)
    sig = np.concatenate([np.zeros(233, np.complex64), x, np.zeros(467, np.complex64)])
    sig = sig * 0.3 * np.exp(1j * (0.7 + 0.004 * n))                   # gain, phase, CFO
    sig = sig + 0.003 * (rng.randn(len(sig)) + 1j * rng.randn(len(sig)))  # ~37 dB SNR
    for soft in (True, False):
        fs = oracle.FlexFrameSync(M, cp, tp, soft=soft)
        fs.execute(sig.astype(np.complex64))
        assert len(fs.frames) == 1
        f = fs.frames[0]
        assert f.header_valid and f.payload_valid and f.header == hdr and f.payload == pl
        assert f.mod_scheme == mod and f.fec1 == fec1 and f.check == oracle.CRC_32
        assert abs(f.cfo - 0.004 / (2 * np.pi)) < 2e-5
        assert abs(f.rssi - 20 * np.log10(0.3)) < 0.5 and f.evm < -25
        assert len(f.framesyms) == -(-8 * oracle.Packetizer(plen, oracle.CRC_32, 1, fec1).enc_len // f.mod_bps)


def test_sample_at_a_time_equals_bulk(oracle):
    """Execute(buf, n) must equal n calls with one sample (reference app: src/multichannel_rx.cc:211)."""
    N, M, cp, tp = 2, 64, 8, 4
    iq, sent = oracle.synth_traffic(N, M, cp, tp, 1, payload_len=60)
    a = oracle.MultiChannelRx(N, M, cp, tp)
    a.execute(iq)
    b = oracle.MultiChannelRx(N, M, cp, tp)
    for i in range(0, len(iq), 7):
        b.execute(iq[i:i + 7])
    assert len(a.frames) == len(b.frames) == 2
    for fa, fb in zip(a.frames, b.frames):
        assert fa.payload == fb.payload and fa.header == fb.header and fa.evm == fb.evm
        assert np.array_equal(fa.framesyms, fb.framesyms)


@pytest.mark.parametrize("N,M,cp", [(1, 64, 8), (8, 64, 8), (3, 48, 6)])
def test_multichannel_loopback_recovers_every_payload(oracle, N, M, cp):
    iq, sent = oracle.synth_traffic(N, M, cp, 4, 2, payload_len=200)
    rx = oracle.MultiChannelRx(N, M, cp, 4)
    rx.execute(iq)
    assert len(rx.frames) == 2 * N
    for f in rx.frames:
        assert f.header_valid and f.payload_valid and f.header[2] == f.channel
        pid = (f.header[0] << 8) | f.header[1]
        assert sent[f.channel][pid] == (f.header, f.payload)
    # callbacks are ordered by frame end time, ties by channel index
    # (lib/multichannelrx.cc:193-194)
    chans = [f.channel for f in rx.frames]
    assert chans[:N] == sorted(chans[:N])


def test_constructor_argument_validation(oracle):
    for args in [(0, 64, 8, 4), (2, 7, 8, 4), (2, 64, 0, 0), (2, 64, 4, 5)]:   # lib/multichannelrx.cc:54-66
        with pytest.raises(ValueError):
            oracle.MultiChannelRx(*args)
        with pytest.raises(ValueError):
            oracle.MultiChannelTx(*args)


def test_msresamp_half_rate_and_arbitrary(oracle):
    L = oracle.lib()
    for rate in (0.5, 0.37, 0.2):
        q = L.ll_msresamp_create(rate, 60.0)
        n = 4000
        t = np.arange(n)
        f_in = 0.04
        x = np.exp(2j * np.pi * f_in * t).astype(np.complex64)
        y = np.zeros(n + 64, np.complex64)
        ny = C.c_uint(0)
        L.ll_msresamp_execute(q, x.ctypes.data, n, y.ctypes.data, C.byref(ny))
        assert abs(ny.value - rate * n) <= 2
        yy = y[200:ny.value]
        # output is a unit tone at f_in / rate
        ph = np.angle(yy[1:] * np.conj(yy[:-1]))
        assert abs(np.median(ph) / (2 * np.pi) - f_in / rate) < 1e-4
        assert abs(np.mean(np.abs(yy)) - 1.0) < 2e-2
        L.ll_msresamp_destroy(q)


@pytest.mark.parametrize("rate", [2.0, 1.5, 4.0, 6.3])
def test_msresamp_interpolating(oracle, rate):
    """rate > 1 (the transmit side's msresamp_crcf_create(2.0, 60), src/flexframe_tx.cc:170): a tone comes out as a
    unit tone at f / rate with rate * n samples, images suppressed; and interpolating then decimating by the same
    factor gives the input back (delayed)."""
    n, f_in = 4000, 0.11
    x = np.exp(2j * np.pi * f_in * np.arange(n)).astype(np.complex64)
    y = oracle.MsResamp(rate).execute(x)
    assert abs(len(y) - rate * n) <= 8
    yy = y[int(200 * rate):]
    ph = np.angle(yy[1:] * np.conj(yy[:-1]))
    assert abs(np.median(ph) / (2 * np.pi) - f_in / rate) < 1e-4
    assert abs(np.mean(np.abs(yy)) - 1.0) < 2e-2
    spec = np.abs(np.fft.fft(yy[:2048] * np.hanning(2048)))
    k0 = int(round(f_in / rate * 2048))
    img = np.delete(spec, np.arange(k0 - 6, k0 + 7) % 2048)
    assert 20 * np.log10(img.max() / spec.max()) < -50            # images of the zero-order hold are gone
    back = oracle.MsResamp(1.0 / rate).execute(y)
    d = int(np.argmax(np.abs(np.correlate(back[300:900], x[250:800], "valid")))) + 50
    seg = back[300:900][:]
    ref = x[300 - d:900 - d]
    assert np.max(np.abs(seg - ref * (seg @ np.conj(ref)) / (ref @ np.conj(ref)))) < 2e-2


@pytest.mark.parametrize("M,m", [(8, 2), (16, 4), (64, 7)])
def test_oversampled_bank_equals_float64_direct_form(oracle, M, m):
    """firpfbch2 analyzer restatement (two-phase window shuffle) against an independent float64 model:
    y_s[n] = (-1)^(n s) / M * sum_t h[t] e^{j 2 pi n t / M} u[(s+1) M/2 - 1 - t] -- a bank of band-pass filters
    centred on n/M, sampled every M/2 inputs -- and against the float64 prototype (Kaiser, fc = 1/M, sum = M)."""
    ch = oracle.Channelizer2(M, m)
    h = ch.taps().astype(np.float64)
    n_h = 2 * M * m + 1
    beta = 0.1102 * (60.0 - 8.7)
    t = np.arange(n_h) - (n_h - 1) / 2.0
    proto = np.sinc(2.0 / M * t) * scipy.special.i0(beta * np.sqrt(np.maximum(0.0, 1.0 - (2.0 * t / n_h) ** 2))) / scipy.special.i0(beta)
    proto = proto * M / proto.sum()
    assert np.max(np.abs(h - proto[:2 * M * m])) < 2e-6 and abs(h.sum() - M) < 1e-3
    rng = np.random.RandomState(M)
    ns = 60
    x = (rng.randn(ns * M // 2) + 1j * rng.randn(ns * M // 2)).astype(np.complex64)
    y = ch.analyze(x[:len(x) // 2])
    y = np.concatenate([y, ch.analyze(x[len(x) // 2:])])                # state carries over between calls
    u = np.concatenate([np.zeros(len(h), np.complex128), x.astype(np.complex128)])
    tt = np.arange(len(h))
    E = np.exp(2j * np.pi * np.outer(np.arange(M), tt) / M)            # [n][t]
    worst = 0.0
    for s in range(ns):
        seg = u[(s + 1) * (M // 2) - 1 + len(h) - tt]
        d = (E * (h * seg)).sum(axis=1) / M * ((-1.0) ** (np.arange(M) * s))
        worst = max(worst, np.max(np.abs(d - y[s])))
    assert worst < 2e-6, worst
    ch.reset()
    assert np.array_equal(ch.analyze(x), y)                             # reset returns to the cold start


def test_oversampled_bank_tone_and_oversampling(oracle):
    """Prototype cut-off 1/M puts the -6 dB point on the neighbouring channel centres: a tone on channel k's
    centre comes out of k at 0 dB, of k+-1 at -6 dB and of nothing else (< -75 dB); a tone half a spacing away is
    passed at full level by both nearer channels -- what the 2x oversampling buys over the critically sampled bank."""
    M, m, k = 32, 6, 5
    ch = oracle.Channelizer2(M, m)
    n = np.arange(400 * M // 2)
    y = ch.analyze(np.exp(2j * np.pi * k * n / M).astype(np.complex64))[4 * m:]
    p = 10 * np.log10(np.mean(np.abs(y) ** 2, axis=0) + 1e-30)
    assert abs(p[k]) < 0.01 and abs(p[k - 1] + 6.02) < 0.05 and abs(p[k + 1] + 6.02) < 0.05
    assert np.max(np.delete(p, [k - 1, k, k + 1])) < -75.0
    assert np.max(np.abs(np.abs(y[:, k]) - 1.0)) < 1e-3                 # constant envelope: no step-to-step sign error
    ch.reset()
    y = ch.analyze(np.exp(2j * np.pi * (k + 0.5) * n / M).astype(np.complex64))[4 * m:]
    p = 10 * np.log10(np.mean(np.abs(y) ** 2, axis=0) + 1e-30)
    assert abs(p[k]) < 0.05 and abs(p[k + 1]) < 0.05 and np.max(np.delete(p, [k, k + 1])) < -70.0


def test_all_cores_receiver_equals_serial(oracle):
    """The OpenMP form of the oracle receiver (bench.py's all-cores CPU leg) produces the serial loop's frames bit
    for bit, also when the stream is split over calls with different thread counts and carries a ragged tail."""
    N, M, cp = 8, 64, 8
    iq, sent = oracle.synth_traffic(N, M, cp, 4, 5, payload_len=200)
    a = oracle.MultiChannelRx(N, M, cp, 4)
    a.execute(iq)
    b = oracle.MultiChannelRx(N, M, cp, 4)
    cut = (len(iq) // 2) // (2 * N) * (2 * N)
    b.execute_parallel(iq[:cut], 4)
    b.execute_parallel(iq[cut:-5], 3)
    b.execute_parallel(iq[-5:], 8)                                  # too short for threads: serial fallback
    key = lambda f: (f.channel, f.header, f.payload, f.header_valid, f.payload_valid, f.evm, f.rssi, f.cfo, f.framesyms.tobytes())
    assert len(a.frames) == 5 * N and sorted(map(key, a.frames)) == sorted(map(key, b.frames))
    c = oracle.MultiChannelRx(N, M, cp, 4, count_only=True)         # the bench's counting callback (no Python per frame)
    c.execute_parallel(iq, 4)
    assert c.counts() == (5 * N, 5 * N, 5 * N, 5 * N * 200)


@pytest.mark.parametrize("N,M,cp", [(4, 64, 8), (2, 256, 32), (16, 64, 8)])
def test_oversampled_front_end_receiver(oracle, N, M, cp):
    """multichannelrx with the oversampled front end (firpfbch2, 2N channels, + half-band decimator per channel):
    recovers exactly what the transmitter sent, the same frames as the critically sampled receiver, with an error
    vector magnitude that is no worse (the prototype leaves the channel edges alone); a tone in channel k comes
    out on channel k only; sample-at-a-time equals bulk."""
    iq, sent = oracle.synth_traffic(N, M, cp, 4, 3, payload_len=90, seed=N)
    a = oracle.MultiChannelRx(N, M, cp, 4); a.execute(iq)
    b = oracle.MultiChannelRx(N, M, cp, 4, front_end=1); b.execute(iq)
    key = lambda fr: sorted((f.channel, f.header, f.payload, int(f.payload_valid)) for f in fr)
    assert key(a.frames) == key(b.frames) and len(b.frames) == 3 * N
    for f in b.frames:
        assert f.payload_valid and sent[f.channel][(f.header[0] << 8) | f.header[1]] == (f.header, f.payload)
    assert np.mean([f.evm for f in b.frames]) <= np.mean([f.evm for f in a.frames]) + 0.5
    c = oracle.MultiChannelRx(N, M, cp, 4, front_end=1)
    for i in range(0, len(iq), 37):
        c.execute(iq[i:i + 37])
    assert key(c.frames) == key(b.frames)
    # tone at the centre of channel k (the transmitter's channel plan: bin k of the 2N-point bank, shifted down by the
    # receiver's oscillator offset)
    K = 2 * N
    k = N // 2
    f0 = k / K - 0.25 * (N - 1) / N                      # cycles per sample at the antenna
    t = np.arange(200 * K)
    y = oracle.MultiChannelRx(N, M, cp, 4).channelize_oversampled(np.exp(2j * np.pi * f0 * t).astype(np.complex64))
    p = np.mean(np.abs(y[60:]) ** 2, axis=0)
    assert np.argmax(p) == k and 10 * np.log10(np.sort(p)[-2] / p[k]) < -50
