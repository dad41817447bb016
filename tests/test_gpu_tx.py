"""GPU transmitter (multichanneltx on the GPU, the synthetic IQ source) against the oracle's
multichanneltx fed the same headers and payloads, and GPU TX -> GPU RX round trips."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def oracle_waveform(oracle, N, M, cp, taper, sent, mod, fec0, fec1, gain, nblocks):
    """Reference traffic loop (src/multichannel_tx.cc:163-213) with the given frames."""
    tx = oracle.MultiChannelTx(N, M, cp, taper)
    nxt = [0] * N
    L = M + cp
    chunks, produced = [], 0
    while produced < nblocks:
        for c in range(N):
            if nxt[c] < len(sent[c]) and tx.ready(c):
                h, p = sent[c][nxt[c]]
                tx.update(c, h, p, mod, fec0, fec1)
                nxt[c] += 1
        chunks.append(tx.generate(L))
        produced += L
    return (np.concatenate(chunks)[:nblocks * 2 * N] * np.float32(gain)).astype(np.complex64)


@pytest.mark.parametrize("N,M,cp,mod,fec1,plen,nf", [
    (1, 64, 8, 40, 6, 50, 2),
    (8, 64, 8, 40, 6, 300, 2),
    (4, 256, 32, 27, 7, 200, 2),
    (2, 128, 16, 29, 1, 77, 1),
    (2, 64, 8, 39, 7, 0, 2),
    (2, 48, 6, 40, 6, 100, 2),                  # src/multichannel_tx.cc defaults (M = 48: direct inverse DFT)
    # the BASELINE shapes: one whole 1200-byte frame per channel through the txifft_kernel<128 / 512 /
    # 1024> instantiations that make the IQ of configs[2], configs[4], configs[3] and of the headline bench
    (64, 256, 32, 27, 7, 1200, 1),
    (256, 64, 8, 40, 6, 1200, 1),
    (512, 64, 8, 40, 6, 1200, 1),
])
def test_gpu_tx_waveform_matches_oracle(oracle, product, N, M, cp, mod, fec1, plen, nf):
    import torch
    tx = product.multichanneltx(N, M, cp, 4)
    iq, sent = tx.generate(nf, plen, mod=mod, fec1=fec1, gain=1.0 / N, seed=1234)
    torch.cuda.synchronize()
    got = iq.cpu().numpy()
    nb = len(got) // (2 * N)
    ref = oracle_waveform(oracle, N, M, cp, 4, sent, mod, 1, fec1, 1.0 / N, nb)
    assert len(ref) == len(got)
    err = np.max(np.abs(got - ref)) / np.max(np.abs(ref))
    assert err <= 1e-5, err
    rms = float(np.sqrt(np.mean(np.abs(got - ref) ** 2)) / np.sqrt(np.mean(np.abs(ref) ** 2)))
    print("GPU multichanneltx vs oracle, N=%d M=%d: %d blocks, max-norm rel err %.3g, rms rel err %.3g" % (N, M, nb, err, rms))
    for c in range(N):                                      # traffic recipe: pid, channel id in the header
        for f, (h, p) in enumerate(sent[c]):
            assert h[0] == (f >> 8) and h[1] == (f & 0xff) and h[2] == (c & 0xff) and len(p) == plen      # (the channel id is one header byte, src/multichannel_tx.cc:175)
    tx.close()


# The fused synthesis kernel (csrc/synth_tile.hpp) picks its input side per launch: the aligned symbol loader (8 | L, cp, M and
# taper <= 4, rounds of 8 blocks: K = 512 / 1024) or the block-by-block walk (anything else).  Both against the oracle at
# sizes where they are instantiated, with the geometry that decides between them varied: taper 2 / 4 (aligned), taper 8 (walk),
# cp = 16 (S0a reads 32 samples back), M = 48 / cp = 6 (L = 54: walk, direct-DFT symbol bodies).
@pytest.mark.parametrize("N,M,cp,taper,plen", [
    (256, 64, 8, 2, 100),
    (256, 64, 8, 8, 100),
    (256, 64, 16, 4, 64),
    (256, 48, 6, 4, 60),
    (512, 64, 8, 3, 40),
])
def test_gpu_tx_fused_synthesis_input_paths(oracle, product, N, M, cp, taper, plen):
    import torch
    tx = product.multichanneltx(N, M, cp, taper)
    iq, sent = tx.generate(2, plen, mod=40, fec1=6, gain=1.0 / N, seed=4321)
    torch.cuda.synchronize()
    got = iq.cpu().numpy()
    nb = len(got) // (2 * N)
    ref = oracle_waveform(oracle, N, M, cp, taper, sent, 40, 1, 6, 1.0 / N, nb)
    assert len(ref) == len(got)
    err = np.max(np.abs(got - ref)) / np.max(np.abs(ref))
    assert err <= 1e-5, err
    tx.close()


def test_gpu_tx_to_gpu_rx_round_trip_stays_in_hbm(oracle, product):
    import torch
    N, M, cp = 16, 64, 8
    tx = product.multichanneltx(N, M, cp, 4)
    iq, sent = tx.generate(3, 400, seed=77)
    rx = product.multichannelrx(N, M, cp, 4)
    n = int(iq.numel()) // (32 * N) * (32 * N)
    rx.Execute(iq[:n])
    rx.Flush()
    assert len(rx.frames) == 3 * N
    for f in rx.frames:
        pid = (f.header[0] << 8) | f.header[1]
        assert f.payload_valid and sent[f.channel][pid] == (f.header, f.payload)
    rx.close(); tx.close()


def test_gpu_tx_argument_errors(product):
    for args in [(0, 64, 8, 4), (2, 7, 8, 4), (2, 64, 0, 0), (2, 64, 4, 5)]:       # lib/multichanneltx.cc:48-60
        with pytest.raises(ValueError):
            product.multichanneltx(*args)


# ---- streaming form: the reference class interface (IsChannelReadyForData / UpdateData / GenerateSamples / Reset)
@pytest.mark.parametrize("N,M,cp,mods,plens,maxpl", [
    (2, 64, 8, [(40, 1, 6)], [60], 512),
    (4, 64, 16, [(40, 1, 6), (27, 1, 7), (39, 7, 7), (29, 1, 1)], [0, 33, 150, 411], 512),  # mixed schemes, ragged lengths
    (8, 128, 16, [(40, 1, 6), (27, 1, 7)], [200, 90], 512),
    (4, 64, 8, [(40, 1, 6), (39, 7, 7)], [5, 40, 90, 260], 4),                              # frame slots grow while frames are in flight
])
def test_gpu_tx_streaming_class_matches_oracle(oracle, product, N, M, cp, mods, plens, maxpl):
    """Same call sequence on both classes: poll readiness, update the channels that are ready (some channels
    are left idle for a while so that frames start at different symbol periods), pull one block at a time."""
    rng = np.random.RandomState(11)
    ref = oracle.MultiChannelTx(N, M, cp, 4)
    tx = product.multichanneltx(N, M, cp, 4, max_payload_len=maxpl)
    L = M + cp
    nperiods = 70
    got, want = [], []
    nupd = 0
    busy_seen = False
    for per in range(nperiods):
        for c in range(N):
            r_ref, r_got = ref.ready(c), tx.IsChannelReadyForData(c)
            assert r_ref == r_got, (per, c)
            if not r_got:
                if not busy_seen:                                   # UpdateData on a busy channel: refused, state unchanged
                    assert tx.UpdateData(c, b"\0" * 8, b"x") is False
                    busy_seen = True
                continue
            if rng.rand() < 0.35:                                   # leave the channel idle this period
                continue
            mod, f0, f1 = mods[rng.randint(len(mods))]
            pl = bytes(rng.randint(0, 256, plens[rng.randint(len(plens))]).astype(np.uint8))
            h = bytes(rng.randint(0, 256, 8).astype(np.uint8))
            assert ref.update(c, h, pl, mod, f0, f1) == 0
            assert tx.UpdateData(c, h, pl, mod, f0, f1) is True
            nupd += 1
        want.append(ref.generate(L))
        got.append(np.concatenate([tx.GenerateSamples().copy() for _ in range(L)]))
    got, want = np.concatenate(got), np.concatenate(want)
    assert nupd >= 2 * N and busy_seen
    err = np.max(np.abs(got - want)) / np.max(np.abs(want))
    assert err <= 1e-5, err
    tx.close()


def test_gpu_tx_streaming_reset_mid_frame(oracle, product):
    """Reset (lib/multichanneltx.cc:126-149) drops the frames and the filter state but not the oscillator phase."""
    N, M, cp = 2, 64, 8
    L = M + cp
    ref = oracle.MultiChannelTx(N, M, cp, 4)
    tx = product.multichanneltx(N, M, cp, 4, max_payload_len=256)
    rng = np.random.RandomState(5)

    def step(nblocks):
        a = ref.generate(nblocks)
        b = np.concatenate([tx.GenerateSamples().copy() for _ in range(nblocks)])
        return a, b

    def load():
        for c in range(N):
            if ref.ready(c):
                h, pl = bytes(rng.randint(0, 256, 8).astype(np.uint8)), bytes(rng.randint(0, 256, 100).astype(np.uint8))
                ref.update(c, h, pl); tx.UpdateData(c, h, pl)

    load()
    a0, b0 = step(3 * L + 17)                                       # stop in the middle of a symbol period
    ref.reset(); tx.Reset()
    assert all(ref.ready(c) and tx.IsChannelReadyForData(c) for c in range(N))
    load()
    a1, b1 = step(30 * L)
    a, b = np.concatenate([a0, a1]), np.concatenate([b0, b1])
    assert np.max(np.abs(a - b)) / np.max(np.abs(a)) <= 1e-5
    tx.close()


def test_gpu_tx_streaming_feeds_gpu_rx(oracle, product):
    """Class-interface transmitter -> receiver round trip with ragged traffic."""
    import torch
    N, M, cp = 4, 64, 8
    L = M + cp
    tx = product.multichanneltx(N, M, cp, 4, max_payload_len=300)
    rng = np.random.RandomState(3)
    sent = {}
    pid = [0] * N
    blocks = []
    idle = 0
    while idle < 3:                                                 # until every frame has gone out, plus the filter tail
        idle = idle + 1 if all(pid[c] == 4 and tx.IsChannelReadyForData(c) for c in range(N)) else 0
        for c in range(N):
            if pid[c] < 4 and tx.IsChannelReadyForData(c) and rng.rand() < 0.7:
                pl = bytes(rng.randint(0, 256, int(rng.randint(1, 300))).astype(np.uint8))
                h = bytes([0, pid[c], c]) + bytes(rng.randint(0, 256, 5).astype(np.uint8))
                assert tx.UpdateData(c, h, pl)
                sent[(c, pid[c])] = (h, pl); pid[c] += 1
        blocks.extend(tx.GenerateSamples().copy() for _ in range(L))
    iq = (np.concatenate(blocks) / np.float32(N)).astype(np.complex64)
    rx = product.multichannelrx(N, M, cp, 4)
    n = len(iq) // (32 * N) * (32 * N)
    rx.Execute(torch.from_numpy(iq[:n]).cuda())
    rx.Flush()
    assert len(rx.frames) == len(sent) == 4 * N
    for f in rx.frames:
        assert f.payload_valid and sent[(f.channel, f.header[1])] == (f.header, f.payload)
    rx.close(); tx.close()


# ---- ragged traffic (src/multichannel_txrx.cc:227-267): every frame its own length, irregular pauses
def oracle_waveform_ragged(oracle, N, M, cp, taper, sent, starts, mod, fec0, fec1, gain, nblocks):
    """The oracle's class driven so that channel c's frame f starts at block starts[c][f] (a symbol boundary)."""
    tx = oracle.MultiChannelTx(N, M, cp, taper)
    L = M + cp
    nxt = [0] * N
    chunks = []
    for t in range(0, nblocks, L):
        for c in range(N):
            if nxt[c] < len(sent[c]) and starts[c][nxt[c]] == t:
                assert tx.ready(c), (c, nxt[c], t)
                h, p = sent[c][nxt[c]]
                tx.update(c, h, p, mod, fec0, fec1)
                nxt[c] += 1
        chunks.append(tx.generate(min(L, nblocks - t)))
    assert all(nxt[c] == len(sent[c]) for c in range(N))
    return (np.concatenate(chunks)[:nblocks * 2 * N] * np.float32(gain)).astype(np.complex64)


# (N = 256: the fused synthesis kernel's aligned symbol loader with the per-symbol role map -- synth_kernel<512, 8, SYN_SYMS>,
#  symbol-major bodies and roles; the small cases run the two-kernel path through frame_sample_sym)
@pytest.mark.parametrize("N,M,cp,mod,fec1", [(4, 64, 8, 40, 6), (2, 128, 16, 27, 7), (256, 64, 8, 40, 6)])
def test_gpu_tx_ragged_traffic_matches_oracle_and_both_receivers_agree(oracle, product, N, M, cp, mod, fec1):
    import torch
    from test_gpu_parity import check_frames
    L = M + cp
    nb = L * 420 // 8 * 8
    tx = product.multichanneltx(N, M, cp, 4)
    iq, sent, starts = tx.generate_ragged(nb, len_lo=0, len_hi=300, gap_max=3, long_every=4, long_max=40, mod=mod, fec1=fec1,
                                          gain=1.0 / N, seed=99)
    tx.close()
    torch.cuda.synchronize()
    got = iq.cpu().numpy()
    assert all(len(s) >= 2 for s in sent) and len({len(p) for s in sent for (_, p) in s}) > 4      # really ragged
    for c in range(N):
        assert all(b % L == 0 for b in starts[c]) and starts[c] == sorted(starts[c])
    ref = oracle_waveform_ragged(oracle, N, M, cp, 4, sent, starts, mod, 1, fec1, 1.0 / N, nb)
    err = np.max(np.abs(got - ref)) / np.max(np.abs(ref))
    assert err <= 1e-5, err
    # ... and the two receivers decode it identically, every frame what was sent
    x = got[:len(got) // (32 * N) * (32 * N)]
    ora = oracle.MultiChannelRx(N, M, cp, 4)
    ora.execute(x)
    nsent = sum(len(s) for s in sent)
    # (a channel that idles can lock onto its neighbours' leakage -- the detector is gain-normalised -- and miss the frame
    #  that starts underneath: a property of the reference's synchronizer; what matters here is that both receivers do the same)
    good = [f for f in ora.frames if f.header_valid and f.payload_valid]
    assert len(good) >= nsent - N, (len(good), nsent)
    rx = product.multichannelrx(N, M, cp, 4, max_payload_len=300)
    step = 32 * N * 61
    for i in range(0, len(x), step):
        rx.Execute(x[i:i + step])
    rx.Flush()
    check_frames(rx.frames, ora.frames, leak_rssi=-40.0)          # (N = 256: one idle channel decodes its neighbour's frame at -58.8 dB)
    for f in rx.frames:
        if f.header_valid and f.payload_valid and f.rssi > -40.0:
            assert sent[f.channel][(f.header[0] << 8) | f.header[1]] == (f.header, f.payload)
    rx.close()
