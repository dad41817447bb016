#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz from the CPU oracle (run in the build container).

The reference holds no golden vectors and liquid-dsp is unavailable, so
these fixtures pin the ORACLE's behaviour at commit time: design data (prototype taps, training
symbols, allocation), one transmitted frame per PHY configuration with its decoded bytes and
equalised symbols, a short multichannel stream with the channelizer output, and the oversampled analysis bank's
prototype and outputs for a seeded input.  They guard the
oracle against drift and give the GPU path data-only test cases."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import oracle as O  # noqa: E402


def design():
    d = {}
    for K in (2, 16, 128, 1024):
        d["taps_rx_K%d" % K] = O.Channelizer(O.ANALYZER, K, 7).taps()
    d["taps_tx_K16"] = O.Channelizer(O.SYNTHESIZER, 16, 13).taps()
    for M in (48, 64, 256):
        p = O.default_sctype(M)
        S = O.init_S0S1(p)
        d["sctype_M%d" % M] = p
        d["S0_M%d" % M], d["s0_M%d" % M] = S["S0"][0], S["S0"][1]
        d["S1_M%d" % M], d["s1_M%d" % M] = S["S1"][0], S["S1"][1]
        d["pilotfit_M%d" % M] = O.pilot_fit(p)
    np.savez_compressed(os.path.join(HERE, "design.npz"), **d)


def frame(name, M, cp, taper, mod, fec1, plen, seed):
    rng = np.random.RandomState(seed)
    hdr = bytes(rng.randint(0, 256, 8).astype(np.uint8))
    pl = bytes(rng.randint(0, 256, plen).astype(np.uint8))
    fg = O.FlexFrameGen(M, cp, taper, fec1=fec1, mod=mod)
    x = fg.frame(hdr, pl)
    n = np.arange(len(x) + 400)
    sig = np.concatenate([np.zeros(137, np.complex64), x, np.zeros(263, np.complex64)])
    sig = (sig * 0.4 * np.exp(1j * (0.3 + 0.002 * n))).astype(np.complex64)
    sig = (sig + 0.004 * (rng.randn(len(sig)) + 1j * rng.randn(len(sig)))).astype(np.complex64)
    fs = O.FlexFrameSync(M, cp, taper)
    fs.execute(sig)
    assert len(fs.frames) == 1 and fs.frames[0].payload_valid and fs.frames[0].payload == pl
    f = fs.frames[0]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), M=M, cp=cp, taper=taper, mod=mod, fec1=fec1,
                        tx=x, rx=sig, header=np.frombuffer(hdr, np.uint8), payload=np.frombuffer(pl, np.uint8),
                        framesyms=f.framesyms, evm=f.evm, rssi=f.rssi, cfo=f.cfo)


def multichannel():
    N, M, cp, tp = 8, 64, 8, 4
    iq, sent = O.synth_traffic(N, M, cp, tp, 1, payload_len=40, seed=7)
    nb = len(iq) // (2 * N) // 8 * 8
    iq = iq[:nb * 2 * N]
    chan = O.MultiChannelRx(N, M, cp, tp).channelize(iq)
    rx = O.MultiChannelRx(N, M, cp, tp)
    rx.execute(iq)
    assert len(rx.frames) == N
    np.savez_compressed(os.path.join(HERE, "mc8.npz"), N=N, M=M, cp=cp, taper=tp, iq=iq, chan=chan,
                        headers=np.array([np.frombuffer(f.header, np.uint8) for f in rx.frames]),
                        payloads=np.array([np.frombuffer(f.payload, np.uint8) for f in rx.frames]),
                        channels=np.array([f.channel for f in rx.frames]))


def oversampled_bank():
    """firpfbch2-style analysis bank: prototype and the outputs for a seeded input (M = 16, m = 4 and M = 64, m = 7)."""
    d = {}
    for M, m in ((16, 4), (64, 7)):
        rng = np.random.RandomState(100 + M)
        x = (rng.randn(40 * M // 2) + 1j * rng.randn(40 * M // 2)).astype(np.complex64)
        ch = O.Channelizer2(M, m)
        d["taps_M%d" % M], d["x_M%d" % M], d["y_M%d" % M] = ch.taps(), x, ch.analyze(x)
    np.savez_compressed(os.path.join(HERE, "pfb2.npz"), **d)


if __name__ == "__main__":
    design()
    oversampled_bank()
    frame("frame_m64_qpsk_h128", 64, 8, 4, O.MODEM_QPSK, O.FEC_HAMMING128, 64, 1)
    frame("frame_m256_qam16_g2412", 256, 32, 4, O.MODEM_QAM16, O.FEC_GOLAY2412, 100, 2)
    frame("frame_m48_bpsk_none", 48, 6, 4, O.MODEM_BPSK, O.FEC_NONE, 21, 3)
    multichannel()
    print(sorted(os.listdir(HERE)))
