"""Committed golden fixtures (tests/golden/*.npz, produced by make_golden.py from the oracle):
the oracle must keep reproducing them (CPU), and the HIP path must decode them (GPU)."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FRAMES = ["frame_m64_qpsk_h128", "frame_m256_qam16_g2412", "frame_m48_bpsk_none"]


def test_oracle_design_matches_golden(oracle):
    d = np.load(os.path.join(G, "design.npz"))
    for K in (2, 16, 128, 1024):
        assert np.array_equal(oracle.Channelizer(oracle.ANALYZER, K, 7).taps(), d["taps_rx_K%d" % K])
    assert np.array_equal(oracle.Channelizer(oracle.SYNTHESIZER, 16, 13).taps(), d["taps_tx_K16"])
    for M in (48, 64, 256):
        p = oracle.default_sctype(M)
        assert np.array_equal(p, d["sctype_M%d" % M])
        S = oracle.init_S0S1(p)
        assert np.array_equal(S["S0"][0], d["S0_M%d" % M]) and np.array_equal(S["S1"][0], d["S1_M%d" % M])
        assert np.allclose(S["S0"][1], d["s0_M%d" % M], atol=1e-7) and np.allclose(S["S1"][1], d["s1_M%d" % M], atol=1e-7)
        assert np.allclose(oracle.pilot_fit(p), d["pilotfit_M%d" % M], atol=1e-9)


@pytest.mark.parametrize("name", FRAMES)
def test_oracle_reproduces_golden_frame(oracle, name):
    g = np.load(os.path.join(G, name + ".npz"))
    M, cp, tp = int(g["M"]), int(g["cp"]), int(g["taper"])
    fg = oracle.FlexFrameGen(M, cp, tp, fec1=int(g["fec1"]), mod=int(g["mod"]))
    tx = fg.frame(bytes(g["header"]), bytes(g["payload"]))
    assert np.allclose(tx, g["tx"], atol=2e-7)
    fs = oracle.FlexFrameSync(M, cp, tp)
    fs.execute(g["rx"])
    f = fs.frames[0]
    assert f.header == bytes(g["header"]) and f.payload == bytes(g["payload"]) and f.payload_valid
    assert np.max(np.abs(f.framesyms - g["framesyms"])) <= 2e-6 * np.max(np.abs(g["framesyms"]))


def test_oracle_reproduces_golden_multichannel(oracle):
    g = np.load(os.path.join(G, "mc8.npz"))
    N, M, cp, tp = int(g["N"]), int(g["M"]), int(g["cp"]), int(g["taper"])
    chan = oracle.MultiChannelRx(N, M, cp, tp).channelize(g["iq"])
    assert np.max(np.abs(chan - g["chan"])) <= 2e-6 * np.max(np.abs(g["chan"]))


def test_oracle_reproduces_golden_oversampled_bank(oracle):
    g = np.load(os.path.join(G, "pfb2.npz"))
    for M, m in ((16, 4), (64, 7)):
        ch = oracle.Channelizer2(M, m)
        assert np.array_equal(ch.taps(), g["taps_M%d" % M])
        y = ch.analyze(g["x_M%d" % M])
        assert np.max(np.abs(y - g["y_M%d" % M])) <= 2e-6 * np.max(np.abs(g["y_M%d" % M]))


@pytest.mark.gpu
def test_gpu_oversampled_bank_reproduces_golden(product):
    import torch
    g = np.load(os.path.join(G, "pfb2.npz"))
    for M, m in ((16, 4), (64, 7)):
        pfb = product.firpfbch2(M, m)
        assert np.array_equal(pfb.taps(), g["taps_M%d" % M])
        y = pfb.analyze(torch.from_numpy(g["x_M%d" % M]).cuda()).cpu().numpy()
        assert np.max(np.abs(y - g["y_M%d" % M])) <= 1e-5 * np.max(np.abs(g["y_M%d" % M]))
        pfb.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", FRAMES)
def test_gpu_sync_decodes_golden_frame(product, name):
    import torch
    g = np.load(os.path.join(G, name + ".npz"))
    M, cp, tp = int(g["M"]), int(g["cp"]), int(g["taper"])
    rx_iq = g["rx"]
    n = len(rx_iq) // product.TILE * product.TILE
    rx = product.multichannelrx(1, M, cp, tp)
    d = torch.from_numpy(rx_iq[:n].copy()).cuda()         # one channel: [tile][1][TILE] is the stream itself
    rx.sync(d, 0, n)
    rx.Flush()
    assert len(rx.frames) == 1
    f = rx.frames[0]
    assert f.header == bytes(g["header"]) and f.payload == bytes(g["payload"]) and f.payload_valid == 1
    assert np.max(np.abs(f.framesyms - g["framesyms"])) <= 1e-5 * np.max(np.abs(g["framesyms"]))
    assert abs(f.cfo - float(g["cfo"])) < 1e-6 and abs(f.rssi - float(g["rssi"])) < 1e-3
    rx.close()


@pytest.mark.gpu
def test_gpu_multichannel_matches_golden(product):
    import torch
    g = np.load(os.path.join(G, "mc8.npz"))
    N, M, cp, tp = int(g["N"]), int(g["M"]), int(g["cp"]), int(g["taper"])
    rx = product.multichannelrx(N, M, cp, tp)
    iq = g["iq"]
    nb0 = len(iq) // (2 * N)
    nb = -(-nb0 // product.TILE) * product.TILE              # whole tiles: zeros behind the recorded stream
    d_x = torch.from_numpy(np.concatenate([iq, np.zeros((nb - nb0) * 2 * N, np.complex64)])).cuda()
    d_out = torch.zeros(nb * N, dtype=torch.complex64, device="cuda")
    rx.channelize(d_x, nb, 0, d_out)
    torch.cuda.synchronize()
    got = product.tiles_to_channels(d_out, N).T[:nb0]
    assert np.max(np.abs(got - g["chan"])) <= 1e-5 * np.max(np.abs(g["chan"]))
    rx.Execute(d_x)
    rx.Flush()
    got_frames = {f.channel: (f.header, f.payload) for f in rx.frames if f.payload_valid}
    for ch, h, p in zip(g["channels"], g["headers"], g["payloads"]):
        assert got_frames[int(ch)] == (bytes(h), bytes(p))
    rx.close()
