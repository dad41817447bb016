"""GPU parity at the BASELINE.json configurations' own shapes, against the CPU oracle (the IQ comes from the GPU
transmitter, which tests/test_gpu_tx.py holds to the oracle's transmitter).

configs[2]: 64-ch multichannelrx, M=256 (cp=32), QAM16 + Golay(24,12) (the reference's r=1/2 default,
            src/multichannel_txrx.cc:132), msresamp(0.5) front end fed by the 2x stream a TX-side msresamp(2.0)
            makes (src/flexframe_tx.cc:170)
configs[3]: 512-ch multichannelrx, M=64, QPSK + Hamming(12,8), single GPU, and the 8-rank round-robin sharding
            of it emulated on one GPU (one handle per rank, the all-to-all played by slicing)."""
import numpy as np
import pytest

from test_gpu_parity import relerr_elem, REL_ELEM

pytestmark = pytest.mark.gpu
REL = 1e-5


def _torch():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _compare(gpu_frames, ora_frames, rel=REL):
    o = {}
    for f in ora_frames:
        o.setdefault(f.channel, []).append(f)
    g = {}
    for f in gpu_frames:
        g.setdefault(f.channel, []).append(f)
    assert sorted(g) == sorted(o)
    worst = worst_e = 0.0
    for c in o:
        assert len(g[c]) == len(o[c]), (c, len(g[c]), len(o[c]))
        for a, b in zip(g[c], o[c]):
            assert (a.header, a.payload, a.header_valid, a.payload_valid) == (b.header, b.payload, b.header_valid, b.payload_valid)
            assert len(a.framesyms) == len(b.framesyms)
            worst = max(worst, float(np.max(np.abs(a.framesyms - b.framesyms)) / np.max(np.abs(b.framesyms))))
            worst_e = max(worst_e, relerr_elem(a.framesyms, b.framesyms))
    assert worst <= rel, worst
    assert worst_e <= REL_ELEM, (worst, worst_e)      # element-wise figure, reported beside the bar (see test_gpu_parity.py)
    _compare.last_elem = worst_e
    return worst


def test_config3_512_channels_full_chain_vs_oracle(oracle, product):
    """BASELINE configs[3] on one GPU: N=512 (K=1024), one 1200-byte frame per channel, every frame and its
    equalised symbols against the oracle."""
    torch = _torch()
    N, M, cp = 512, 64, 8
    tx = product.multichanneltx(N, M, cp, 4)
    iq, sent = tx.generate(1, 1200, seed=0xC0FFEE)
    tx.close()
    x = iq.cpu().numpy()
    ora = oracle.MultiChannelRx(N, M, cp, 4)
    for i in range(0, len(x), 1 << 22):
        ora.execute(x[i:i + (1 << 22)])
    assert len(ora.frames) == N and all(f.payload_valid for f in ora.frames)
    rx = product.multichannelrx(N, M, cp, 4, max_payload_len=1200)
    rx.Execute(iq); rx.Flush()
    worst = _compare(rx.frames, ora.frames)
    for f in rx.frames:
        assert sent[f.channel][(f.header[0] << 8) | f.header[1]] == (f.header, f.payload)
    rx.close()
    print("config 3 (N=512) worst framesyms rel err %.3g (element-wise %.3g)" % (worst, _compare.last_elem))


def test_config5_256_channels_receive_side_vs_oracle(oracle, product):
    """BASELINE configs[4]'s receive side at its own size: N=256 (K=512), M=64, QPSK + Hamming(12,8),
    two 1200-byte frames per channel from the GPU transmitter; every frame, flag and equalised symbol against the oracle
    receiver on the same IQ."""
    torch = _torch()
    N, M, cp = 256, 64, 8
    tx = product.multichanneltx(N, M, cp, 4)
    iq, sent = tx.generate(2, 1200, seed=0x5EED)
    tx.close()
    x = iq.cpu().numpy()
    ora = oracle.MultiChannelRx(N, M, cp, 4)
    for i in range(0, len(x), 1 << 22):
        ora.execute(x[i:i + (1 << 22)])
    assert len(ora.frames) == 2 * N and all(f.payload_valid for f in ora.frames)
    rx = product.multichannelrx(N, M, cp, 4, max_payload_len=1200)
    n = int(iq.numel()) // (32 * N) * (32 * N)
    rx.Execute(iq[:n]); rx.Flush()
    worst = _compare(rx.frames, ora.frames)
    for f in rx.frames:
        assert sent[f.channel][(f.header[0] << 8) | f.header[1]] == (f.header, f.payload)
    rx.close()
    print("config 5 receive side (N=256) worst framesyms rel err %.3g (element-wise %.3g)" % (worst, _compare.last_elem))


def test_config2_64_channels_m256_qam16_golay_resampled_vs_oracle(oracle, product):
    """BASELINE configs[2] at its own shape: N=64, M=256, cp=32, QAM16, Golay(24,12), 2 frames per channel; the
    wideband stream is interpolated by msresamp(2.0) on the transmit side and brought back by the GPU msresamp(0.5)
    front end; the oracle runs the same chain (its own resamplers) on the same 2x stream."""
    torch = _torch()
    N, M, cp = 64, 256, 32
    tx = product.multichanneltx(N, M, cp, 4)
    iq, sent = tx.generate(2, 1200, mod=oracle.MODEM_QAM16, fec1=oracle.FEC_GOLAY2412, seed=77)
    tx.close()
    up_gpu = product.msresamp(2.0)
    d_up = up_gpu.execute(iq)                                  # the 2x-oversampled antenna stream (TX side, src/flexframe_tx.cc:170)
    up_gpu.close()
    x2 = d_up.cpu().numpy()
    # oracle chain
    o_y = oracle.MsResamp(0.5).execute(x2)
    ora = oracle.MultiChannelRx(N, M, cp, 4)
    ora.execute(o_y)
    assert len(ora.frames) == 2 * N and all(f.payload_valid for f in ora.frames)
    # GPU chain
    rs = product.msresamp(0.5)
    d_y = rs.execute(d_up)
    m = min(len(o_y), int(d_y.numel()))
    err_rs = float(np.max(np.abs(d_y.cpu().numpy()[:m] - o_y[:m])) / np.max(np.abs(o_y[:m])))
    assert err_rs <= REL, err_rs
    rx = product.multichannelrx(N, M, cp, 4, max_payload_len=1200)
    n = int(d_y.numel()) // (32 * N) * (32 * N)
    rx.Execute(d_y[:n].contiguous()); rx.Flush()
    worst = _compare(rx.frames, [f for f in ora.frames])
    for f in rx.frames:
        assert sent[f.channel][(f.header[0] << 8) | f.header[1]] == (f.header, f.payload)
    rx.close(); rs.close()
    print("config 2 (N=64, M=256, QAM16+Golay, msresamp) resampler err %.3g, worst framesyms rel err %.3g (element-wise %.3g)" % (err_rs, worst, _compare.last_elem))


def test_config3_eight_rank_round_robin_sharding_emulated(oracle, product):
    """The 8-GPU form of configs[3] with all eight ranks' handles in one process: sub-slabs round robin over the
    ranks (halo = the 13 blocks before the sub-slab), per-destination groups, the all-to-all played by slicing,
    every rank's 64-channel shard synchronized round after round with its history tiles in front -- exactly the
    calls sharding.Pipeline makes.  Frames = what was sent, and = one unsharded receiver's, symbol for symbol."""
    torch = _torch()
    from liquid_usrp_amd import sharding
    N, M, cp, world, rounds = 512, 64, 8, 8, 2
    K, cg = 2 * N, N // world
    tx = product.multichanneltx(N, M, cp, 4)
    iq, sent = tx.generate(2, 400, seed=31)
    tx.close()
    unit = product.TILE * world * rounds
    T = int(iq.numel()) // K
    tot = (T + unit - 1) // unit * unit
    stream = torch.cat([iq, torch.zeros((tot - T) * K, dtype=torch.complex64, device="cuda")])
    Tc = tot // (world * rounds)
    TS = product.TILE
    tiles = Tc // TS
    # reference: one handle over the whole stream
    one = product.multichannelrx(N, M, cp, 4, max_payload_len=400)
    one.Execute(stream); one.Flush()
    assert len(one.frames) == 2 * N
    rxs = []
    for r in range(world):
        c0, cnt = sharding.shard_of(r, world, N)
        rxs.append(product.multichannelrx(N, M, cp, 4, max_payload_len=400, channel_first=c0, channel_count=cnt))
    H = rxs[0].hist_tiles
    prev = [None] * world
    for c in range(rounds):
        outs = []
        for r in range(world):
            u = c * world + r
            o = torch.empty(world * tiles * cg * TS, dtype=torch.complex64, device="cuda")
            halo = stream[(u * Tc - 13) * K:u * Tc * K] if u > 0 else None
            rxs[r].channelize(stream[u * Tc * K:(u + 1) * Tc * K], Tc, u * Tc * K, o, groups=world, d_halo=halo)
            outs.append(o)
        torch.cuda.synchronize()
        per = tiles * cg * TS
        for r in range(world):
            new = torch.cat([outs[s][r * per:(r + 1) * per] for s in range(world)])          # all_to_all_single
            hist = prev[r][-H * cg * TS:] if prev[r] is not None else torch.zeros(H * cg * TS, dtype=torch.complex64, device="cuda")
            buf = torch.cat([hist, new])
            rxs[r].sync(buf, c * world * Tc - H * TS, H * TS + world * Tc)
            prev[r] = buf
        torch.cuda.synchronize()
    got = []
    for r in range(world):
        rxs[r].Flush()
        c0, cnt = sharding.shard_of(r, world, N)
        assert all(c0 <= f.channel < c0 + cnt for f in rxs[r].frames)
        got += rxs[r].frames
        rxs[r].close()
    assert len(got) == 2 * N
    for f in got:
        assert f.payload_valid and sent[f.channel][(f.header[0] << 8) | f.header[1]] == (f.header, f.payload)
    worst = _compare(got, one.frames)
    one.close()
    print("8-rank emulation vs one handle: worst framesyms rel err %.3g" % worst)
