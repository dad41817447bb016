"""Custom subcarrier allocations (the `_p` argument of the reference's constructors: include/multichannelrx.h:39-45,
lib/multichannelrx.cc:82, lib/multichanneltx.cc:62-66, lib/ofdmtxrx.cc:91; the applications pass NULL = liquid's default).

VERDICT r4, weak #6: design.hpp accepts any allocation, but the lean workers, the segment waves and the transmit kernels bake pilot and
data geometry into lane constants -- and no GPU test had ever passed a non-NULL `_p`.  Here: wider guard bands, denser pilots, a
pilot count that pushes a 64-subcarrier design off the lean M = 64 kernels (> 16 pilots), at M = 64 and M = 256, receive and transmit,
through the default (segment-wave) path, the lean segment waves (csrc/acq_lean.hpp) and the one-kernel scout."""
import numpy as np
import pytest

from test_gpu_parity import check_frames

pytestmark = pytest.mark.gpu

NULL, PILOT, DATA = 0, 1, 2


def alloc(M, guard, pilot_every, pilot_phase=None):
    """liquid's default rule (ofdmframe_init_default_sctype) with its two parameters free: `guard` null subcarriers on either side
    of the band edge at M / 2, DC null, a pilot on every `pilot_every`-th enabled subcarrier."""
    p = np.zeros(M, np.uint8)
    ph = pilot_every // 2 if pilot_phase is None else pilot_phase
    for i in range(1, M // 2 - guard):
        t = PILOT if (i + ph) % pilot_every == 0 else DATA
        p[i] = t
        p[M - i] = t
    return p


ALLOCS = {
    "m64_wide_guard": (64, 8, alloc(64, 16, 6)),            # 30 enabled, 4 pilots
    "m64_dense_pilots": (64, 8, alloc(64, 6, 4)),           # 12 pilots: still inside one DPP row (lean M = 64 kernels)
    "m64_20_pilots": (64, 8, alloc(64, 4, 3, 1)),           # 18 pilots > 16: off the lean M = 64 worker and the lean segment waves
    "m256_wide_guard": (256, 32, alloc(256, 60, 8)),
    "m256_dense_pilots": (256, 32, alloc(256, 25, 4)),      # 50 pilots (default: 26)
}


def _counts(p):
    return int(np.sum(p == PILOT)), int(np.sum(p == DATA))


@pytest.mark.parametrize("name", sorted(ALLOCS))
@pytest.mark.parametrize("build", ["default", "lean_segments", "one_kernel_scout"])
def test_receiver_with_a_custom_allocation_matches_oracle(oracle, product, name, build):
    M, cp, p = ALLOCS[name]
    npil, ndat = _counts(p)
    assert npil >= 2 and ndat >= 1
    N, nf, plen = 4, 3, 150
    mod, fec1 = (27, 7) if "dense" in name else (40, 6)
    iq, sent = oracle.synth_traffic(N, M, cp, 4, nf, payload_len=plen, mod=mod, fec1=fec1, seed=11, p=p)
    x = iq[:len(iq) // (32 * N) * (32 * N)]
    ora = oracle.MultiChannelRx(N, M, cp, 4, p=p)
    ora.execute(x)
    assert len(ora.frames) == nf * N and all(f.payload_valid for f in ora.frames)
    for f in ora.frames:
        assert sent[f.channel][(f.header[0] << 8) | f.header[1]] == (f.header, f.payload)
    cfg = {"default": {}, "lean_segments": {"scout_build": 2}, "one_kernel_scout": {"acquisition": 4}}[build]
    rx = product.multichannelrx(N, M, cp, 4, p=bytes(p), max_payload_len=plen, **cfg)
    half = len(x) // 2 // (32 * N) * (32 * N)
    rx.Execute(x[:half]); rx.Execute(x[half:]); rx.Flush()          # (two pushes: frames straddle the cut)
    check_frames(rx.frames, ora.frames)
    rx.close()


@pytest.mark.parametrize("name", sorted(ALLOCS))
def test_transmitter_with_a_custom_allocation_matches_oracle_and_round_trips(oracle, product, name):
    import torch
    from test_gpu_tx import oracle_waveform  # noqa: F401  (the default-allocation helper; the loop below is its `_p` form)
    M, cp, p = ALLOCS[name]
    N, nf, plen = 4, 2, 120
    mod, fec1 = (27, 7) if "dense" in name else (40, 6)
    tx = product.multichanneltx(N, M, cp, 4, p=bytes(p))
    iq, sent = tx.generate(nf, plen, mod=mod, fec1=fec1, gain=1.0 / N, seed=77)
    torch.cuda.synchronize()
    got = iq.cpu().numpy()
    nb = len(got) // (2 * N)
    # the oracle's class driven with the same frames (src/multichannel_tx.cc:163-213)
    otx = oracle.MultiChannelTx(N, M, cp, 4, p)
    nxt, chunks, produced, L = [0] * N, [], 0, M + cp
    while produced < nb:
        for c in range(N):
            if nxt[c] < len(sent[c]) and otx.ready(c):
                h, pl = sent[c][nxt[c]]
                otx.update(c, h, pl, mod, 1, fec1)
                nxt[c] += 1
        chunks.append(otx.generate(L))
        produced += L
    ref = (np.concatenate(chunks)[:nb * 2 * N] * np.float32(1.0 / N)).astype(np.complex64)
    err = float(np.max(np.abs(got - ref)) / np.max(np.abs(ref)))
    assert err <= 1e-5, err
    # ... and what the GPU transmitter sent comes back through the GPU receiver with the same allocation
    rx = product.multichannelrx(N, M, cp, 4, p=bytes(p), max_payload_len=plen)
    rx.Execute(iq[:int(iq.numel()) // (32 * N) * (32 * N)]); rx.Flush()
    assert len(rx.frames) == nf * N
    for f in rx.frames:
        assert f.payload_valid and sent[f.channel][(f.header[0] << 8) | f.header[1]] == (f.header, f.payload)
    rx.close(); tx.close()


def test_allocation_errors_like_the_reference(product):
    """liquid's ofdmframe_validate_sctype: at least one data and two pilot subcarriers, nothing but 0 / 1 / 2."""
    for bad in (np.zeros(64, np.uint8), np.full(64, DATA, np.uint8), np.full(64, 3, np.uint8)):
        with pytest.raises((ValueError, RuntimeError)):
            product.multichannelrx(2, 64, 8, 4, p=bytes(bad))
        with pytest.raises((ValueError, RuntimeError)):
            product.multichanneltx(2, 64, 8, 4, p=bytes(bad))
