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
    for c in range(N):                                      # traffic recipe: pid, channel id in the header
        for f, (h, p) in enumerate(sent[c]):
            assert h[0] == (f >> 8) and h[1] == (f & 0xff) and h[2] == c and len(p) == plen
    tx.close()


def test_gpu_tx_to_gpu_rx_round_trip_stays_in_hbm(oracle, product):
    import torch
    N, M, cp = 16, 64, 8
    tx = product.multichanneltx(N, M, cp, 4)
    iq, sent = tx.generate(3, 400, seed=77)
    rx = product.multichannelrx(N, M, cp, 4)
    n = int(iq.numel()) // (16 * N) * (16 * N)
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
