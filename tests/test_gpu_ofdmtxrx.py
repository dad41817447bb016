"""BASELINE config 1 -- one ofdmflexframegen -> one ofdmflexframesync, no channelizer (the DSP core of the
reference's ofdmtxrx, lib/ofdmtxrx.cc:78-91,297-342,620-626) -- on the GPU kernels, against the oracle."""
import numpy as np
import pytest

from test_gpu_parity import check_frames, relerr

pytestmark = pytest.mark.gpu


def traffic(oracle, M, cp, taper, nframes, plen, mod, fec0, fec1, seed, gain=0.2512, gap=(0, 300)):
    """Frames as ofdmtxrx::transmit_packet sends them (each symbol x tx_gain, the last symbol buffer sent twice,
    lib/ofdmtxrx.cc:318-342), separated by idle gaps of random length."""
    rng = np.random.RandomState(seed)
    fg = oracle.FlexFrameGen(M, cp, taper, None, oracle.CRC_32, fec0, fec1, mod)
    L = M + cp
    out, sent = [np.zeros(rng.randint(1, 200), np.complex64)], []
    for f in range(nframes):
        h = bytes([f >> 8, f & 0xff]) + bytes(rng.randint(0, 256, 6).astype(np.uint8))
        pl = bytes(rng.randint(0, 256, plen if np.isscalar(plen) else rng.randint(*plen)).astype(np.uint8))
        x = fg.frame(h, pl) * np.float32(gain)
        out += [x, x[-L:], np.zeros(rng.randint(*gap), np.complex64)]
        sent.append((h, pl))
    out.append(np.zeros(4 * L, np.complex64))
    return np.concatenate(out).astype(np.complex64), sent


@pytest.mark.parametrize("M,cp,mod,fec0,fec1,plen", [
    (64, 8, 40, 1, 6, 1200),         # the ofdmtxrx defaults: QPSK, crc32, none + h128
    (64, 16, 39, 7, 7, 40),
    (256, 32, 27, 1, 7, 700),
    (128, 16, 29, 1, 1, 333),
    (64, 8, 40, 1, 6, 0),
    (48, 6, 40, 1, 7, 1200),         # the reference applications' default numerology: direct inverse DFT
    (80, 10, 27, 1, 6, 100),
    (64, 8, 40, 1, 2, 300),          # the short block codes the applications' -c / -k options name first: rep3 ...
    (64, 8, 27, 3, 4, 77),           # ... rep5 inside Hamming(7,4) (bit-packed 7-bit symbols) ...
    (128, 16, 40, 5, 1, 200),        # ... Hamming(8,4) as the inner code alone
    (64, 8, 39, 4, 5, 51),
])
def test_frame_generator_matches_oracle(oracle, product, M, cp, mod, fec0, fec1, plen):
    rng = np.random.RandomState(M + plen)
    fg = product.ofdmflexframegen(M, cp, 4)
    ref = oracle.FlexFrameGen(M, cp, 4, None, oracle.CRC_32, fec0, fec1, mod)
    for _ in range(2):                                      # the generator is reusable frame after frame
        h = bytes(rng.randint(0, 256, 8).astype(np.uint8)); pl = bytes(rng.randint(0, 256, plen).astype(np.uint8))
        want = ref.frame(h, pl)
        got = fg.frame(h, pl, mod, fec0, fec1)
        assert got.shape == want.shape
        assert relerr(got, want) <= 1e-5
        assert relerr(fg.frame(h, pl, mod, fec0, fec1, gain=0.25), want * np.float32(0.25)) <= 1e-5
    fg.close()


def test_config1_loopback_1e6_samples_matches_oracle(oracle, product):
    """BASELINE configs[0]: M=64 QPSK, ~1e6 samples, fed in ragged pieces like a radio would."""
    M, cp, tp = 64, 8, 4
    iq, sent = traffic(oracle, M, cp, tp, 78, 1200, 40, 1, 6, seed=1)
    assert 0.95e6 < len(iq) < 1.1e6
    ora = oracle.FlexFrameSync(M, cp, tp)
    ora.execute(iq)
    rx = product.ofdmflexframesync(M, cp, tp, batch_samples=65536)
    rng = np.random.RandomState(2)
    i = 0
    while i < len(iq):
        n = int(rng.randint(1, 9000))
        rx.execute(iq[i:i + n]); i += n
    rx.execute(np.zeros(8, np.complex64))                   # (samples are consumed 8 at a time)
    rx.Flush()
    assert len(ora.frames) == len(sent) == len(rx.frames)
    worst = check_frames(rx.frames, ora.frames)
    assert [(f.header, f.payload) for f in rx.frames] == sent and all(f.payload_valid for f in rx.frames)
    print("config 1: %d frames, worst framesyms error %.2e" % (len(sent), worst))
    rx.close()


@pytest.mark.parametrize("M,cp,mod,fec0,fec1,plen", [(256, 32, 27, 1, 7, (1, 900)), (64, 16, 39, 7, 7, (0, 60)), (48, 6, 40, 1, 6, (10, 200)),
                                                     (64, 8, 40, 1, 2, (1, 400)), (64, 8, 27, 1, 3, (1, 300)), (64, 8, 40, 1, 4, (0, 500)),
                                                     (128, 16, 29, 1, 5, (1, 500)), (64, 8, 40, 2, 5, (1, 200)), (48, 6, 39, 4, 3, (1, 100)),
                                                     (64, 8, 40, 5, 4, (1, 300)), (64, 8, 27, 3, 6, (1, 200))])
def test_single_synchronizer_other_schemes(oracle, product, M, cp, mod, fec0, fec1, plen):
    import torch
    iq, sent = traffic(oracle, M, cp, 4, 9, plen, mod, fec0, fec1, seed=M)
    iq = np.concatenate([iq, np.zeros((-len(iq)) % product.TILE, np.complex64)])
    ora = oracle.FlexFrameSync(M, cp, 4)
    ora.execute(iq)
    rx = product.ofdmflexframesync(M, cp, 4)
    rx.execute(torch.from_numpy(iq).cuda())                 # samples already in HBM
    rx.Flush()
    assert len(rx.frames) == len(sent)
    check_frames(rx.frames, ora.frames, rel=2e-5 if M >= 1024 else 1e-5)
    assert [(f.header, f.payload) for f in rx.frames] == sent
    rx.close()


def test_schemes_the_path_does_not_carry_are_refused_not_sent_uncoded(product):
    """a scheme id may only ever be transmitted if the bits behind it are that scheme's.  liquid's SEC-DED codes
    (8 .. 10), the other convolutional / punctured / Reed-Solomon ids and 'unknown' are refused by every transmit entry point."""
    fg = product.ofdmflexframegen(64, 8, 4)
    h = bytes(8)
    for bad in (0, 8, 9, 10, 12, 27, 31):
        with pytest.raises((RuntimeError, ValueError)):
            fg.frame(h, b"abc", 40, 1, bad)
        with pytest.raises((RuntimeError, ValueError)):
            fg.frame(h, b"abc", 40, bad, 6)
    fg.close()
    tx = product.multichanneltx(2, 64, 8, 4)
    with pytest.raises((RuntimeError, ValueError)):
        tx.generate(1, 20, fec1=9)
    tx.close()


def test_gpu_generator_to_gpu_synchronizer_and_reset(oracle, product):
    M, cp = 64, 8
    fg = product.ofdmflexframegen(M, cp, 4)
    rx = product.ofdmflexframesync(M, cp, 4)
    got = []
    rx.callback[0] = lambda h, hv, p, n, pv, st, ud: got.append((bytes(h), bytes(p), hv, pv)) or 0
    x = fg.frame(b"abcdefgh", b"payload one" * 20, gain=0.5)
    half = len(x) // 2 // product.TILE * product.TILE
    rx.execute(x[:half]); rx.Flush()
    rx.reset()                                              # mid-frame reset: the first frame is lost (ofdmflexframesync_reset)
    y = fg.frame(b"12345678", b"payload two" * 30, gain=0.5)
    rx.execute(np.concatenate([np.zeros(40, np.complex64), y, np.zeros(2 * (M + cp) + 16, np.complex64)])[: (40 + len(y) + 2 * (M + cp)) // 16 * 16])
    rx.Flush()
    assert got == [(b"12345678", b"payload two" * 30, 1, 1)]
    with pytest.raises(ValueError):
        product.multichannelrx(2, 64, 8, 4, single_channel=1)
    rx.close(); fg.close()
