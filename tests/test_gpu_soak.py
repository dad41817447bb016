"""Randomised soak of the whole receive path against the CPU oracle: random frame lengths, modems, codes, gains,
carrier offsets and noise; random Execute() piece sizes; three passes per handle (the later ones with the scout's
speculative acquisition predicting from the earlier ones).  scratch/soak.py is the long-running version."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_random_traffic_matches_oracle(oracle, product, seed):
    import torch
    rng = np.random.RandomState(seed)
    iters = 6
    worst, nframes, nbad = 0.0, 0, 0
    worst, nframes, nbad = 0.0, 0, 0
    for it in range(iters):
        N = int(rng.choice([1, 2, 4, 8])); M, cp = [(64, 8), (64, 16), (128, 16), (256, 32)][rng.randint(4)]
        tx = product.multichanneltx(N, M, cp, 4)
        parts = []
        for seg in range(rng.randint(1, 4)):
            mod = int(rng.choice([39, 40, 27, 29])); fec1 = int(rng.choice([1, 6, 7, 11]))
            plen = int(rng.randint(0, 600)); nf = int(rng.randint(1, 5))
            x, _ = tx.generate(nf, plen, mod=mod, fec1=fec1, seed=int(rng.randint(1 << 30)), gain=float(rng.uniform(0.2, 1.0)) / N)
            parts.append(x)
        tx.close()
        iq = torch.cat(parts)
        n = int(iq.numel()) // (32 * N) * (32 * N)
        x = iq[:n].cpu().numpy()
        t = np.arange(n)
        snr = rng.uniform(22, 40)
        sig = np.sqrt(np.mean(np.abs(x) ** 2))
        x = (x * np.exp(1j * (rng.uniform(-3e-4, 3e-4) * t + rng.uniform(0, 6.28))) +
             sig * 10 ** (-snr / 20) / np.sqrt(2) * (rng.randn(n) + 1j * rng.randn(n))).astype(np.complex64)
        o = oracle.MultiChannelRx(N, M, cp, 4); o.execute(x)
        rx = product.multichannelrx(N, M, cp, 4, max_payload_len=640, batch_samples=32 * N * int(rng.randint(8, 200)))
        xd = torch.from_numpy(x).cuda()
        seen = 0
        for rep in range(3):
            if rep: rx.Reset()
            i = 0
            while i < n:
                step = 32 * N * int(rng.randint(1, 400))
                rx.Execute(xd[i:min(i + step, n)]); i += step
            rx.Flush()
            got = rx.frames[seen:]; seen = len(rx.frames)
            gk = sorted([(f.channel, f.end_sample) for f in got])
            if len(got) != len(o.frames):
                nbad += 1; print("iter", it, "rep", rep, "frame count", len(got), len(o.frames)); continue
            by = {}
            for f in o.frames: by.setdefault(f.channel, []).append(f)
            gy = {}
            for f in got: gy.setdefault(f.channel, []).append(f)
            for ch in by:
                for fg, fo in zip(gy.get(ch, []), by[ch]):
                    nframes += 1
                    if (fg.header_valid, fg.payload_valid, fg.header, fg.payload) != (fo.header_valid, fo.payload_valid, fo.header, fo.payload):
                        nbad += 1; print("iter", it, "rep", rep, "ch", ch, "mismatch", fg, fo)
                    elif len(fo.framesyms):
                        e = float(np.max(np.abs(fg.framesyms - fo.framesyms)) / np.max(np.abs(fo.framesyms)))
                        worst = max(worst, e)
        rx.close()
    assert nframes > 20 and nbad == 0, (nframes, nbad)
    assert worst <= 1e-5, worst


@pytest.mark.parametrize("seed", [71, 72])
def test_low_snr_contract(oracle, product, seed):
    """Where "bit exact" ends: at 1-9 dB most payloads are beyond their code's reach.  The contract
    the suite holds the GPU path to there, frame for frame against the oracle on the same samples:
      * the same frames are detected, per channel and in order, with the same header_valid / payload_valid flags;
      * headers are identical (valid or not: they are hard decisions behind a Golay code);
      * every frame EITHER side delivers as valid has identical payload bytes;
      * payload bytes may differ only in frames BOTH sides flag invalid -- an 8-bit soft decision of a QAM bit is a quantised function of
        float distances, and where it sits on a quantisation boundary the two pipelines' rounding (the 1e-5 the symbols are held to) can
        land it either side -- and such frames are counted and bounded: scratch/soak.py found 5 of 4 389 (0.11 %) on seeds 71-73; here
        at most 1 % of the frames and at most 4 bytes per frame, BPSK / QPSK frames never."""
    import torch
    rng = np.random.RandomState(seed)
    nframes = ndiff = nvalid = 0
    worst_bytes = 0
    for it in range(5):
        N = int(rng.choice([1, 2, 4, 8])); M, cp = [(64, 8), (64, 16), (128, 16), (256, 32)][rng.randint(4)]
        tx = product.multichanneltx(N, M, cp, 4)
        parts = []
        for seg in range(rng.randint(1, 4)):
            mod = int(rng.choice([39, 40, 27, 29])); fec1 = int(rng.choice([1, 6, 7, 11]))
            plen = int(rng.randint(0, 600)); nf = int(rng.randint(1, 5))
            x, _ = tx.generate(nf, plen, mod=mod, fec1=fec1, seed=int(rng.randint(1 << 30)), gain=float(rng.uniform(0.2, 1.0)) / N)
            parts.append(x)
        tx.close()
        iq = torch.cat(parts)
        n = int(iq.numel()) // (32 * N) * (32 * N)
        x = iq[:n].cpu().numpy()
        t = np.arange(n)
        snr = rng.uniform(1, 9)
        sig = np.sqrt(np.mean(np.abs(x) ** 2))
        x = (x * np.exp(1j * (rng.uniform(-3e-4, 3e-4) * t + rng.uniform(0, 6.28))) +
             sig * 10 ** (-snr / 20) / np.sqrt(2) * (rng.randn(n) + 1j * rng.randn(n))).astype(np.complex64)
        o = oracle.MultiChannelRx(N, M, cp, 4); o.execute(x)
        rx = product.multichannelrx(N, M, cp, 4, max_payload_len=640, batch_samples=32 * N * int(rng.randint(8, 200)), conv_scratch=1)
        xd = torch.from_numpy(x).cuda()
        seen = 0
        for rep in range(2):
            if rep: rx.Reset()
            i = 0
            while i < n:
                step = 32 * N * int(rng.randint(1, 400))
                rx.Execute(xd[i:min(i + step, n)]); i += step
            rx.Flush()
            got = rx.frames[seen:]; seen = len(rx.frames)
            assert len(got) == len(o.frames), (it, rep, len(got), len(o.frames))
            by, gy = {}, {}
            for f in o.frames: by.setdefault(f.channel, []).append(f)
            for f in got: gy.setdefault(f.channel, []).append(f)
            assert sorted(by) == sorted(gy)
            for ch in by:
                assert len(by[ch]) == len(gy[ch])
                for fg, fo in zip(gy[ch], by[ch]):
                    nframes += 1
                    assert (fg.header_valid, fg.payload_valid) == (fo.header_valid, fo.payload_valid), (it, rep, ch)
                    assert fg.header == fo.header, (it, rep, ch)
                    if fo.payload_valid or fg.payload_valid:
                        nvalid += 1
                        assert fg.payload == fo.payload, (it, rep, ch)
                    elif fg.payload != fo.payload:
                        assert len(fg.payload) == len(fo.payload) and fo.header_valid and fo.mod_scheme in (27, 29), (it, rep, ch, fo.mod_scheme)
                        nb = sum(1 for a, b in zip(fg.payload, fo.payload) if a != b)
                        ndiff += 1; worst_bytes = max(worst_bytes, nb)
        rx.close()
    print("low-SNR soak seed %d: %d frames compared, %d delivered valid by either side (all identical), %d invalid QAM frames with different bytes "
          "(at most %d bytes each)" % (seed, nframes, nvalid, ndiff, worst_bytes))
    assert nframes >= 40
    assert ndiff <= max(1, nframes // 100) and worst_bytes <= 4, (nframes, ndiff, worst_bytes)
