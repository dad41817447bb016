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
