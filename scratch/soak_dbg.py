"""debug form of tests/test_gpu_soak.py: prints the configuration, the piece boundaries and which frames differ"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_product, load_oracle
import torch
product, oracle = load_product(), load_oracle()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 11
rng = np.random.RandomState(seed)
for it in range(6):
    N = int(rng.choice([1, 2, 4, 8])); M, cp = [(64, 8), (64, 16), (128, 16), (256, 32)][rng.randint(4)]
    tx = product.multichanneltx(N, M, cp, 4)
    parts, segs = [], []
    for seg in range(rng.randint(1, 4)):
        mod = int(rng.choice([39, 40, 27, 29])); fec1 = int(rng.choice([1, 6, 7]))
        plen = int(rng.randint(0, 600)); nf = int(rng.randint(1, 5))
        x, _ = tx.generate(nf, plen, mod=mod, fec1=fec1, seed=int(rng.randint(1 << 30)), gain=float(rng.uniform(0.2, 1.0)) / N)
        parts.append(x); segs.append((mod, fec1, plen, nf, int(x.numel()) // (2 * N)))
    tx.close()
    iq = torch.cat(parts)
    n = int(iq.numel()) // (16 * N) * (16 * N)
    x = iq[:n].cpu().numpy()
    t = np.arange(n)
    snr = rng.uniform(22, 40)
    sig = np.sqrt(np.mean(np.abs(x) ** 2))
    x = (x * np.exp(1j * (rng.uniform(-3e-4, 3e-4) * t + rng.uniform(0, 6.28))) +
         sig * 10 ** (-snr / 20) / np.sqrt(2) * (rng.randn(n) + 1j * rng.randn(n))).astype(np.complex64)
    o = oracle.MultiChannelRx(N, M, cp, 4); o.execute(x)
    bs = 16 * N * int(rng.randint(8, 200))
    rx = product.multichannelrx(N, M, cp, 4, max_payload_len=640, batch_samples=bs)
    if os.environ.get("ONLY_ITER") and int(os.environ["ONLY_ITER"]) != it:
        for rep in range(3):
            i = 0
            while i < n:
                i += 16 * N * int(rng.randint(1, 400))
        continue
    print("iter", it, "N", N, "M", M, "cp", cp, "segs", segs, "blocks", n // (2 * N), "batch", bs // (2 * N), "snr %.1f" % snr, flush=True)
    xd = torch.from_numpy(x).cuda()
    seen = 0
    for rep in range(3):
        if rep: rx.Reset()
        i = 0; cuts = []
        while i < n:
            step = 16 * N * int(rng.randint(1, 400))
            rx.Execute(xd[i:min(i + step, n)]); i += step; cuts.append(min(i, n) // (2 * N))
        rx.Flush()
        got = rx.frames[seen:]; seen = len(rx.frames)
        if len(got) != len(o.frames):
            print("  rep", rep, "frame count", len(got), len(o.frames)); continue
        by = {}
        for f in o.frames: by.setdefault(f.channel, []).append(f)
        gy = {}
        for f in got: gy.setdefault(f.channel, []).append(f)
        base = got[0].end_sample - 0 if got else 0
        nb = 0
        for ch in by:
            for k, (fg, fo) in enumerate(zip(gy.get(ch, []), by[ch])):
                if (fg.header_valid, fg.payload_valid, fg.header, fg.payload) != (fo.header_valid, fo.payload_valid, fo.header, fo.payload):
                    nb += 1
                    if ch == 0: print("  rep", rep, "ch0 frame", k, "end", fg.end_sample, "len", len(fg.payload), "mod", fg.mod_scheme, "fec1", fg.fec1, "pv", fg.payload_valid, fo.payload_valid,
                                      "nbytes diff", sum(a != b for a, b in zip(fg.payload, fo.payload)),
                                      "nsyms", len(fg.framesyms), "bad syms at", np.nonzero(np.abs(fg.framesyms - fo.framesyms) > 1e-3)[0][:6], "n", int(np.sum(np.abs(fg.framesyms - fo.framesyms) > 1e-3)))
        print("  rep", rep, "bad", nb, "of", len(got), "cuts", cuts[:12], flush=True)
    rx.close()
