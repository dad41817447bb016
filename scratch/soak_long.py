#!/usr/bin/env python3
"""Soak for the long-push path (slot windows, ranked chains): few channels, hundreds of short ragged frames per channel and push, noise,
several pushes with the device idle between them (so that the frame counts size the next push's slots) -- every frame compared with
the CPU oracle: flags, bytes, order."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from __graft_entry__ import load_product, load_oracle
from test_gpu_parity import match_frames
prod, ora = load_product(), load_oracle()
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
tot = bad = 0
for it in range(iters):
    rng = np.random.RandomState(seed0 * 1000 + it)
    N = int(rng.choice([1, 2, 4, 8])); M, cp = 64, 8
    L = M + cp
    tx = prod.multichanneltx(N, M, cp, 4)
    periodic = bool(rng.randint(2))
    if periodic:
        x, _ = tx.generate(int(rng.randint(150, 500)), int(rng.randint(8, 60)), seed=int(rng.randint(1 << 30)))
    else:
        x, _, _ = tx.generate_ragged(L * int(rng.randint(3000, 9000)) // 16 * 16, len_lo=8, len_hi=int(rng.randint(40, 160)), gap_max=3, long_every=int(rng.randint(5, 40)), long_max=60, seed=int(rng.randint(1 << 30)))
    tx.close()
    pushes = int(rng.randint(2, 5))
    n = int(x.numel()) // (32 * N * pushes) * (32 * N * pushes)
    g = torch.Generator(device="cuda"); g.manual_seed(int(rng.randint(1 << 30)))
    sig = float(x[:n].abs().pow(2).mean().sqrt())
    y = x[:n] + (sig * 10 ** (-25 / 20) / 2 ** 0.5) * torch.view_as_complex(torch.randn(n, 2, generator=g, device="cuda"))
    o = ora.MultiChannelRx(N, M, cp, 4); o.execute(y.cpu().numpy())
    rx = prod.multichannelrx(N, M, cp, 4, max_payload_len=200, max_frames=len(o.frames) + 64)
    # the stream twice over: the second pass runs with the frame counts of the first
    for rep in range(2):
        for i in range(pushes):
            rx.Execute(y[i * (n // pushes):(i + 1) * (n // pushes)]); torch.cuda.synchronize()
        if rep == 0:
            rx.Flush(); first = list(rx.frames); rx.frames.clear(); rx.Reset(); rx.spec_stats(reset=True)
    rx.Flush()
    walked, adopted = rx.spec_stats()
    for name, fr in (("first pass", first), ("second pass", rx.frames)):
        pairs = list(match_frames(fr, o.frames))
        ok = len(pairs) == len(o.frames) == len(fr) and all((a.header, a.payload, a.header_valid, a.payload_valid) == (b.header, b.payload, b.header_valid, b.payload_valid) for a, b in pairs)
        tot += len(o.frames); bad += 0 if ok else 1
        if not ok:
            print("MISMATCH", seed0, it, name, N, periodic, len(fr), len(o.frames), flush=True)
    print(json.dumps({"seed": seed0, "it": it, "N": N, "periodic": periodic, "pushes": pushes, "frames": len(o.frames), "frames_per_channel_and_push": round(len(o.frames) / N / pushes, 1), "walked": walked, "adopted": adopted}), flush=True)
    rx.close()
print(json.dumps({"seed": seed0, "frames_compared": tot, "mismatching_passes": bad}))
