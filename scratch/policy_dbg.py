import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch
from __graft_entry__ import load_product, load_oracle
product, oracle = load_product(), load_oracle(); oracle.build()
N, M, cp = 8, 64, 8
L = M + cp
tx = product.multichanneltx(N, M, cp, 4)
a, _ = tx.generate(40, 90, seed=5)
b, _, _ = tx.generate_ragged(L * 2600 // 8 * 8, len_lo=10, len_hi=200, gap_max=3, long_every=6, long_max=30, seed=6)
c, _ = tx.generate(40, 60, seed=7)
tx.close()
iq = torch.cat([a, b, c]); n = int(iq.numel()) // (16 * N) * (16 * N); x = iq[:n].cpu().numpy()
print("blocks a,b,c", a.numel() // 16, b.numel() // 16, c.numel() // 16)
ora = oracle.MultiChannelRx(N, M, cp, 4); ora.execute(x)
for name, step in (("one push", n), ("small pushes", 16 * N * 26)):
    rx = product.multichannelrx(N, M, cp, 4, max_payload_len=200)
    for i in range(0, n, step): rx.Execute(iq[i:min(i + step, n)])
    rx.Flush()
    by = {}
    for f in ora.frames: by.setdefault(f.channel, []).append(f)
    byg = {}
    for f in rx.frames: byg.setdefault(f.channel, []).append(f)
    worst = []
    for ch in by:
        for k, (fg, fo) in enumerate(zip(byg[ch], by[ch])):
            if len(fo.framesyms):
                e = float(np.max(np.abs(fg.framesyms - fo.framesyms)) / np.max(np.abs(fo.framesyms)))
                worst.append((e, ch, k, fo.payload_len if hasattr(fo, "payload_len") else len(fo.payload), fg.end_sample, fo.header_valid, fo.payload_valid, float(np.max(np.abs(fo.framesyms)))))
    worst.sort(reverse=True)
    print(name, "frames", len(rx.frames), len(ora.frames), "worst:", worst[:6])
    rx.close()
