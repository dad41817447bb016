"""512 channels at the reference applications' default numerology (M = 48, cp = 6): lean segment waves (scout_build 0) against the general
state machine's (scout_build 2), periodic and ragged traffic, one continuous stream each"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from __graft_entry__ import load_product
prod = load_product()
N, M, cp, frames, plen = 512, int(sys.argv[1]) if len(sys.argv) > 1 else 48, 0, 16, 1200
cp = M // 8
tx = prod.multichanneltx(N, M, cp, 4)
slabs = [tx.generate(frames, plen, seed=50 + i)[0] for i in range(2)]
nb = int(slabs[0].numel()) // (2 * N)
rag = [tx.generate_ragged(nb, len_lo=64, len_hi=plen, gap_max=3, long_every=8, long_max=184, seed=70 + i)[0] for i in range(2)] if (M & (M - 1)) == 0 else None
tx.close()
torch.cuda.synchronize()
for name, data in (("periodic", slabs), ("ragged", rag)):
    if data is None: continue
    for rep in range(2):
        for sb in (0, 2):
            rx = prod.multichannelrx(N, M, cp, 4, max_payload_len=plen, max_frames=N * frames * 2 + 64, scout_build=sb)
            for _ in range(6):
                for d in data: rx.Execute(d); rx.Discard()
            torch.cuda.synchronize(); rx.spec_stats(reset=True); rx.kernel_stats(reset=True)
            t0 = time.perf_counter()
            for _ in range(10):
                for d in data: rx.Execute(d); rx.Discard()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            w, a = rx.spec_stats()
            print("M=%d %s scout_build %d: %.1f Gsample/s  walked %d adopted %d  %s" % (M, name, sb, sum(int(d.numel()) for d in data) * 10 / dt / 1e9, w, a,
                  {k: round(v[0] / max(v[1], 1), 3) for k, v in rx.kernel_stats().items()}))
            rx.close()
