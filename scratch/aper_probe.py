"""ragged traffic through a SERIAL handle (every kernel alone): where does the time go?  MCRX_* env knobs apply."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
from __graft_entry__ import load_product
prod = load_product()
N, M, cp, taper = 512, 64, 8, 4
serial = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nb = 202752
tx = prod.multichanneltx(N, M, cp, taper)
slabs = []
for i in range(3):
    d, s, _ = tx.generate_ragged(nb, len_lo=64, len_hi=1200, gap_max=3, long_every=8, long_max=184, seed=0xA9E210 + 104729 * i)
    slabs.append(d)
tx.close()
nfr = sum(len(c) for c in s)
rx = prod.multichannelrx(N, M, cp, taper, serial=serial, max_payload_len=1200, max_frames=2 * nfr)
for _ in range(3):
    for d in slabs: rx.Execute(d); rx.Discard()
torch.cuda.synchronize(); rx.kernel_stats(reset=True); rx.spec_stats(reset=True)
t0 = time.perf_counter()
for _ in range(6):
    for d in slabs: rx.Execute(d); rx.Discard()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
w, a = rx.spec_stats()
print("serial" if serial else "pipelined", "ragged: %.1f Gsample/s, %.3f ms per slab;" % (18 * nb * 1024 / dt / 1e9, dt / 18 * 1e3),
      "walked %d adopted %d;" % (w, a), {k: round(v[0] / max(v[1], 1), 4) for k, v in rx.kernel_stats().items()}, "frames/slab ~%d" % nfr)
rx.close()
