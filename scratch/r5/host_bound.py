"""configs[1] (8 channels, 100 frames per push): is a push's time the host's enqueueing or the device's work?
host = wall time of the Execute + Discard calls alone (no synchronisation inside the loop), device = the loop's end-to-end time"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
prod = load_product()
N, M, cp, frames, plen = 8, 64, 8, int(os.environ.get("FRAMES", "100")), 1200
tx = prod.multichanneltx(N, M, cp, 4)
d, sent = tx.generate(frames, plen, mod=40, fec1=6, seed=3)
tx.close()
rx = prod.multichannelrx(N, M, cp, 4, max_payload_len=plen, max_frames=N * frames + 64)
for _ in range(8):
    rx.Execute(d); rx.Discard()
torch.cuda.synchronize()
n = 40
t0 = time.perf_counter()
th = 0.0
for _ in range(n):
    a = time.perf_counter()
    rx.Execute(d); rx.Discard()
    th += time.perf_counter() - a
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("frames/ch %d: host enqueue %.3f ms per push, loop %.3f ms per push, to the last kernel's end %.3f ms per push -> %.1f Gsample/s"
      % (frames, th / n * 1e3, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, d.numel() * n / (t2 - t0) / 1e9))
