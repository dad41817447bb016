#!/usr/bin/env python3
"""Timing mode B of SURVEY 8(d): host-fed bulk Execute(buf, n) including the H2D copies (PCIe bound)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from __graft_entry__ import load_product
prod = load_product()
N, M, cp = 512, 64, 8
tx = prod.multichanneltx(N, M, cp, 4)
iq, sent = tx.generate(8, 1200, seed=0xC0FFEE)
torch.cuda.synchronize(); tx.close()
x = iq.cpu().numpy()
n = len(x) // (16 * N) * (16 * N)
x = x[:n]
rx = prod.multichannelrx(N, M, cp, 4, max_payload_len=1200, max_frames=N * 8 * 4 + 64)
rx.Execute(x); rx.Flush()                      # warm up (allocations, predictions)
seen = len(rx.frames)
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    rx.Execute(x)
rx.Flush()
dt = time.perf_counter() - t0
ok = sum(1 for f in rx.frames[seen:] if f.payload_valid)
print(json.dumps({"mode": "B (host buffers, H2D included, callbacks delivered)", "samples": n * reps, "seconds": round(dt, 4),
                  "Msamples_per_s": round(n * reps / dt / 1e6, 1), "GB_per_s_over_pcie": round(n * reps * 8 / dt / 1e9, 2), "frames_valid": ok}))
