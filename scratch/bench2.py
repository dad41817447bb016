#!/usr/bin/env python3
"""Experiment: upper bound of cross-step overlap -- two independent receivers on two streams, alternating."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
prod = load_product()
from liquid_usrp_amd import sharding
dev = torch.device("cuda", 0)
N, M, cp, taper, reps = 512, 64, 8, 4, 8
K = 2 * N
tx = prod.multichanneltx(N, M, cp, taper)
d_iq, sent = tx.generate(reps, 1200, seed=0xC0FFEE, device=dev)
torch.cuda.synchronize(); tx.close()
T = int(d_iq.numel()) // K
ntiles = T // 8
NRX = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rxs, outs, streams = [], [], []
for i in range(NRX):
    rxs.append(prod.multichannelrx(N, M, cp, taper, max_payload_len=1200, max_frames=N * reps + 64))
    outs.append(torch.empty(ntiles * N * 8, dtype=torch.complex64, device=dev))
    streams.append(torch.cuda.Stream())
def step(k):
    i = k % NRX
    with torch.cuda.stream(streams[i]):
        sharding.step(rxs[i], d_iq, T, 0, 1, None, outs[i], outs[i], halo=None, stream=streams[i])
for k in range(4): step(k)
torch.cuda.synchronize()
steps = 20
t0 = time.perf_counter()
for k in range(steps): step(k)
torch.cuda.synchronize()
el = time.perf_counter() - t0
ok = []
for rx in rxs:
    rx.Flush()
    fr = rx.frames
    ok.append((len(fr), sum(1 for f in fr if f.payload_valid and sent[f.channel][(f.header[0] << 8) | f.header[1]] == (f.header, f.payload))))
print(json.dumps({"nrx": NRX, "ms_per_step": el / steps * 1e3, "Msps": T * K * steps / el / 1e6, "frames_ok": ok}))
