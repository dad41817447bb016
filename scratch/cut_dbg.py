"""minimal form of the soak failure: one segment of (mod 29, Golay) frames, N=4, M=64, pushed in pieces that cut a payload twice"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_product, load_oracle
import torch
product, oracle = load_product(), load_oracle()
N, M, cp = 4, 64, 8
tx = product.multichanneltx(N, M, cp, 4)
x, _ = tx.generate(2, 208, mod=29, fec1=7, seed=5, gain=0.5 / N)
tx.close()
n = int(x.numel()) // (16 * N) * (16 * N)
xs = x[:n].cpu().numpy()
o = oracle.MultiChannelRx(N, M, cp, 4); o.execute(xs)
print("oracle", [(f.channel, f.payload_valid) for f in o.frames])
cuts = [int(c) for c in sys.argv[1:]] or [1496, 1824, 2016, 3232, 3440]
rx = product.multichannelrx(N, M, cp, 4, max_payload_len=640)
i = 0
for c in cuts + [n // (2 * N)]:
    j = min(c, n // (2 * N)) * 2 * N
    if j > i:
        rx.Execute(x[i:j]); i = j
rx.Flush()
print("gpu   ", [(f.channel, f.payload_valid, f.end_sample) for f in rx.frames])
rx.close()
