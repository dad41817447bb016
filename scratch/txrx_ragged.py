"""Ragged class-interface traffic -> GPU RX and oracle RX: do both find the same frames?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, "tests")
from conftest import load_product
import oracle as O
O.build()
product = load_product()
N, M, cp = 4, 64, 8
L = M + cp
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
tx = product.multichanneltx(N, M, cp, 4, max_payload_len=300)
rng = np.random.RandomState(seed)
sent = {}; pid = [0] * N; blocks = []; starts = {}
for per in range(160):
    for c in range(N):
        if pid[c] < 4 and tx.IsChannelReadyForData(c) and rng.rand() < 0.7:
            pl = bytes(rng.randint(0, 256, int(rng.randint(1, 300))).astype(np.uint8))
            h = bytes([0, pid[c], c]) + bytes(rng.randint(0, 256, 5).astype(np.uint8))
            assert tx.UpdateData(c, h, pl)
            sent[(c, pid[c])] = (h, pl); starts[(c, pid[c])] = per; pid[c] += 1
    blocks.extend(tx.GenerateSamples().copy() for _ in range(L))
iq = (np.concatenate(blocks) / np.float32(N)).astype(np.complex64)
n = len(iq) // (16 * N) * (16 * N)
rx = product.multichannelrx(N, M, cp, 4)
rx.Execute(torch.from_numpy(iq[:n]).cuda()); rx.Flush()
orx = O.MultiChannelRx(N, M, cp, 4)
orx.execute(iq[:n])
g = sorted((f.channel, f.header[1], f.payload_valid) for f in rx.frames)
o = sorted((f.channel, f.header[1], f.payload_valid) for f in orx.frames)
print("sent", len(sent), "gpu", len(g), "oracle", len(o))
print("missing on gpu:", [(k, starts[k]) for k in sent if (k[0], k[1], True) not in g])
print("missing on oracle:", [(k, starts[k]) for k in sent if (k[0], k[1], True) not in o])
print("starts", sorted((k, v) for k, v in starts.items()))
