"""Single synchronizer, frames separated by short random gaps (what the loopback stand-in produces), fed in
4096-sample packets with 65536-sample batches: does the GPU receiver ever lose a frame the oracle finds?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
from conftest import load_product
import oracle as O
O.build()
P = load_product()
M, cp = 64, 8
L = M + cp
fg = P.ofdmflexframegen(M, cp, 4)
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
nf = 60
bad = 0
for seed in range(nseeds):
    rng = np.random.RandomState(1000 + seed)
    parts = [np.zeros(int(rng.randint(0, 500)), np.complex64)]
    for f in range(nf):
        h = bytes([f >> 8, f & 0xff]) + bytes(rng.randint(0, 256, 6).astype(np.uint8))
        x = fg.frame(h, bytes(rng.randint(0, 256, 500).astype(np.uint8)), 27, 6, 1, gain=0.2512)
        parts += [x, x[-L:], np.zeros(64 * int(rng.randint(0, 4)), np.complex64)]
    parts.append(np.zeros(4 * L, np.complex64))
    iq = np.concatenate(parts)
    iq = np.concatenate([iq, np.zeros((-len(iq)) % 8, np.complex64)])
    rx = P.ofdmflexframesync(M, cp, 4, batch_samples=65536)
    pk = int(rng.choice([4096, 1000, 72, 0]))
    i = 0
    while i < len(iq):                                      # pk = 0: ragged packets, like a receiver that keeps up with the sender
        step = pk if pk else int(rng.choice([72, 64, 144, 8, 3, 216]))
        rx.execute(iq[i:i + step]); i += step
    rx.Flush()
    gid = [(f.header[0] << 8) | f.header[1] for f in rx.frames if f.header_valid and f.payload_valid]
    rx.close()
    if gid != list(range(nf)):
        ora = O.FlexFrameSync(M, cp, 4); ora.execute(iq)
        oid = [(f.header[0] << 8) | f.header[1] for f in ora.frames if f.header_valid and f.payload_valid]
        print("seed", seed, "packet", pk, "gpu missing", sorted(set(range(nf)) - set(gid)), "oracle missing", sorted(set(range(nf)) - set(oid)), flush=True)
        if oid != gid:
            bad += 1
            np.save("gpurun_out/gap_hunt_seed%d.npy" % seed, iq)
print("seeds", nseeds, "disagreements with the oracle", bad)
