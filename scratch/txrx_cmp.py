"""Decode the teed loopback stream of the txrx app with the oracle and with the GPU receiver; compare."""
import sys, os, re
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
from conftest import load_product
import oracle as O
O.build()
P = load_product()
N, M, cp = 4, 64, 8
iq = np.fromfile("/tmp/tee.bin", np.complex64)
n = len(iq) // (16 * N) * (16 * N)
iq = iq[:n]
print("samples", n)
orx = O.MultiChannelRx(N, M, cp, 4); orx.execute(iq)
rx = P.multichannelrx(N, M, cp, 4, max_frames=20000)
step = 1 << 20
for i in range(0, n, step * 16):
    rx.Execute(torch.from_numpy(iq[i:i + step * 16]).cuda())
rx.Flush()
key = lambda f: (f.channel, f.header_valid, f.payload_valid, f.header if f.header_valid else b"", len(f.payload) if f.header_valid else 0)
go = sorted(map(key, orx.frames)); gg = sorted(map(key, rx.frames))
print("oracle frames", len(go), "valid", sum(1 for k in go if k[2]), "gpu frames", len(gg), "valid", sum(1 for k in gg if k[2]))
print("identical", go == gg)
sent = open("gpurun_out/txrx.out").read()
ids = set(int(p) for p in re.findall(r"transmitting packet\s+(\d+)", sent))
oids = set((k[3][0] << 8) | k[3][1] for k in go if k[2])
print("sent", len(ids), "oracle decoded ids", len(oids), "missing in oracle", len(ids - oids), sorted(ids - oids)[:40])
