"""Decode the recording of a fullduplex loopback run with the oracle and the GPU synchronizer (offline)."""
import sys, os, re
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
from conftest import load_product
import oracle as O
O.build()
P = load_product()
iq = np.fromfile("/tmp/fdx.bin", np.complex64)
print("recorded samples", len(iq))
live = [int(x) for x in re.findall(r"rx packet id:\s+(\d+)", open("/tmp/fdx.out").read())]
ora = O.FlexFrameSync(64, 8, 4); ora.execute(iq)
oid = [(f.header[0] << 8) | f.header[1] for f in ora.frames if f.header_valid]
x = np.concatenate([iq, np.zeros((-len(iq)) % 8, np.complex64)])
rx = P.ofdmflexframesync(64, 8, 4); rx.execute(x); rx.Flush()
gid = [(f.header[0] << 8) | f.header[1] for f in rx.frames if f.header_valid]
rx2 = P.ofdmflexframesync(64, 8, 4, batch_samples=65536)
for i in range(0, len(x), 4096):
    rx2.execute(x[i:i + 4096])
rx2.Flush()
g2 = [(f.header[0] << 8) | f.header[1] for f in rx2.frames if f.header_valid]
full = set(range(200))
print("live", len(live), "missing", sorted(full - set(live)))
print("oracle offline", len(oid), "missing", sorted(full - set(oid)))
print("gpu offline (one call)", len(gid), "missing", sorted(full - set(gid)))
print("gpu offline (4096-sample packets, 65536 batches)", len(g2), "missing", sorted(full - set(g2)))
np.save("gpurun_out/fdx_iq_head.npy", iq[:10])
