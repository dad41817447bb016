"""A/B of libmcrx_hip.so builds (MCRX_LIB): K = 1024 channelizer alone -- time, and output bit-compared with a reference dump.
usage: MCRX_LIB=... python scratch/chan_ab.py <dump.pt> [write]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_product
prod = load_product()
N, K = 512, 1024
nblocks = 101376 * 2
g = torch.Generator(device="cuda"); g.manual_seed(7)
x = torch.view_as_complex(torch.randn(nblocks * K, 2, device="cuda", generator=g))
out = torch.empty(nblocks // 8 * N * 8, dtype=torch.complex64, device="cuda")
rx = prod.multichannelrx(N, 64, 8, 4)
for _ in range(5):
    rx.channelize(x, nblocks, 12345, out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(30):
    rx.channelize(x, nblocks, 12345, out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 30
small = 4096
res = {}
for groups in (1, 8):
    o2 = torch.zeros(small // 8 * N * 8, dtype=torch.complex64, device="cuda")
    rx.channelize(x[:small * K], small, 777, o2, groups=groups)
    torch.cuda.synchronize()
    res[groups] = o2.cpu()
path = sys.argv[1]
if len(sys.argv) > 2:
    torch.save(res, path); same = "written"
else:
    ref = torch.load(path)
    same = all(torch.equal(torch.view_as_real(ref[k]), torch.view_as_real(res[k])) for k in ref)
    if not same:
        same = "no, max |diff| / max |ref| = " + ", ".join("%.3g" % float((ref[k] - res[k]).abs().max() / ref[k].abs().max()) for k in ref)
print("%s: %.4f ms per %d blocks (%.2f TB/s algorithmic, %.1f %% of 8 TB/s)  bit-identical to reference: %s" %
      (os.environ.get("MCRX_LIB", "default"), dt * 1e3, nblocks, 12.0 * nblocks * K / dt / 1e12, 12.0 * nblocks * K / dt / 8e10, same), flush=True)
rx.close()
