"""time the K = 1024 channelizer kernel alone through the stage-level call, for a front end (argv[1]) under MCRX_LIB"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
prod = load_product()
fe = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N, K = 512, 1024
nblocks = 202752
x = (torch.randn(nblocks * K, device="cuda") + 1j * torch.randn(nblocks * K, device="cuda")).to(torch.complex64)
out = torch.empty(nblocks * N, dtype=torch.complex64, device="cuda")
rx = prod.multichannelrx(N, 64, 8, 4, front_end=fe)
for _ in range(5):
    rx.channelize(x, nblocks, 0, out)
torch.cuda.synchronize()
ts = []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20):
        rx.channelize(x, nblocks, 0, out)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 20)
dt = min(ts)
print("%s fe=%d: %.4f ms  (%.2f TB/s on 12 B/sample = %.1f %%)" % (os.path.basename(os.environ.get("MCRX_LIB", "product")), fe, dt * 1e3, 12.0 * nblocks * K / dt / 1e12, 12.0 * nblocks * K / dt / 8e10), flush=True)
rx.close()
