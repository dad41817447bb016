"""time the K=1024 channelizer kernel alone (stage-level call) for a list of MCRX_ABLATE values"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_product
prod = load_product()
N, K = 512, 1024
nblocks = 101376
x = (torch.randn(nblocks * K, device="cuda") + 1j * torch.randn(nblocks * K, device="cuda")).to(torch.complex64)
out = torch.empty(nblocks // 8 * N * 8, dtype=torch.complex64, device="cuda")
for ab in [int(v) for v in sys.argv[1:]] or [0]:
    os.environ["MCRX_ABLATE"] = str(ab)
    rx = prod.multichannelrx(N, 64, 8, 4)
    for _ in range(5):
        rx.channelize(x, nblocks, 0, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        rx.channelize(x, nblocks, 0, out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30
    print("ablate %4d: %.4f ms  (%.2f TB/s algorithmic)" % (ab, dt * 1e3, 12.0 * nblocks * K / dt / 1e12), flush=True)
    rx.close()
