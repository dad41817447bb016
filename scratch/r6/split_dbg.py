import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_product
prod = load_product()
for N, fe, H in ((64, 1, 27), (64, 0, 13), (256, 1, 27), (512, 1, 27), (16, 1, 27)):
    K = 2 * N
    nblocks = 96
    rng = np.random.RandomState(100 + N)
    x = (rng.randn(nblocks * K) + 1j * rng.randn(nblocks * K)).astype(np.complex64)
    rx = prod.multichannelrx(N, 64, 8, 4, front_end=fe)
    d_x = torch.from_numpy(x).cuda()
    d_out = torch.zeros(nblocks * N, dtype=torch.complex64, device="cuda")
    rx.channelize(d_x, nblocks, 0, d_out); torch.cuda.synchronize()
    got = prod.tiles_to_channels(d_out, N).T
    d_out2 = torch.zeros(nblocks * N, dtype=torch.complex64, device="cuda")
    rx.channelize(d_x, nblocks, 0, d_out2); torch.cuda.synchronize()
    print("N", N, "fe", fe, "repeat identical:", np.array_equal(prod.tiles_to_channels(d_out2, N).T, got))
    for h in (16, 32, 48, 64, 80):
        d_a = torch.zeros(h * N, dtype=torch.complex64, device="cuda")
        d_b = torch.zeros((nblocks - h) * N, dtype=torch.complex64, device="cuda")
        rx.channelize(d_x[:h * K], h, 0, d_a)
        lo = max(h - H, 0)
        halo = d_x[(h - H) * K:h * K] if h >= H else torch.cat([torch.zeros((H - h) * K, dtype=torch.complex64, device="cuda"), d_x[:h * K]])
        rx.channelize(d_x[h * K:], nblocks - h, h * K, d_b, d_halo=halo)
        torch.cuda.synchronize()
        g2 = np.concatenate([prod.tiles_to_channels(d_a, N).T, prod.tiles_to_channels(d_b, N).T])
        bad = np.argwhere(g2 != got)
        print("  h", h, "differing elements", len(bad), "blocks", sorted(set(bad[:, 0].tolist()))[:20], "chans", sorted(set(bad[:, 1].tolist()))[:10])
    rx.close()
