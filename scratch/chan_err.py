"""Channelizer output error against the oracle (relative to the largest output), per channel count."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
from conftest import load_product
import oracle as O
O.build()
P = load_product()
for N in (2, 8, 64, 512):
    K = 2 * N
    nblocks = 64 if N >= 64 else 256
    rng = np.random.RandomState(N)
    x = (rng.randn(nblocks * K) + 1j * rng.randn(nblocks * K)).astype(np.complex64)
    ref = O.MultiChannelRx(N, 64, 8, 4).channelize(x)
    rx = P.multichannelrx(N, 64, 8, 4)
    d_out = torch.zeros(nblocks // 8 * N * 8, dtype=torch.complex64, device="cuda")
    rx.channelize(torch.from_numpy(x).cuda(), nblocks, 0, d_out)
    torch.cuda.synchronize()
    got = P.tiles_to_channels(d_out, N).T
    print(N, "max rel err %.3e  rms rel err %.3e" % (np.max(np.abs(got - ref)) / np.max(np.abs(ref)), np.sqrt(np.mean(np.abs(got - ref) ** 2)) / np.sqrt(np.mean(np.abs(ref) ** 2))))
    rx.close()
