"""one bench slab from the GPU transmitter (run under rocprofv3 --kernel-trace --stats for the kernel times)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_product
prod = load_product()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
tx = prod.multichanneltx(N, 64, 8, 4)
nb = int(prod.lib().mctx_hip_blocks_for(tx._h, 16, 1200, 40, 1, 6))
for i in range(3):
    d, s = tx.generate(16, 1200, seed=1 + i, nblocks=nb)
torch.cuda.synchronize()
print("generated", d.numel(), "samples x3")
