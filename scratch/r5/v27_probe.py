"""configs[1] with the K = 7 code: which kernels carry the decode (rocprofv3 --kernel-trace --stats around this script)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
prod = load_product()
N, M, cp, frames, plen = 8, 64, 8, int(os.environ.get("V27_FRAMES", "100")), 1200
tx = prod.multichanneltx(N, M, cp, 4)
d, sent = tx.generate(frames, plen, mod=40, fec1=11, seed=3)
tx.close()
rx = prod.multichannelrx(N, M, cp, 4, max_payload_len=plen, max_frames=N * frames + 64, serial=int(os.environ.get('V27_SERIAL', '0')))
for _ in range(4):
    rx.Execute(d); rx.Discard()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    rx.Execute(d); rx.Discard()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("Gsample/s", d.numel() * 10 / dt / 1e9, "ms/push", dt / 10 * 1e3, rx.kernel_stats())
rx.Flush(); rx.viterbi_stats(reset=True)
rx.Execute(d); rx.Flush()
print("frames", len(rx.frames), sum(f.payload_valid for f in rx.frames), "viterbi (frames, fwd repeats, tb repeats)", rx.viterbi_stats())
