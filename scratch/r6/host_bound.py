"""Is the few-channel receiver host-bound?  Host time of Execute() + Discard() per push against the device period (configs[1] shape).
   python scratch/r6/host_bound.py [channels frames fec1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench, torch
from __graft_entry__ import load_product
prod = load_product()
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 100
fec1 = int(sys.argv[3]) if len(sys.argv) > 3 else 6
M, cp, taper, plen, mod = int(os.environ.get("HB_M", 64)), int(os.environ.get("HB_CP", 8)), 4, 1200, int(os.environ.get("HB_MOD", 40))
tx = prod.multichanneltx(N, M, cp, taper)
base = int(prod.lib().mctx_hip_blocks_for(tx._h, frames, plen, mod, 1, fec1))
slabs = [tx.generate(frames, plen, mod=mod, fec1=fec1, seed=0xBEEF + 7919 * i, nblocks=base + (0, 48)[i], device=dev)[0] for i in range(2)]
torch.cuda.synchronize(); tx.close()
rx = prod.multichannelrx(N, M, cp, taper, max_payload_len=plen, max_frames=N * frames + 64, **bench.LEG_CFG)
for _ in range(6):
    for x in slabs: rx.Execute(x); rx.Discard()
torch.cuda.synchronize()
for pushes in (12, 60):
    t0 = time.perf_counter()
    for i in range(pushes):
        rx.Execute(slabs[i & 1]); rx.Discard()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%d ch x %d frames, fec1 %d: %d pushes: host enqueue %.1f us per push, with the device %.1f us per push (%.1f Gsample/s)" %
          (N, frames, fec1, pushes, (t1 - t0) / pushes * 1e6, (t2 - t0) / pushes * 1e6, sum(int(s.numel()) for s in slabs) / 2 / ((t2 - t0) / pushes) / 1e9), flush=True)
if os.environ.get("MCRX_EVT_DUMP"):
    rx.kernel_stats()          # (development build: folds the event pairs, which prints them)
t0 = time.perf_counter()
for i in range(60):
    rx.Execute(slabs[i & 1])
t1 = time.perf_counter()
torch.cuda.synchronize()
print("Execute alone: %.1f us per push on the host" % ((t1 - t0) / 60 * 1e6))
rx.close()
