"""Does another kernel's wave share a CU with a resident K = 1024 channelizer workgroup?  (MCRX_LIB selects the build.)
usage: python scratch/probe/coresident.py [nacc=72] [iters] [nwaves]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from __graft_entry__ import load_product
prod = load_product()
P = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libprobe.so"))
P.probe_burn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
nacc = int(sys.argv[1]) if len(sys.argv) > 1 else 72
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
nwaves = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
N, K = 512, 1024
nblocks = 101376 * 2
x = torch.view_as_complex(torch.randn(nblocks * K, 2, device="cuda"))
out = torch.empty(nblocks // 8 * N * 8, dtype=torch.complex64, device="cuda")
rx = prod.multichannelrx(N, 64, 8, 4)
buf = torch.zeros(nwaves * 4, dtype=torch.int64, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def ev(): return torch.cuda.Event(enable_timing=True)
def chan(n):
    for _ in range(n):
        rx.channelize(x, nblocks, 0, out, stream=s1)
def burn():
    P.probe_burn(C.c_void_p(s2.cuda_stream), nwaves, nacc, iters, C.c_void_p(buf.data_ptr()))
# warm
chan(3); burn(); torch.cuda.synchronize()
# alone
a0, a1 = ev(), ev(); a0.record(s1); chan(4); a1.record(s1); torch.cuda.synchronize(); t_chan = a0.elapsed_time(a1) / 4
b0, b1 = ev(), ev(); b0.record(s2); burn(); b1.record(s2); torch.cuda.synchronize(); t_burn = b0.elapsed_time(b1)
alone = buf.cpu().numpy().reshape(-1, 4).copy()
# together: channelizer first (4 launches), burner beside it
a0, a1, b0, b1 = ev(), ev(), ev(), ev()
t0 = time.perf_counter()
a0.record(s1); chan(4); a1.record(s1)
b0.record(s2); burn(); b1.record(s2)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) * 1e3
tc, tb = a0.elapsed_time(a1), b0.elapsed_time(b1)
both = buf.cpu().numpy().reshape(-1, 4)
def conc(r):
    """mean number of burner waves alive at once per CU, over the burner's lifetime"""
    w0, w1, hw = r[:, 0].astype(np.int64), r[:, 1].astype(np.int64), r[:, 3].astype(np.uint64)
    cu = ((hw >> np.uint64(32)) & np.uint64(0xf)) * np.uint64(1 << 16) + (hw & np.uint64(0xff00)).astype(np.uint64)       # xcc | se/sh/cu bits
    ncu = len(np.unique(cu))
    life = float(np.sum(w1 - w0)); span = float(w1.max() - w0.min())
    return ncu, life / span / ncu, span / 100.0       # wall clock is 100 MHz -> us
print("lib", os.environ.get("MCRX_LIB", "default"), "burner: %d waves x %d iters x %d accumulators" % (nwaves, iters, nacc))
print("alone   : channelizer %.3f ms per launch; burner %.3f ms (mean wave %.0f cycles)" % (t_chan, t_burn, alone[:, 2].mean()))
n1, c1, sp1 = conc(alone)
print("          burner alone: %d CUs seen, %.1f waves alive per CU on average, span %.0f us" % (n1, c1, sp1))
print("together: channelizer x4 %.3f ms (%.3f per launch), burner %.3f ms, wall %.3f ms; sum of alone = %.3f ms" % (tc, tc / 4, tb, wall, 4 * t_chan + t_burn))
n2, c2, sp2 = conc(both)
print("          burner beside it: %d CUs seen, %.1f waves alive per CU on average, mean wave %.0f cycles, span %.0f us" % (n2, c2, both[:, 2].mean(), sp2))
rx.close()
