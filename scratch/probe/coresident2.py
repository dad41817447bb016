"""SMT test: ONE long channelizer launch whose 254 workgroups stay resident (slab = whole stream / 254), short burner waves beside it.
usage: MCRX_LIB=... python scratch/probe/coresident2.py [nacc] [iters] [nwaves]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from __graft_entry__ import load_product
prod = load_product()
P = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libprobe.so"))
P.probe_burn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
nacc = int(sys.argv[1]) if len(sys.argv) > 1 else 72
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
nwaves = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
N, K = 512, 1024
slab = 4000
nblocks = 254 * slab
x = torch.view_as_complex(torch.randn(nblocks * K, 2, device="cuda"))
out = torch.empty(nblocks // 8 * N * 8, dtype=torch.complex64, device="cuda")
rx = prod.multichannelrx(N, 64, 8, 4, slab_blocks=slab)
buf = torch.zeros(nwaves * 4, dtype=torch.int64, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def ev(): return torch.cuda.Event(enable_timing=True)
def chan(): rx.channelize(x, nblocks, 0, out, stream=s1)
def burn(): P.probe_burn(C.c_void_p(s2.cuda_stream), nwaves, nacc, iters, C.c_void_p(buf.data_ptr()))
chan(); burn(); torch.cuda.synchronize()
a0, a1 = ev(), ev(); a0.record(s1); chan(); a1.record(s1); torch.cuda.synchronize(); t_chan = a0.elapsed_time(a1)
b0, b1 = ev(), ev(); b0.record(s2); burn(); b1.record(s2); torch.cuda.synchronize(); t_burn = b0.elapsed_time(b1)
alone = buf.cpu().numpy().reshape(-1, 4).copy()
a0, a1, b0, b1 = ev(), ev(), ev(), ev()
a0.record(s1); chan(); a1.record(s1)
time.sleep(0.0005)
b0.record(s2); burn(); b1.record(s2)
torch.cuda.synchronize()
tc, tb = a0.elapsed_time(a1), b0.elapsed_time(b1)
both = buf.cpu().numpy().reshape(-1, 4)
def stats(r, t_lo=None, t_hi=None):
    w0, w1, hw = r[:, 0].astype(np.int64), r[:, 1].astype(np.int64), r[:, 3].astype(np.uint64)
    cu = ((hw >> np.uint64(32)) & np.uint64(0xf)) * np.uint64(1 << 16) + (hw & np.uint64(0xff00))
    simd = (hw >> np.uint64(4)) & np.uint64(3)
    t0 = w0.min()
    # waves alive at the midpoint of the burner's span, per CU
    mid = (w0.min() + w1.max()) // 2
    alive = (w0 <= mid) & (w1 > mid)
    per_cu = np.unique(cu[alive], return_counts=True)[1]
    per_simd = np.unique(cu[alive] * np.uint64(4) + simd[alive], return_counts=True)[1]
    return dict(cus=len(np.unique(cu)), cus_mid=len(per_cu), per_cu_mid_mean=float(per_cu.mean()), per_cu_mid_max=int(per_cu.max()),
                per_simd_mid_max=int(per_simd.max()), span_us=float(w1.max() - w0.min()) / 100.0, wave_cycles=float(r[:, 2].mean()))
print("lib", os.path.basename(os.environ.get("MCRX_LIB", "default")), "| burner %d waves x %d iters x %d acc" % (nwaves, iters, nacc))
print("  alone   : channelizer (254 resident workgroups) %.3f ms; burner %.3f ms" % (t_chan, t_burn), stats(alone))
print("  together: channelizer %.3f ms (x%.2f); burner %.3f ms (x%.2f)" % (tc, tc / t_chan, tb, tb / t_burn), stats(both))
rx.close()
