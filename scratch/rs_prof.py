"""msresamp / firpfbch2 kernels alone, for rocprofv3 (scratch/prof_rs.sh): configs[2]'s resampler on a 77.9 M-sample stream.
usage: rs_prof.py [rs0.5|rs0.8|rs0.37|rs2.0|pfb1024|pfb128|pfb16 ...]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from __graft_entry__ import load_product
P = load_product()
n = 77_900_000 // 1024 * 1024
x = torch.randn(n, dtype=torch.complex64, device="cuda")
what = sys.argv[1:] or ["rs0.5", "rs0.8", "rs0.37", "rs2.0", "pfb1024", "pfb128", "pfb16"]
for w in what:
    if w.startswith("rs"):
        rate = float(w[2:])
        rs = P.msresamp(rate)
        nn = n if rate <= 1 else n // 4
        for rep in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); y = rs.execute(x[:nn]); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print("msresamp(%.2f): %.3f ms for %d samples in -> %.1f Gsample/s in, %d out" % (rate, ms, nn, nn / ms / 1e6, y.numel()), flush=True)
        rs.close(); del y
    else:
        M = int(w[3:]); ns = {1024: 100000, 128: 400000, 16: 2000000}[M]
        pfb = P.firpfbch2(M, 7)
        nn = ns * M // 2
        for rep in range(4):
            pfb.reset()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); pfb.analyze(x[:nn]); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print("firpfbch2 M=%d: %.3f ms for %d samples -> %.1f Gsample/s in, %.0f GB/s algorithmic (24 B/sample)" % (M, ms, nn, nn / ms / 1e6, nn * 24 / ms / 1e6), flush=True)
        pfb.close()
