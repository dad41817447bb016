"""which frames of the N = 256 ragged stream differ between the GPU receiver and the oracle beyond 1e-5 (framesyms)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from __graft_entry__ import load_product, load_oracle
prod = load_product(); oracle = load_oracle()
N, M, cp = 256, 64, 8
L = M + cp
nb = L * 420 // 8 * 8
tx = prod.multichanneltx(N, M, cp, 4)
iq, sent, starts = tx.generate_ragged(nb, len_lo=0, len_hi=300, gap_max=3, long_every=4, long_max=40, mod=40, fec1=6, gain=1.0 / N, seed=99)
tx.close(); torch.cuda.synchronize()
got = iq.cpu().numpy()
x = got[:len(got) // (16 * N) * (16 * N)]
ora = oracle.MultiChannelRx(N, M, cp, 4); ora.execute(x)
rx = prod.multichannelrx(N, M, cp, 4, max_payload_len=300)
step = 16 * N * 61
for i in range(0, len(x), step): rx.Execute(x[i:i + step])
rx.Flush()
from test_gpu_parity import match_frames, relerr
bad = 0
for fg, fo in match_frames(rx.frames, ora.frames):
    if len(fo.framesyms):
        e = relerr(fg.framesyms, fo.framesyms)
        if e > 1e-5:
            bad += 1
            d = np.abs(np.asarray(fg.framesyms) - np.asarray(fo.framesyms))
            print("ch %3d hv %d pv %d len %4d evm %.2f/%.2f rssi %.3f end %d err %.3g at sym %d of %d, |ref| max %.3g, n>1e-5: %d" % (
                fo.channel, fo.header_valid, fo.payload_valid, len(fo.payload), fg.evm, fo.evm, fo.rssi, getattr(fo, 'end_sample', -1), e,
                int(np.argmax(d)), len(d), float(np.max(np.abs(fo.framesyms))), int(np.sum(d > 1e-5 * np.max(np.abs(fo.framesyms))))))
print("frames", len(ora.frames), "bad", bad)
