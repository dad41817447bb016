#!/usr/bin/env python3
"""The other BASELINE.json configurations on one MI355X, beside bench.py's 512-channel line:
  C2  8-ch multichannelrx, M=64 QPSK h128, 100 frames/ch (20 M wideband samples)
  C3  64-ch multichannelrx, M=256 QAM16 + Golay(24,12), msresamp(0.5) front end fed a 2x oversampled stream
  C4  (= bench.py) 512-ch
IQ comes from the GPU transmitter (untimed); C3's 2x stream is made by zero stuffing + a half-band low-pass
(torch conv1d, untimed).  Every frame is checked against what was sent."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from __graft_entry__ import load_product
prod = load_product()
dev = torch.device("cuda", 0)

def timed(fn, steps=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps

def check(rx, sent, nexp):
    rx.Flush(); fr = rx.frames
    ok = sum(1 for f in fr if f.payload_valid and sent[f.channel][(f.header[0] << 8) | f.header[1]] == (f.header, f.payload))
    return {"frames": len(fr), "expected": nexp, "bit_exact": ok}

def run(N, M, cp, frames, plen, mod, fec1, resamp):
    K = 2 * N
    tx = prod.multichanneltx(N, M, cp, 4)
    iq, sent = tx.generate(frames, plen, mod=mod, fec1=fec1, seed=7, device=dev)
    torch.cuda.synchronize(); tx.close()
    rx = prod.multichannelrx(N, M, cp, 4, max_payload_len=plen, max_frames=N * frames + 64)
    rs = None
    x_in = iq
    if resamp:
        from scipy.signal import firwin
        h = torch.tensor(2.0 * firwin(63, 0.5), dtype=torch.float32, device=dev)
        up = torch.zeros(2 * iq.numel(), dtype=torch.complex64, device=dev); up[::2] = iq
        re = torch.nn.functional.conv1d(torch.view_as_real(up).T.reshape(2, 1, -1), h.view(1, 1, -1), padding=31)
        x_in = torch.view_as_complex(re.reshape(2, -1).T.contiguous())
        rs = prod.msresamp(0.5, 60.0)
    T = iq.numel() // K // 8 * 8
    d_out = torch.empty((T // 8) * N * 8, dtype=torch.complex64, device=dev)
    def step():
        rx.restart()
        y = iq
        if rs is not None:
            rs.reset(); y = rs.execute(x_in)
            nb = min(T, int(y.numel()) // K // 8 * 8)
        else:
            nb = T
        rx.channelize(y, nb, 0, d_out)
        rx.sync(d_out, 0, nb)
    dt = timed(step)
    rx.kernel_stats(reset=True)
    step(); torch.cuda.synchronize()
    per = {k: round(v[0] / max(v[1], 1), 4) for k, v in rx.kernel_stats().items()}
    res = check(rx, sent, N * frames)
    n_in = int(x_in.numel())
    rx.close()
    if rs is not None: rs.close()
    return {"channels": N, "M": M, "mod": mod, "fec1": fec1, "resamp": resamp, "wideband_samples_per_step": n_in,
            "ms_per_step": round(dt * 1e3, 4), "Msamples_per_s": round(n_in / dt / 1e6, 1), "kernels_ms": per, "verified": res}

out = {"C2": run(8, 64, 8, 100, 1200, 40, 6, False),
       "C3": run(64, 256, 32, 32, 1200, 27, 7, True)}
print(json.dumps(out))
