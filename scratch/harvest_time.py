"""where the harvest leg's host time goes (MCRX_DEBUG=8 prints the phase timers at destroy)"""
import os, sys, time
os.environ["MCRX_DEBUG"] = "8"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from __graft_entry__ import load_product
P = load_product()
N, M, cp = 512, 64, 8
tx = P.multichanneltx(N, M, cp, 4)
slabs = [tx.generate(16, 1200, seed=1 + i, nblocks=int(P.lib().mctx_hip_blocks_for(tx._h, 16, 1200, 40, 1, 6)) + 24 * i)[0] for i in range(3)]
tx.close()
for skip, wait in ((1, True), (0, True)):
    rx = P.multichannelrx(N, M, cp, 4, max_payload_len=1200, max_frames=N * 16 + 64, skip_framesyms=skip)
    for d in slabs:
        rx.Execute(d); rx.Poll(deliver=False); rx.drain_count()
    rx.Flush(); rx.drain_count(); torch.cuda.synchronize()
    steps = 6
    t0 = time.perf_counter(); tp = td = te = 0.0
    for _ in range(steps):
        for d in slabs:
            a = time.perf_counter(); rx.Execute(d); b = time.perf_counter(); rx.Poll(deliver=False); c = time.perf_counter(); rx.drain_count(); e = time.perf_counter()
            te += b - a; tp += c - b; td += e - c
    P.lib().mcrx_hip_flush(rx._h); rx.drain_count()
    dt = time.perf_counter() - t0
    n = sum(int(d.numel()) for d in slabs) * steps
    print("wait=%s " % wait + "skip_framesyms=%d: %.1f Gsample/s; per slab: execute %.3f ms, poll %.3f ms, drain %.3f ms" % (skip, n / dt / 1e9, te / steps / 3 * 1e3, tp / steps / 3 * 1e3, td / steps / 3 * 1e3), flush=True)
    rx.close()
