"""the headline stream through handles that differ in one thing at a time: skip_framesyms, Discard / Poll"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from __graft_entry__ import load_product
prod = load_product()
dev = torch.device("cuda", 0)
N, M, cp, taper, frames, payload = 512, 64, 8, 4, 16, 1200
tx = prod.multichanneltx(N, M, cp, taper)
base = int(prod.lib().mctx_hip_blocks_for(tx._h, frames, payload, 40, 1, 6))
slabs = [tx.generate(frames, payload, seed=0xC0FFEE + 7919 * i, nblocks=base + (0, 80, 32)[i], device=dev)[0] for i in range(3)]
torch.cuda.synchronize(); tx.close()
samples = sum(int(d.numel()) for d in slabs)
for skip in (0, 1):
    for mode in ("discard", "poll", "poll_nodrain"):
        rx = prod.multichannelrx(N, M, cp, taper, max_payload_len=payload, max_frames=N * frames + 64, skip_framesyms=skip)
        def step():
            for d in slabs:
                rx.Execute(d)
                if mode == "discard":
                    rx.Discard()
                else:
                    rx.Poll(deliver=False)
                    if mode == "poll":
                        rx.drain_count()
        for _ in range(4): step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(12): step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("skip_framesyms=%d %-12s %8.1f Gsample/s" % (skip, mode, samples * 12 / dt / 1e9), flush=True)
        rx.Flush(); rx.close()
