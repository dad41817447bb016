#!/usr/bin/env python3
"""The other BASELINE.json configurations on one MI355X as continuous, never restarted streams (round-2 form of
scratch/configs.py): two different slabs per step pushed through one receiver with Discard(), then one step harvested
and every frame checked against what was sent.
  C2  8-ch multichannelrx, M=64 QPSK h128, 100 frames/ch per slab
  C2v the same with the r = 1/2 K = 7 convolutional code as outer code
  C3  64-ch multichannelrx, M=256 QAM16 + Golay(24,12), msresamp(0.5) front end fed the 2x stream of a TX-side msresamp(2.0)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
from __graft_entry__ import load_product
prod = load_product()
dev = torch.device("cuda", 0)


def run(N, M, cp, frames, plen, mod, fec1, resamp, steps=int(os.environ.get('CFG_STEPS', 20)), warm=int(os.environ.get('CFG_WARM', 4))):
    K = 2 * N
    tx = prod.multichanneltx(N, M, cp, 4)
    base = int(prod.lib().mctx_hip_blocks_for(tx._h, frames, plen, mod, 1, fec1))
    slabs = []
    for i in range(2):
        iq, sent = tx.generate(frames, plen, mod=mod, fec1=fec1, seed=70 + i, nblocks=base + 64 * i, device=dev)
        idx = [{(h[0] << 8) | h[1]: (h, p) for (h, p) in ch} for ch in sent]
        x = iq
        if resamp:
            up = prod.msresamp(2.0); x = up.execute(iq).clone(); up.close()
        slabs.append((x, idx, int(iq.numel())))
    torch.cuda.synchronize(); tx.close()
    rx = prod.multichannelrx(N, M, cp, 4, max_payload_len=plen, max_frames=N * frames + 64)
    rs = prod.msresamp(0.5) if resamp else None
    side = torch.cuda.Stream(device=dev)
    USE_SIDE = os.environ.get('CFG_SIDE', '1') != '0'
    alive = []

    def push(x):
        y = x
        if rs is not None:
            # (on the caller's stream, like the receiver's push below: a resampler left on the NULL stream -- the legacy default
            #  stream, which every blocking stream of the process waits for and which waits for all of them -- is a barrier between
            #  consecutive pushes: nothing of push k + 1 starts before push k's decoder has finished.  43 -> see DESIGN 4.5)
            rs.reset(); y = rs.execute(x, stream=side if USE_SIDE else None)
            y = y[:int(y.numel()) // (16 * N) * (16 * N)]
            alive.append(y); del alive[:-4]         # (the receiver's channelizer may still read a push's samples when the next push is enqueued)
        rx.Execute(y, stream=side if (USE_SIDE and rs is not None) else None)

    def step(keep=False):
        for x, _, _ in slabs:
            push(x)
            rx.Poll() if keep else rx.Discard()
    # (side is created above push())  an explicit caller stream: resampler and receiver order on it, nothing
    side.wait_stream(torch.cuda.current_stream())   # serialises against the legacy default stream
    with torch.cuda.stream(side):
        for _ in range(warm): step()
        torch.cuda.synchronize(); rx.kernel_stats(reset=True); t0 = time.perf_counter()
        for _ in range(steps): step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
    per = {k: round(v[0] / max(v[1], 1), 4) for k, v in rx.kernel_stats().items()}
    walked, adopted = rx.spec_stats()
    with torch.cuda.stream(side):
        rx.Flush(); rx.frames.clear()
        step(keep=True); rx.Flush()
    ok = 0
    for f in rx.frames:
        if f.payload_valid and any(s[1][f.channel].get((f.header[0] << 8) | f.header[1]) == (f.header, f.payload) for s in slabs): ok += 1
    if ok != len(rx.frames) and os.environ.get("CFG_DEBUG"):
        badf = [f for f in rx.frames if not (f.payload_valid and any(s[1][f.channel].get((f.header[0] << 8) | f.header[1]) == (f.header, f.payload) for s in slabs))]
        print("bad frames", len(badf), [(f.channel, (f.header[0] << 8) | f.header[1], f.header_valid, f.payload_valid, f.end_sample, len(f.payload)) for f in badf[:16]], file=sys.stderr)
    n_in = sum(int(s[0].numel()) for s in slabs)
    res = {"channels": N, "M": M, "mod": mod, "fec1": fec1, "resamp": resamp, "wideband_samples_per_step": n_in,
           "ms_per_step": round(dt * 1e3, 4), "Msamples_per_s": round(n_in / dt / 1e6, 1), "kernels_ms_overlapped": per, "frames_acquired": {"by_scout_walk": walked, "adopted_from_speculation": adopted},
           "verified": {"frames": len(rx.frames), "expected": 2 * N * frames, "bit_exact": ok}}
    rx.close()
    if rs is not None: rs.close()
    return res


which = sys.argv[1:] or ["C2", "C2_conv_v27", "C3"]
cfgs = {"C2": (8, 64, 8, 100, 1200, 40, 6, False), "C2_conv_v27": (8, 64, 8, 100, 1200, 40, 11, False), "C3": (64, 256, 32, 32, 1200, 27, 7, True), "C3_noresamp": (64, 256, 32, 32, 1200, 27, 7, False), "C3_qpsk_h128": (64, 256, 32, 32, 1200, 40, 6, False), "M64_64ch": (64, 64, 8, 32, 1200, 40, 6, False)}
out = {k: run(*cfgs[k]) for k in which}
print(json.dumps(out))
