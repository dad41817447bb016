#!/usr/bin/env python3
"""One rank of the 8-GPU sharding emulated on one GPU: 64 of 512 channels over 8 time slabs (what rank 0 of
bench.py --gpus 8 synchronizes after the all-to-all).  Times the synchronizer stages only."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
prod = load_product()
dev = torch.device("cuda", 0)
N, M, cp, world, reps = 512, 64, 8, 8, 8
K, cg = 2 * N, N // world
tx = prod.multichanneltx(N, M, cp, 4)
iq, sent = tx.generate(reps, 1200, seed=0xC0FFEE, device=dev)
torch.cuda.synchronize(); tx.close()
T = iq.numel() // K
rx = prod.multichannelrx(N, M, cp, 4, max_payload_len=1200, channel_first=0, channel_count=cg, max_frames=cg * reps * world + 64)
out = torch.empty(world * (T // 8) * cg * 8, dtype=torch.complex64, device=dev)
rx.restart(); rx.channelize(iq, T, 0, out, groups=world)
chunk = (T // 8) * cg * 8
chan = out[:chunk].repeat(world).contiguous()           # the same slab 8 times = what every source rank would send
def step():
    rx.restart(); rx.sync(chan, 0, world * T)
for _ in range(3): step()
torch.cuda.synchronize(); rx.kernel_stats(reset=True)
t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
per = {k: round(v[0] / max(v[1], 1), 4) for k, v in rx.kernel_stats().items()}
rx.Flush(); fr = rx.frames
ok = sum(1 for f in fr if f.payload_valid and sent[f.channel][(f.header[0] << 8) | f.header[1]] == (f.header, f.payload))
print(json.dumps({"spec": os.environ.get("MCRX_NO_SPEC") is None, "sync_ms_per_step": round(dt * 1e3, 4), "kernels_ms": per, "frames": len(fr), "ok": ok, "expected": cg * reps * world}))
