#!/usr/bin/env python3
"""One rank of the 8-GPU sharding emulated on one GPU as a CONTINUOUS stream (scratch/mg8.py restarts every step): 64 of 512 channels
over rounds of 8 time slabs -- what rank 0 of `bench.py --gpus 8` synchronizes after every all-to-all, history tiles in front like
sharding.Pipeline.  The same slab serves as every source rank's contribution, so the cadence holds from round to round.
Times the synchronizer stages (acquisition, placement, payload workers, decoder) per round; frames per channel and round = 8 x argv[1]."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from __graft_entry__ import load_product
prod = load_product()
dev = torch.device("cuda", 0)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N, M, cp, world = 512, 64, 8, int(os.environ.get("MG_WORLD", "8"))
K, cg, TILE = 2 * N, N // world, prod.TILE
tx = prod.multichanneltx(N, M, cp, 4)
iq, sent = tx.generate(frames, 1200, seed=0xC0FFEE, device=dev)
torch.cuda.synchronize(); tx.close()
T = iq.numel() // K // TILE * TILE
rx = prod.multichannelrx(N, M, cp, 4, max_payload_len=1200, channel_first=0, channel_count=cg, max_frames=cg * frames * world + 64, defer_samples=16384)
out = torch.empty(world * T * cg, dtype=torch.complex64, device=dev)
rx.channelize(iq[:T * K], T, 0, out, groups=world)
per = T * cg
hist = rx.hist_tiles
he = hist * cg * TILE
bufs = []
for i in range(3):
    b = torch.zeros(he + world * per, dtype=torch.complex64, device=dev)
    b[he:] = out[:per].repeat(world)
    b[:he] = b[-he:]                                   # the previous round's tail (every round is the same 8 slabs)
    bufs.append(b)
torch.cuda.synchronize()
def rnd(c):
    rx.sync(bufs[c % 3], c * world * T - hist * TILE, hist * TILE + world * T)
    rx.Discard()
c = 0
for _ in range(int(os.environ.get('MG8_WARM', '6'))): rnd(c); c += 1
torch.cuda.synchronize(); rx.kernel_stats(reset=True); rx.spec_stats(reset=True)
n = 24
t0 = time.perf_counter()
for _ in range(n): rnd(c); c += 1
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
perk = {k: round(v[0] / max(v[1], 1), 4) for k, v in rx.kernel_stats().items()}
walked, adopted = rx.spec_stats()
print(json.dumps({"frames_per_channel_and_round": frames * world, "sync_ms_per_round": round(dt * 1e3, 4), "kernels_ms": perk,
                  "walked": walked, "adopted": adopted, "expected_per_round": cg * frames * world}))
