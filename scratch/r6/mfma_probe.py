"""VERDICT r5 "next" #4a: is the matrix pipe really free beside the receiver?  The pipelined 512-channel receiver (the headline's loop) alone;
then with a guest kernel that issues only f32 MFMA (one wave per SIMD, resident the whole time) on a stream of its own; then with a
guest that issues only f32 VALU fma.  Prints one JSON line: receiver Gsample/s in each case, guest TFLOP/s alone and beside."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from __graft_entry__ import load_product
prod = load_product()
G = ctypes.CDLL(os.path.join(ROOT, "scratch", "r6", "ubench", "libguest.so"))
G.guest_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]; G.guest_flops.restype = ctypes.c_double; G.guest_flops.argtypes = [ctypes.c_int, ctypes.c_int]
dev = torch.device("cuda:0")
N, M, cp, frames, plen = 512, 64, 8, 16, 1200
K = 2 * N
tx = prod.multichanneltx(N, M, cp, 4)
base = int(prod.lib().mctx_hip_blocks_for(tx._h, frames, plen, 40, 1, 6))
slabs = [tx.generate(frames, plen, mod=40, fec1=6, seed=0xBEEF + 7919 * i, nblocks=base + (0, 48, 96)[i], device=dev)[0] for i in range(3)]
tx.close(); torch.cuda.synchronize()
rx = prod.multichannelrx(N, M, cp, 4, max_payload_len=plen, max_frames=N * frames + 64)
nsamp = sum(int(s.numel()) for s in slabs)
gs = torch.cuda.Stream(device=dev)

def step():
    for s in slabs:
        rx.Execute(s); rx.Discard()

def guest_alone(kind, iters, reps=5):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(gs):
        e0.record(gs)
        for _ in range(reps): G.guest_launch(kind, iters, ctypes.c_void_p(gs.cuda_stream))
        e1.record(gs)
    torch.cuda.synchronize()
    return G.guest_flops(kind, iters) * reps / (e0.elapsed_time(e1) * 1e-3) / 1e12

def run(kind, iters, steps=30):
    """receiver for `steps` steps; kind >= 0: the guest relaunched back to back on its own stream for the whole time"""
    for _ in range(5): step()
    torch.cuda.synchronize()
    nl = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    if kind >= 0: e0.record(gs)
    for k in range(steps):
        step()
        if kind >= 0:
            G.guest_launch(kind, iters, ctypes.c_void_p(gs.cuda_stream)); nl += 1
    torch.cuda.current_stream().synchronize(); rx.Flush()
    dt = time.perf_counter() - t0
    if kind >= 0: e1.record(gs)
    torch.cuda.synchronize()
    gt = G.guest_flops(kind, iters) * nl / (e0.elapsed_time(e1) * 1e-3) / 1e12 if kind >= 0 else None
    return nsamp * steps / dt / 1e9, gt

out = {}
# guest launch length: about one receiver step (3.4 ms) so that it is resident throughout
it_m = 60000; it_v = 30000
out["guest_mfma_alone_TFLOPs"] = round(guest_alone(0, it_m), 1)
out["guest_valu_alone_TFLOPs"] = round(guest_alone(1, it_v), 1)
res = {"alone": [], "beside_mfma_guest": [], "beside_valu_guest": []}
gg = {"beside_mfma_guest": [], "beside_valu_guest": []}
for rep in range(3):
    res["alone"].append(run(-1, 0)[0])
    v, g = run(0, it_m); res["beside_mfma_guest"].append(v); gg["beside_mfma_guest"].append(g)
    v, g = run(1, it_v); res["beside_valu_guest"].append(v); gg["beside_valu_guest"].append(g)
for k, v in res.items(): out["receiver_Gsamples_" + k] = [round(x, 1) for x in v]
for k, v in gg.items(): out["guest_TFLOPs_" + k] = [round(x, 1) for x in v]
a = float(np.median(res["alone"]))
out["receiver_slowdown_beside_mfma_guest"] = round(1 - float(np.median(res["beside_mfma_guest"])) / a, 4)
out["receiver_slowdown_beside_valu_guest"] = round(1 - float(np.median(res["beside_valu_guest"])) / a, 4)
out["f32_peak_TFLOPs"] = 157.3
print(json.dumps(out))
