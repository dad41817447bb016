"""Controlled form of VERDICT r5 "next" #4a: an f32-VALU-only kernel (v_fma_f32, one wave per SIMD) and an f32-MFMA-only kernel (one wave per SIMD)
alone and side by side on two streams.  Prints TFLOP/s of each alone and together."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
G = ctypes.CDLL(os.path.join(ROOT, "scratch", "r6", "ubench", "libguest.so"))
G.guest_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]; G.guest_flops.restype = ctypes.c_double; G.guest_flops.argtypes = [ctypes.c_int, ctypes.c_int]
s = [torch.cuda.Stream(), torch.cuda.Stream()]
IT = {0: 40000, 1: 20000}
def run(kinds, reps=4):
    torch.cuda.synchronize()
    ev = {k: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for k in kinds}
    for k in kinds: ev[k][0].record(s[k])
    for _ in range(reps):
        for k in kinds: G.guest_launch(k, IT[k], ctypes.c_void_p(s[k].cuda_stream))
    for k in kinds: ev[k][1].record(s[k])
    torch.cuda.synchronize()
    return {("mfma" if k == 0 else "valu"): round(G.guest_flops(k, IT[k]) * reps / (ev[k][0].elapsed_time(ev[k][1]) * 1e-3) / 1e12, 1) for k in kinds}
run([0]); run([1])
print(json.dumps({"mfma_alone": run([0]), "valu_alone": run([1]), "side_by_side": run([0, 1]), "note": "one wave per SIMD each; v_mfma_f32_32x32x2_f32 x 4 accumulators; v_fma_f32 x 16 accumulators"}))
