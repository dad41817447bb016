"""Throughput of the (untuned) 2x-oversampled analysis bank kernel."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
from conftest import load_product
P = load_product()
for M, ns in ((1024, 100000), (128, 400000), (16, 2000000)):
    pfb = P.firpfbch2(M, 7)
    x = torch.randn(ns * M // 2, dtype=torch.complex64, device="cuda")
    pfb.analyze(x); torch.cuda.synchronize()
    pfb.reset()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); pfb.analyze(x); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    n = ns * M // 2
    print("M=%d: %.3f ms for %d samples -> %.1f Gsample/s in, %.0f GB/s algorithmic (24 B/sample)" % (M, ms, n, n / ms / 1e6, n * 24 / ms / 1e6))
    pfb.close()
