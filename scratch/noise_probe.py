"""the headline stream with white noise on the wideband samples: throughput, decoder time, frames that still check (how much of the
Hamming soft decision's early-out survives errors in the hard decisions)"""
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
clean = [tx.generate(frames, payload, seed=0xC0FFEE + 7919 * i, nblocks=base + (0, 80, 32)[i], device=dev)[0] for i in range(3)]
torch.cuda.synchronize(); tx.close()
samples = sum(int(d.numel()) for d in clean)
sig = float(clean[0][: 1 << 22].abs().pow(2).mean().sqrt())
for snr in (None, 20.0, 10.0, 6.0, 3.0):
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    slabs = clean if snr is None else [d + (sig * 10 ** (-snr / 20) / 2 ** 0.5) * torch.view_as_complex(torch.randn(d.numel(), 2, generator=g, device="cuda")) for d in clean]
    rx = prod.multichannelrx(N, M, cp, taper, max_payload_len=payload, max_frames=N * frames + 64, skip_framesyms=1)
    def step(poll=False):
        for d in slabs:
            rx.Execute(d)
            rx.Poll(deliver=False) if poll else rx.Discard()
    for _ in range(4): step()
    torch.cuda.synchronize(); rx.kernel_stats(reset=True)
    t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ks = rx.kernel_stats()
    prod.lib().mcrx_hip_flush(rx._h); rx.drain_count()
    nf = nok = 0
    for d in slabs:
        rx.Execute(d); rx.Poll(deliver=False)
        a, b, c = rx.drain_count(); nf += a; nok += b
    prod.lib().mcrx_hip_flush(rx._h)
    a, b, c = rx.drain_count(); nf += a; nok += b
    print(json.dumps({"wideband_snr_dB": snr, "Gsample_per_s": round(samples * 10 / dt / 1e9, 1), "decode_ms_overlapped": round(ks["decode_kernel"][0] / max(ks["decode_kernel"][1], 1), 4),
                      "frames_of_a_step": nf, "valid": nok, "sent": 3 * N * frames}), flush=True)
    rx.close()
    del slabs
