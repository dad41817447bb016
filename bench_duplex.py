#!/usr/bin/env python3
"""BASELINE configs[4]: 256-channel full-duplex multichanneltxrx (src/multichannel_txrx.cc) as a multi-GPU job.
A secondary measurement -- bench.py holds the headline metric.

    python bench_duplex.py [--gpus N --steps K --warmup W]         (N > 1 starts its own ranks, or runs under torch.distributed.run like bench.py)

Per round every rank: channel-rate granules of its channel shard for every rank's sub-slab -> all-to-all -> synthesis
bank + oscillator over its own sub-slab (sharding.TxPipeline) -> the same samples, still on the GPU that made them,
into the round-robin sharded receiver (sharding.Pipeline: channelizer -> all-to-all -> synchronizers of its channel
shard).  Frame bits are assembled once, untimed (the reference's traffic loop does it per packet on the host).
value = wideband samples transmitted AND received per second, whole job.  One JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--channels", type=int, default=256)
    ap.add_argument("--payload", type=int, default=1200)
    ap.add_argument("--sub-blocks", type=int, default=262144, help="blocks of 2N samples one rank synthesizes / channelizes per round")
    ap.add_argument("--dry-run-launch", action="store_true", help="start the ranks, rendezvous under gloo, print one JSON line and exit (no GPU needed)")
    args = ap.parse_args()
    import importlib.util
    spec = importlib.util.spec_from_file_location("mcrx_launch", os.path.join(ROOT, "liquid-usrp_amd", "launch.py"))
    launch = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(launch)
    if args.dry_run_launch:
        sys.exit(launch.dry_run(args.gpus))
    rank, world, local = launch.ensure_ranks(args.gpus)         # --gpus N > 1 without a launcher: the ranks are started here
    import torch
    from __graft_entry__ import load_product
    assert torch.cuda.is_available(), "needs a GPU: the HIP kernels are the only implementation"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=dev)
    prod = load_product()
    out, ok, msg = measure(prod, torch, dev, rank, world, dist, args.channels, args.payload, args.sub_blocks, args.steps, args.warmup)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        sys.exit(msg)


def measure(prod, torch, dev, rank, world, dist, channels=256, payload=1200, sub_blocks=262144, steps=20, warmup=5):
    """One full-duplex measurement on the ranks that exist (world = 1: everything on one GPU).  Returns (JSON-able dict, ok, message)."""
    import argparse
    args = argparse.Namespace(channels=channels, payload=payload, sub_blocks=sub_blocks, steps=steps, warmup=warmup)
    from liquid_usrp_amd import sharding
    N, M, cp, taper, Tc = args.channels, 64, 8, 4, args.sub_blocks
    K = 2 * N
    c0, cg = sharding.shard_of(rank, world, N)
    rounds = args.warmup + args.steps + 1
    tx = prod.multichanneltx(N, M, cp, taper)
    per_frame = int(prod.lib().mctx_hip_blocks_for(tx._h, 1, args.payload, 40, 1, 6)) - 64
    total_blocks = rounds * world * Tc
    frames = max(1, (total_blocks - 4096) // per_frame)                  # the stream is full of frames up to its last blocks
    t0 = time.time()
    tr = tx.traffic(c0, cg, frames, args.payload, seed=0xD0_0D)
    setup = time.time() - t0
    rx = prod.multichannelrx(N, M, cp, taper, max_payload_len=args.payload, channel_first=c0, channel_count=cg,
                             max_frames=cg * (world * Tc // per_frame + 2) + 64, defer_samples=16384)
    nbuf = int(os.environ.get("MCRX_PIPE_NBUF", "5"))              # rotating buffer sets of both pipelines (the receiver has five slots: 62.0 -> 64.8 Gsample/s against three)
    txp = sharding.TxPipeline(tx, tr, rank, world, dist, N, Tc, device=dev, nbuf=nbuf)
    rxp = sharding.Pipeline(rx, rank, world, dist, N, Tc, rx.hist_tiles, device=dev, nbuf=nbuf)
    keep = txp.keep
    state = {"c": 0}

    def step(keep_frames=False):
        c = state["c"]
        consumed = rxp.evA[(c - txp.nbuf) % rxp.nbuf] if c >= txp.nbuf else None
        iq, ev = txp.push(consumed=consumed)
        u = c * world + rank
        rxp.push(iq[keep * K:], halo=iq[(keep - 13) * K:keep * K] if u > 0 else None, after=ev)
        if keep_frames:
            rx.Poll()
        else:
            rx.Discard()
        state["c"] = c + 1

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # verification: one more round with the frames delivered; everything it completes is what this rank's shard sent
    rx.Flush(); rx.frames.clear()
    step(keep_frames=True)
    torch.cuda.synchronize()
    rx.Flush()
    n_ok = sum(1 for f in rx.frames if f.header_valid and f.payload_valid and
               tr.sent[f.channel - c0][(f.header[0] << 8) | f.header[1]] == (f.header, f.payload))
    nfr = len(rx.frames)
    lo = cg * (world * Tc // per_frame - 1)
    ok = nfr >= lo and n_ok == nfr
    value = world * Tc * K * args.steps / elapsed / 1e6
    out = None
    if True:
        out = ({
            "metric": "complex Msamples/s transmitted and received by a full-duplex multichanneltxrx", "value": round(value, 3),
            "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%d-ch full-duplex multichanneltxrx: N x ofdmflexframegen + synthesis bank (m=13) + oscillator "
                                   "concurrent with the %d-ch multichannelrx, M=64 cp=8 taper=4 QPSK CRC32+Hamming128 %dB payloads, "
                                   "frames back to back on every channel" % (N, N, args.payload),
                       "channels": N, "samples_per_step": world * Tc * K,
                       "parallelism": ("TX: %d channels/GPU -> all-to-all -> time-sharded synthesis; RX: time-sharded channelizer -> "
                                       "all-to-all -> %d channels/GPU" % (cg, cg)) if world > 1 else "single GPU"},
            "verified": {"frames": nfr, "at_least": lo, "bit_exact_payloads": n_ok, "ok": ok, "note": "the round after the timed region"},
            "setup_s": {"frame_assembly_and_modulation": round(setup, 2), "frames_per_channel": frames}})
    rx.close(); tr.close(); tx.close()
    return out, ok, "rank %d: verification failed (%d frames, at least %d expected, %d ok)" % (rank, nfr, lo, n_ok)


if __name__ == "__main__":
    main()
