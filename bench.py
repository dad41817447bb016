#!/usr/bin/env python3
"""bench.py -- complex Msamples/s through a 512-channel multichannelrx on synthetic IQ.

One step = one pass of the hot path (NCO + polyphase analysis channelizer -> per-channel
OFDM frame synchronizer incl. header/payload decode) over one batch of wideband cf32 samples
that is already resident in HBM.  Workload = BASELINE.json config "512-ch multichannelrx":
N=512 channels (K=1024), M=64 subcarriers, cp=8, taper=4, QPSK, CRC-32 + Hamming(12,8),
1200-byte payloads, frames back to back on every channel (reference traffic recipe,
src/multichannel_tx.cc:163-213).

--gpus N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling.  Rank r
channelizes time slab r of the wideband stream, one RCCL all-to-all turns the time-sharded
channelizer output into channel shards (64 channels per GPU at N=8), rank r synchronizes
channels [r*512/N, (r+1)*512/N) over all slabs.  value = samples all ranks accepted / time.

The input is synthesised on the GPU by the product's own multichanneltx (txgen.hip; untimed,
parity-tested against the oracle's transmitter in tests/test_gpu_tx.py).  The CPU oracle
(oracle/) appears only as the `cpu_baseline` leg, timed on the same IQ on one host thread.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
B_CHANNELIZER = 12.0           # algorithmic bytes / wideband sample: 8 read + 4 written (N of 2N bins)
B_SYNC = 4.0                   # algorithmic bytes / wideband sample: the kept bins read once


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # a step is under a millisecond: 200 timed steps after 20 warm-up steps let the clocks settle (10 steps read ~10 % low)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--channels", type=int, default=512)
    ap.add_argument("--frames", type=int, default=8, help="frames per channel per GPU slab")
    ap.add_argument("--payload", type=int, default=1200)
    ap.add_argument("--cpu-reps", type=int, default=2, help="passes over the GPU slab timed on the CPU oracle")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--slab-blocks", type=int, default=0)
    ap.add_argument("--inflight", type=int, default=1,
                    help="steps in flight (throughput mode): consecutive steps are independent (each restarts its "
                         "receiver), so with D > 1 they run on D receiver handles / HIP streams round robin and the serial "
                         "per-channel chains of one step hide under the other steps' kernels.  The default 1 keeps every "
                         "kernel's HIP-event duration free of queueing behind other streams, which the roofline block needs")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    from __graft_entry__ import load_product, load_oracle
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    assert torch.cuda.is_available(), "bench.py needs a GPU: the HIP kernels are the only implementation"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=dev)      # RCCL over xGMI

    prod, ora = load_product(), load_oracle()
    N, M, cp, taper = args.channels, 64, 8, 4
    K = 2 * N
    assert N % world == 0
    cg = N // world

    # ---- synthetic IQ source (untimed): the GPU multichanneltx writes this rank's time slab
    # straight into HBM -- `frames` frames back to back on every channel, reference traffic recipe.
    # Every rank's slab carries the same frames (same seed); the slab ends in >= 64 idle blocks, so
    # the next slab's cold-start transient falls between frames.
    t0 = time.time()
    reps = args.frames
    tx = prod.multichanneltx(N, M, cp, taper)
    d_iq, sent = tx.generate(reps, args.payload, seed=0xC0FFEE, device=dev)
    torch.cuda.synchronize()
    tx.close()
    gen_s = time.time() - t0
    T = int(d_iq.numel()) // K                           # blocks per rank slab
    d_halo = d_iq[(T - 13) * K:].clone() if rank > 0 else None
    first_sample = rank * T * K
    ntiles = T // 8

    cfg = dict(max_payload_len=max(args.payload, 64), channel_first=rank * cg, channel_count=cg,
               max_frames=cg * reps * world + 64)
    if args.slab_blocks:
        cfg["slab_blocks"] = args.slab_blocks
    # D independent receivers, output buffers and streams: step k runs on slot k % D.  One step is still
    # restart -> channelize slab -> all-to-all (time shards -> channel shards) -> synchronize.
    D = max(1, min(args.inflight, args.steps))
    rxs = [prod.multichannelrx(N, M, cp, taper, **cfg) for _ in range(D)]
    d_outs = [torch.empty(world * ntiles * cg * 8, dtype=torch.complex64, device=dev) for _ in range(D)]    # [dest][tile][c][8]
    d_chans = [torch.empty_like(o) if world > 1 else o for o in d_outs]                                   # [src][tile][c][8]
    streams = [torch.cuda.Stream(device=dev) for _ in range(D)] if D > 1 else [torch.cuda.current_stream()]

    from liquid_usrp_amd import sharding
    assert sharding.slab_first_sample(rank, T, N) == first_sample

    def step(k):
        i = k % D
        with torch.cuda.stream(streams[i]):
            sharding.step(rxs[i], d_iq, T, rank, world, dist, d_outs[i], d_chans[i], halo=d_halo, stream=streams[i])

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(max(args.warmup, D if D > 1 else 0)):        # with D > 1 every slot is warmed at least once
        step(k)
    fence()
    for rx in rxs:
        rx.kernel_stats(reset=True)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    stats = {}
    for rx in rxs:
        for kname_, (ms_, cnt_) in rx.kernel_stats().items():
            a_, b_ = stats.get(kname_, (0.0, 0))
            stats[kname_] = (a_ + ms_, b_ + cnt_)

    # ---- verification (untimed): the last step of every slot -- all frames of the shard decoded and valid
    expect = cg * reps * world
    nfr, n_ok = 0, 0
    for rx in rxs:
        rx.Flush()
        nfr += len(rx.frames)
        n_ok += sum(1 for f in rx.frames if f.header_valid and f.payload_valid
                    and sent[f.channel][(f.header[0] << 8) | f.header[1]] == (f.header, f.payload))
    verified = (nfr == expect * D and n_ok == expect * D)

    samples_per_step = world * T * K
    value = samples_per_step * args.steps / elapsed / 1e6
    out = None
    if rank == 0:
        per = {k: v[0] / max(v[1], 1) for k, v in stats.items()}          # mean ms per launch
        ch_ms, sc_ms = per["channelizer_kernel"], per["sync_kernel"]
        pl_ms, pw_ms, dk_ms = per["place_jobs_kernel"], per["payload_kernel"], per["decode_kernel"]
        sy_ms = sc_ms + pl_ms + pw_ms + dk_ms
        # algorithmic HBM bytes per launch: the channelizer moves 12 B per wideband
        # sample of its slab (8 read + 4 written); the payload workers read each channel sample of their
        # frames once (4 B per wideband sample) and write 8 B per data symbol plus its soft bits; the
        # scout reads the preamble/header windows only; the decoder reads the soft bits once.
        nsym_frame = -(-8 * ((args.payload + 4) * 3 // 2) // 2)                      # QPSK symbols of one h128-coded frame
        nframes = N * reps
        kbytes = {"channelizer_kernel": B_CHANNELIZER * T * K,
                  "payload_kernel": B_SYNC * world * T * K + nframes * nsym_frame * (8 + 2),
                  "sync_kernel": B_SYNC * world * T * K * (10.0 / 176.0),
                  "decode_kernel": nframes * (nsym_frame * 2 + args.payload),
                  "place_jobs_kernel": nframes * 16.0}
        kname = max(per, key=per.get)                                      # dominant kernel by time
        kms = per[kname]
        achieved = kbytes[kname] / (kms * 1e-3) / 1e9
        traffic = measured_traffic(kname, N, reps, args.payload, world)
        out = {
            "metric": "complex Msamples/s through multichannelrx",
            "value": round(value, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "512-ch multichannelrx (firpfbch K=2N m=7 + N x ofdmflexframesync), "
                                   "M=64 cp=8 taper=4 QPSK CRC32+Hamming128 %dB payloads, %d frames/ch/GPU"
                                   % (args.payload, reps),
                       "channels": N, "subcarriers": M, "samples_per_step": samples_per_step, "steps_in_flight": D,
                       "parallelism": "time-sharded channelizer -> all-to-all -> %d channels/GPU" % cg
                                      if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "ms_per_launch": round(kms, 4),
                         "kernels_ms": {k: round(v, 4) for k, v in per.items()},
                         "channelizer_ms": round(ch_ms, 4), "sync_ms": round(sy_ms, 4),
                         "pipeline_frac_of_16B_roofline": round(value * 1e6 * 16.0 / (world * HBM_PEAK_GBS * 1e9), 5)},
            "verified": {"frames": nfr, "expected": expect * D, "bit_exact_payloads": n_ok, "ok": verified,
                         "note": "last step of each of the %d receiver slots" % D},
            "setup_s": {"iq_generation": round(gen_s, 2)},
        }
        if not args.no_cpu and world == 1:                     # the CPU leg is reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline(ora, d_iq.cpu().numpy(), N, M, cp, taper, args.cpu_reps)
    for rx in rxs:
        rx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))
    if not verified:
        sys.exit("rank %d: verification failed (%d/%d frames, %d ok)" % (rank, nfr, expect * D, n_ok))


def measured_traffic(kernel, N, reps, payload, world):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r1_*_traffic.json:
    2 x FETCH_SIZE + WRITE_SIZE, collected on this workload by scratch/prof.sh); None when the run's
    configuration is not the profiled one.  Counters cannot be read from inside the timed run."""
    import glob
    if (N, reps, payload, world) != (512, 8, 1200, 1):
        return None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r1_*_traffic.json")))
    if not files:
        return None
    prof = json.load(open(files[-1]))
    for name, t in prof.get("kernels", {}).items():
        if kernel in name:
            return round(t["hbm_bytes_per_launch"], 0)
    return None


def usable_cores():
    """Threads worth starting: the affinity mask capped by the container's CPU-time quota (cgroup cpu.max /
    cfs_quota_us) -- oversubscribing a quota gets the process throttled."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    note = "%d logical CPUs visible" % n
    try:
        quota = None
        if os.path.exists("/sys/fs/cgroup/cpu.max"):
            q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            if q != "max":
                quota = float(q) / float(p)
        elif os.path.exists("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        if quota is not None and quota < n:
            note += ", container CPU quota %.4g" % quota
            n = max(1, int(quota))
    except Exception:
        pass
    return max(1, n), note


def cpu_baseline(ora, base, N, M, cp, taper, reps):
    """The CPU oracle (a port: liquid-dsp itself is unavailable) on the same IQ, one thread
    (the reference's multichannelrx is single threaded: lib/multichannelrx.cc:184)."""
    rx = ora.MultiChannelRx(N, M, cp, taper, count_only=True)      # frames are counted in C: only the receiver is timed
    chunk = 1 << 22
    t0 = time.perf_counter()
    for _ in range(reps):
        for i in range(0, len(base), chunk):
            rx.execute(base[i:i + chunk])
    dt = time.perf_counter() - t0
    n = len(base) * reps
    ok = rx.counts()[2]
    out = {"value": round(n / dt / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": "port",
           "sample": "the benchmark's whole GPU slab x%d (%d samples, %d frames decoded), oracle "
                     "multichannelrx, single thread" % (reps, n, ok),
           "host_cores": os.cpu_count(), "seconds": round(dt, 2)}
    # the same oracle on every host core (analysis banks split over time, synchronizers over channels; same
    # frames bit for bit, tests/test_oracle_dsp.py) -- what the reference's loop would allow, not what it does
    try:
        nthr, quota_note = usable_cores()
        rx2 = ora.MultiChannelRx(N, M, cp, taper, count_only=True)
        t0 = time.perf_counter()
        for _ in range(reps):
            rx2.execute_parallel(base, nthr)
        dt2 = time.perf_counter() - t0
        ok2 = rx2.counts()[2]
        out["all_cores"] = {"value": round(n / dt2 / 1e6, 3), "unit": "Msamples/s", "cores": nthr, "seconds": round(dt2, 2),
                            "frames_decoded": ok2, "note": "OpenMP over time blocks (channelizer) and channels (synchronizers); " + quota_note}
    except Exception as e:                                  # the single-thread figure above is the contract
        out["all_cores"] = {"error": str(e)}
    return out


if __name__ == "__main__":
    main()
