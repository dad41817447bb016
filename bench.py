#!/usr/bin/env python3
"""bench.py -- complex Msamples/s through a 512-channel multichannelrx on synthetic IQ.

Workload = BASELINE.json config "512-ch multichannelrx": N=512 channels (K=1024), M=64 subcarriers, cp=8,
taper=4, QPSK, CRC-32 + Hamming(12,8), 1200-byte payloads, frames back to back on every channel (reference
traffic recipe, src/multichannel_tx.cc:163-213), 8 frames per channel and slab.

One step = one pass of the hot path (NCO + polyphase analysis channelizer -> per-channel OFDM frame synchronizer
incl. header/payload decode) over one batch of wideband cf32 samples already resident in HBM.  The batch is
`--slabs` (3) DIFFERENT slabs -- different seeds, different idle gaps between them -- pushed one after the other
through ONE streaming receiver that is never restarted: the stream is continuous across slabs and steps, so the
scouts' frame-position speculation only hits where the traffic really is periodic (`spec_hit_rate`).
`value` leaves the decoded frames in HBM (dropped per slab on the device); `value_with_harvest` is the same loop
with every frame's record + payload copied to pinned host memory and walked through the C-ABI frame iterator
(the reference's callbacks fire inside Execute, lib/multichannelrx.cc:193-194); `value_with_full_harvest` also
moves the equalised symbols (59 KB per frame: host-link bound).

`value_aperiodic` is the same loop on ragged traffic -- every frame its own length (uniform in [64, payload] bytes), irregular
pauses, the recipe of src/multichannel_txrx.cc:227-267 -- where frame positions cannot be predicted.

--gpus N > 1 (one rank per GPU; the script starts its own ranks when it was not launched by torch.distributed.run): weak scaling.  Sub-slabs of the stream go
round robin to the ranks; per round every rank channelizes its sub-slab, one RCCL all-to-all turns the
time-sharded output into channel shards (512/N channels per GPU), the rank synchronizes its shard; rounds are
pipelined (channelize(c+1) || all_to_all(c) || synchronizers(c-1), liquid_usrp_amd/sharding.py).
value = samples all ranks accepted / time.

The input is synthesised on the GPU by the product's own multichanneltx (txgen.hip; untimed, parity-tested
against the oracle's transmitter in tests/test_gpu_tx.py).  The CPU oracle (oracle/) appears only as the
`cpu_baseline` leg: timed on the first slab on one host thread, and its decoded frames / equalised symbols are
compared with the GPU's for that slab.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the receiver handle uses four HIP streams beside the caller's; ROCm maps streams onto four hardware queues by default and a
# shared queue makes the harvest's small copies wait behind long kernels: ask for eight (set before
# the HIP runtime starts; a value the caller exported wins)
QUEUES_BEFORE = os.environ.get("GPU_MAX_HW_QUEUES")                 # what the caller exported (None: nothing)
if not os.environ.get("BENCH_DEFAULT_HW_QUEUES"):                    # (set for the `value_default_hw_queues` leg: the runtime's own default)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec); 6.29 TB/s measured for a float4 copy
B_CHANNELIZER = 12.0           # algorithmic bytes / wideband sample: 8 read + 4 written (N of 2N bins)
B_SYNC = 4.0                   # algorithmic bytes / wideband sample: the kept bins read once


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--channels", type=int, default=512)
    ap.add_argument("--frames", type=int, default=16, help="frames per channel per slab (16 -> 207.7 M wideband samples, 1.66 GB per push; rounds 1-2 "
                                                           "used 8: throughput by slab size is in DESIGN.md section 4.5)")
    ap.add_argument("--slabs", type=int, default=3, help="different slabs per step and GPU")
    ap.add_argument("--payload", type=int, default=1200)
    ap.add_argument("--rounds", type=int, default=3, help="--gpus > 1: exchange rounds per step (a round = one sub-slab per rank)")
    ap.add_argument("--serial-steps", type=int, default=8, help="steps of the unpipelined pass that times each kernel alone")
    ap.add_argument("--harvest-steps", type=int, default=40, help="steps of the harvested legs (their timed region ends with the flush of the two pushes still in flight: ~2 ms, 6 %% of ten steps)")
    ap.add_argument("--cpu-reps", type=int, default=1, help="passes over the first slab timed on the CPU oracle")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-harvest", action="store_true")
    ap.add_argument("--no-aperiodic", action="store_true", help="skip the ragged-traffic leg (value_aperiodic)")
    ap.add_argument("--acquisition", type=int, default=0, help="mcrx_hip_config::acquisition for every receiver of the run (A/B runs: 3 = an anchor phase in front of the segment waves, the default of rounds 4-5)")
    ap.add_argument("--worker-build", type=int, default=0, help="mcrx_hip_config::worker_build for every receiver of the run (A/B runs: 1 = the lean workers with their butterfly exchanges on the VALU)")
    ap.add_argument("--scout-build", type=int, default=0, help="mcrx_hip_config::scout_build for the headline receiver (A/B runs: 2 = the lean segment waves of csrc/acq_lean.hpp; 0 = the general state machine's, the default)")
    ap.add_argument("--no-variants", action="store_true", help="skip the headline's variants (30 dB AWGN on the wideband samples; equalised symbols not stored)")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE.json configurations (the `configs` block)")
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the timed region of --steps steps: value = median, value_min / value_max beside it")
    ap.add_argument("--aperiodic-steps", type=int, default=20)
    ap.add_argument("--slab-blocks", type=int, default=0)
    ap.add_argument("--chunk-blocks", type=int, default=0, help="split every push into sub-slabs of this many blocks")
    ap.add_argument("--defer", type=int, default=-1, help="cfg.defer_samples (experiments; default: 16384 on the pipeline paths, 0 on the direct one)")
    ap.add_argument("--serial", action="store_true", help="time the unpipelined receiver (every kernel in order on one stream)")
    ap.add_argument("--contact-timeout", type=float, default=240.0,
                    help="--gpus > 1: seconds the rendezvous, the first collective and the first exchange round may each take before rank 0 "
                         "prints a line with value 0 and the stage that hung")
    ap.add_argument("--pipeline", action="store_true", help="--gpus 1 through the multi-GPU code path (sharding.Pipeline, exchange = local copy)")
    ap.add_argument("--exchange", choices=["auto", "torch", "c"], default="auto",
                    help="multi-GPU path: 'torch' = sharding.Pipeline (torch streams + torch.distributed all_to_all_single: the path the gloo "
                         "tests drive), 'c' = the C-ABI pipeline (mcrx_hip_pipeline_*: HIP events + grouped ncclSend/ncclRecv).  auto: c on one "
                         "GPU (--pipeline), torch on several -- the C exchange has only ever run at world = 1 (no multi-GPU lease so far)")
    ap.add_argument("--dry-run-launch", action="store_true", help="start the ranks, rendezvous under gloo, print one JSON line and exit (no GPU needed)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="--gpus N > 1 with every rank on device 0 and the exchange staged through host memory under gloo (RCCL refuses two ranks on one "
                         "device): executes the whole multi-process code path -- sub-slab cut, halos, pipeline, verification -- on a one-GPU lease.  "
                         "The value it prints is NOT a scaling number")
    return ap.parse_args()


LEG_CFG = {}           # receiver configuration the command line adds to every leg (--acquisition)


class Watchdog(object):
    """with Watchdog(stage, seconds, rank, args): ...   -- if the block does not finish in time, rank 0 prints the benchmark's JSON line
    with value 0 and the reason, and the process exits with code 3 (a hang inside a collective cannot be interrupted from Python)."""

    def __init__(self, stage, seconds, rank, args):
        self.stage, self.seconds, self.rank, self.args = stage, seconds, rank, args
        self.done = threading.Event()

    def _run(self):
        if self.done.wait(self.seconds):
            return
        if self.rank == 0:
            print(json.dumps({"metric": "complex Msamples/s through multichannelrx", "value": 0.0, "unit": "Msamples/s",
                              "n_gpus": self.args.gpus, "steps": self.args.steps, "warmup": self.args.warmup, "ms_per_step": None,
                              "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                              "config": {"workload": "512-ch multichannelrx", "parallelism": "%d ranks" % self.args.gpus},
                              "error": "timed out after %.0f s in: %s" % (self.seconds, self.stage)}), flush=True)
        sys.stderr.write("bench.py rank %d: timed out after %.0f s in: %s\n" % (self.rank, self.seconds, self.stage))
        sys.stderr.flush()
        os._exit(3)

    def __enter__(self):
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()
        return self

    def __exit__(self, *exc):
        self.done.set()
        return False


def launcher():
    """liquid-usrp_amd/launch.py by path (no torch, no HIP library: the parent of a self-launched job stays light)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mcrx_launch", os.path.join(ROOT, "liquid-usrp_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def frame_index(sent):
    """[(header, payload)] per channel -> dict for verification"""
    return [{(h[0] << 8) | h[1]: (h, p) for (h, p) in ch} for ch in sent]


def main():
    args = parse()
    if args.acquisition:
        LEG_CFG["acquisition"] = args.acquisition
    if args.worker_build:
        LEG_CFG["worker_build"] = args.worker_build
    # `python bench.py --gpus N` is the whole command: without a launcher around it the ranks are started here
    # (torch.distributed.run on 127.0.0.1); under torch.distributed.run (the driver's form for N > 1) this returns the ranks
    if args.dry_run_launch:
        sys.exit(launcher().dry_run(args.gpus))
    rank, world, local = launcher().ensure_ranks(args.gpus)
    import torch
    from __graft_entry__ import load_product, load_oracle
    assert torch.cuda.is_available(), "bench.py needs a GPU: the HIP kernels are the only implementation"
    rehearsal = args.rehearse_on_one_gpu and world > 1
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    ranks_seen = None
    if world > 1:
        # First contact with a multi-GPU node (no SCALE record exists yet): the rendezvous and the first collective run under a
        # watchdog.  If either hangs, rank 0 still prints a JSON line -- value 0, the stage that hung -- and every rank exits non-zero,
        # instead of the driver's clock running out on a silent process.
        import datetime
        import torch.distributed as dist
        limit = float(args.contact_timeout)                          # (the process group's own timeout is a minute longer, so that the watchdog speaks first)
        with Watchdog("rendezvous (init_process_group, %s)" % ("gloo" if rehearsal else "nccl"), limit, rank, args):
            if rehearsal:
                dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=limit + 60))   # every rank on cuda:0, exchange through host memory
            else:
                dist.init_process_group(backend="nccl", device_id=dev, timeout=datetime.timedelta(seconds=limit + 60))  # RCCL over xGMI
        with Watchdog("first collective (all_reduce of one int32 per rank)", limit, rank, args):
            one = torch.ones(1, dtype=torch.int32, device="cpu" if rehearsal else dev)
            dist.all_reduce(one)                                    # sum of ones = the ranks that really took part
            if not rehearsal:
                torch.cuda.synchronize()
            ranks_seen = int(one.item())
        assert ranks_seen == world, "all_reduce saw %d of %d ranks" % (ranks_seen, world)

    prod, ora = load_product(), load_oracle()
    from liquid_usrp_amd import sharding
    N, M, cp, taper = args.channels, 64, 8, 4
    K = 2 * N
    assert N % world == 0
    c0, cg = sharding.shard_of(rank, world, N)

    # ---- synthetic IQ (untimed): world * slabs different slabs = one period of the stream; every slab ends in
    # >= 64 idle blocks, their number differs from slab to slab
    t0 = time.time()
    tx = prod.multichanneltx(N, M, cp, taper)
    nslab = world * args.slabs
    base_blocks = int(prod.lib().mctx_hip_blocks_for(tx._h, args.frames, args.payload, 40, 1, 6))
    slabs, sents = [], []
    for i in range(nslab):
        pad = (0, 80, 32, 144, 48, 112)[i % 6]                       # different idle gaps (blocks; whole tiles)
        if os.environ.get("BENCH_EQUAL_PADS"):                       # (experiment: rounds of the pipeline paths then cut the stream between frames)
            pad = 64
        d, s = tx.generate(args.frames, args.payload, seed=0xC0FFEE + 7919 * i, nblocks=base_blocks + pad, device=dev)
        slabs.append(d); sents.append(frame_index(s))
    torch.cuda.synchronize()
    tx.close()
    gen_s = time.time() - t0
    slab_blocks = [int(d.numel()) // K for d in slabs]
    period_blocks = sum(slab_blocks)
    slab_start = np.concatenate([[0], np.cumsum(slab_blocks)])          # channel-rate sample (= block) index of every slab in the period

    max_frames = N * args.frames + 64 if (world == 1 and not args.pipeline) else cg * args.frames * nslab + 64
    cfg = dict(max_payload_len=max(args.payload, 64), max_frames=max_frames)
    if args.scout_build:
        cfg["scout_build"] = args.scout_build
    if args.acquisition:
        cfg["acquisition"] = args.acquisition
    if args.worker_build:
        cfg["worker_build"] = args.worker_build
    if world > 1 or args.pipeline:
        # rounds cut the stream anywhere: a frame that straddles two rounds is acquired again by the next round (whole, by
        # the parallel path) instead of being walked symbol by symbol -- the history in front of every round covers a frame
        cfg["defer_samples"] = 16384
    if args.defer >= 0:
        cfg["defer_samples"] = args.defer
    if args.slab_blocks:
        cfg["slab_blocks"] = args.slab_blocks
    if args.chunk_blocks:
        cfg["chunk_blocks"] = args.chunk_blocks

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- every kernel alone (serial handle, HIP events on the launch stream): the roofline block's durations.
    # Runs first, so it also brings the clocks up before the timed region.
    ser = prod.multichannelrx(N, M, cp, taper, serial=1, channel_first=c0, channel_count=cg, **cfg)
    ser.kernel_timing(True)                             # (HIP-event pairs around every kernel: off in a new receiver, include/mcrx_hip.h)
    for k in range(args.serial_steps + 2):
        if k == 2:
            torch.cuda.synchronize(); ser.kernel_stats(reset=True)
        for d in slabs[:args.slabs]:
            ser.Execute(d)
            ser.Discard()
    torch.cuda.synchronize()
    ser_stats = ser.kernel_stats()
    ser.close()

    if world == 1 and not args.pipeline:
        rx = prod.multichannelrx(N, M, cp, taper, serial=1 if args.serial else 0, **cfg)

        def step(harvest_rx=None):
            h = harvest_rx or rx
            for d in slabs:
                h.Execute(d)
                if harvest_rx is None:
                    h.Discard()
                else:
                    h.Poll()                        # a result generation holds one slab's frames
        samples_per_step = period_blocks * K
        pipe, trial_steps = None, 0
    else:
        # one period of the stream = nslab slabs; cut into rounds * world sub-slabs, sub-slab u -> rank u % world
        rounds = args.rounds
        stream = torch.cat(slabs)
        unit = rounds * world * 16                                   # sub-slabs are whole tiles (MCRX_TILE blocks)
        tot = (period_blocks + unit - 1) // unit * unit
        if tot > period_blocks:
            stream = torch.cat([stream, torch.zeros((tot - period_blocks) * K, dtype=torch.complex64, device=dev)])
        Tc = tot // (rounds * world)
        v = stream.view(rounds, world, Tc * K)
        mine = [v[c, rank].clone() for c in range(rounds)]
        halos = []
        for c in range(rounds):
            u = c * world + rank
            lo = (u * Tc - 13) * K
            halos.append(stream[lo:lo + 13 * K].clone() if lo >= 0 else stream[lo:].clone())   # u = 0: the period's tail (wraps)
        del stream, v
        slabs_keep0 = slabs[0]
        slabs = None
        torch.cuda.empty_cache()
        period_blocks = tot
        rx = prod.multichannelrx(N, M, cp, taper, channel_first=c0, channel_count=cg, **cfg)
        # auto: the extern-C pipeline (HIP events + grouped ncclSend / ncclRecv behind the C-ABI: what north_star asks for) whenever
        # it can run -- one rank, or several with librccl loadable -- else the torch pipeline, with the reason in the line
        use_c, why_not_c, uid = (args.exchange != "torch"), None, None
        if rehearsal and use_c and not os.environ.get("BENCH_REHEARSE_C"):      # (BENCH_REHEARSE_C=1: try it anyway -- RCCL refuses, which exercises the agreed fall-back below)
            use_c, why_not_c = False, "one-GPU rehearsal: RCCL refuses two ranks on one device, the exchange goes through host memory under gloo"
            if args.exchange == "c":
                sys.exit("--exchange c cannot be rehearsed on one GPU")
        if use_c and world > 1:                         # rank 0's ncclUniqueId to everybody (the C-ABI leaves the transport to the caller)
            box = [None, None]
            if rank == 0:
                try:
                    box[0] = prod.pipeline.unique_id()
                except Exception as e:
                    box[1] = repr(e)
            dist.broadcast_object_list(box, src=0)
            uid = box[0]
            if uid is None:
                if args.exchange == "c":
                    sys.exit("--exchange c: " + str(box[1]))
                use_c, why_not_c = False, "librccl could not be loaded behind the C-ABI (%s)" % box[1]
        nbuf = int(os.environ.get("MCRX_PIPE_NBUF", "5"))
        first_push = [True]
        trial_steps = 0
        if use_c:
            # (five rotating buffer sets = the receiver's own five slots: the channelizer runs as far ahead of the payload workers as in
            #  the direct path; six wait for a slot and halve it.  MCRX_PIPE_NBUF)
            # The grouped ncclSend / ncclRecv exchange behind the C-ABI has never run between two GPUs (no multi-GPU lease so far): it
            # is created and tried for one round of the stream under a guard, the ranks agree on the outcome, and if any of them
            # failed all of them take the torch pipeline on a fresh receiver, with the reason in the line.
            err = None
            try:
                with Watchdog("ncclCommInitRank + the first grouped ncclSend / ncclRecv exchange (C-ABI pipeline)", float(args.contact_timeout), rank, args):
                    pipe = prod.pipeline(rx, rank, world, Tc, unique_id=uid, nbuf=nbuf)
                    if world > 1:                           # one whole step, so that the stream stays a whole number of periods long
                        for c in range(rounds):
                            pipe.push(mine[c], None if first_push[0] else halos[c], ready=True)
                            first_push[0] = False
                        rx.Discard()
                        pipe.wait(); torch.cuda.synchronize()
                        trial_steps = 1
            except Exception as e:
                err, pipe = repr(e), None
            all_ok = err is None
            if world > 1:
                flag = torch.tensor([1 if err is None else 0], dtype=torch.int32, device="cpu" if rehearsal else dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                all_ok = bool(int(flag.item()))
            if not all_ok:
                if args.exchange == "c":
                    sys.exit("--exchange c: the C-ABI pipeline failed on a rank (%s)" % err)
                if pipe is not None:
                    pipe.close()
                rx.close()
                rx = prod.multichannelrx(N, M, cp, taper, channel_first=c0, channel_count=cg, **cfg)
                use_c, why_not_c, first_push[0], trial_steps = False, "the C-ABI pipeline failed in its trial step on at least one rank (this rank: %s)" % err, True, 0
        if not use_c:
            pipe = sharding.Pipeline(rx, rank, world, dist, N, Tc, rx.hist_tiles, device=dev, nbuf=nbuf)

        def step(harvest_rx=None):
            for c in range(rounds):
                if use_c:
                    pipe.push(mine[c], None if first_push[0] else halos[c], ready=True)      # (resident since the fence behind the generator)
                else:
                    pipe.push(mine[c], None if first_push[0] else halos[c])
                first_push[0] = False
            if harvest_rx is None:
                rx.Discard()
        samples_per_step = world * rounds * Tc * K

    for k in range(args.warmup):
        step()
    fence()
    rx.spec_stats(reset=True)
    if pipe is not None:
        pipe.time_exchange(True) if callable(getattr(pipe, "time_exchange", None)) else setattr(pipe, "time_exchange", True)
    # the timed region, --reps times over: every repetition is exactly --steps steps between two fences;
    # value = the median repetition, the spread is reported beside it
    rep_s = []
    for r in range(max(1, args.reps)):
        t0 = time.perf_counter()
        for k in range(args.steps):
            step()
        fence()
        rep_s.append(time.perf_counter() - t0)
    if world > 1:                                      # a repetition takes as long as its slowest rank
        tt = torch.tensor(rep_s, dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        rep_s = [float(v) for v in tt.tolist()]
    elapsed = float(np.median(rep_s))
    nsteps_timed = args.steps * len(rep_s)
    xchg = None
    if pipe is not None:
        pipe.time_exchange(False) if callable(getattr(pipe, "time_exchange", None)) else setattr(pipe, "time_exchange", False)
        xms, xn = pipe.exchange_ms()
        if xn:
            sent = pipe.bytes_sent_per_round()
            # which exchange ran, and how many ranks the transport itself counts (the record must be able to answer "did RCCL see N ranks"):
            # the C pipeline asks its communicator (ncclCommCount); the torch pipeline reports the process group's backend and size, and
            # `ranks_seen_by_all_reduce` is the sum of ones a real all_reduce over that group produced at start-up
            if world == 1:
                xpath, rccl_ranks = "none: one rank, the channelizer writes into the synchronizers' buffer", None
            elif use_c:
                xpath, rccl_ranks = "c-abi: grouped ncclSend/ncclRecv on the pipeline's exchange stream (csrc/pipeline.hip)", pipe.comm_count()
            elif rehearsal:
                xpath, rccl_ranks = "rehearsal: all_to_all staged through host memory under gloo (every rank on one GPU) -- NOT RCCL", None
            else:
                xpath, rccl_ranks = "torch: torch.distributed.all_to_all_single on the %s backend (sharding.Pipeline)" % dist.get_backend(), dist.get_world_size()
            xchg = {"path": xpath, "rccl_ranks": rccl_ranks, "ranks_seen_by_all_reduce": ranks_seen,
                    "ms_per_round": round(xms / xn, 4), "rounds_timed": xn, "bytes_sent_per_rank_and_round": sent,
                    "GBps_out_of_each_rank": round(sent / (xms / xn * 1e-3) / 1e9, 2) if world > 1 else None,
                    "what": ("RCCL, HIP events on the exchange stream of rank 0 (%s)" % ("grouped ncclSend/ncclRecv behind the C-ABI" if use_c else "torch.distributed all_to_all_single")) if world > 1
                            else "local copy standing in for the exchange (one GPU)"}
    walked, adopted = rx.spec_stats()
    # the kernels' durations while they overlap: a few more steps of the same loop with the receiver's event timing switched on,
    # outside the timed region (the event pairs are ten more packets per push; the timed region runs the receiver as it is made)
    stat_steps = 4
    rx.kernel_timing(True); rx.kernel_stats(reset=True)
    for k in range(stat_steps):
        step()
    fence()
    ovl_stats = rx.kernel_stats()
    rx.kernel_timing(False)

    # ---- verification (untimed): one more step of the continuing stream, harvested; every frame of the step
    # decoded, valid and equal to what the transmitter sent
    rx.Flush(); rx.frames.clear()
    step_first = int(args.warmup + nsteps_timed + trial_steps + stat_steps) * period_blocks         # channel-rate sample index where this step starts
    step(harvest_rx=rx)
    rx.Flush()
    nfr, n_ok = len(rx.frames), 0
    for f in rx.frames:
        pos = (f.end_sample - step_first) % period_blocks
        si = int(np.searchsorted(slab_start, pos, side="right") - 1)
        want = sents[min(si, nslab - 1)][f.channel].get((f.header[0] << 8) | f.header[1])
        n_ok += 1 if (f.header_valid and f.payload_valid and want == (f.header, f.payload)) else 0
    expect = cg * args.frames * nslab
    verified = (nfr == expect and n_ok == expect)
    rx.frames.clear()

    value = samples_per_step * args.steps / elapsed / 1e6
    out = None
    if world == 1 and pipe is None:
        # (the legs below make their own receivers: this one's four streams go first, so that theirs do not share hardware queues with
        #  them -- HIP maps a process's streams onto GPU_MAX_HW_QUEUES queues in turn)
        rx.close(); rx = None
    harvest = None
    if world == 1 and not args.no_harvest and not args.pipeline:
        harvest = harvest_legs(prod, N, M, cp, taper, cfg, slabs, K, args, torch)
    aper = None
    if world == 1 and not args.no_aperiodic and not args.pipeline:
        aper = aperiodic_leg(prod, N, M, cp, taper, slab_blocks[:args.slabs], K, args, torch, dev)
    variants = None
    if world == 1 and not args.no_variants and not args.pipeline and not args.serial:
        variants = variant_legs(prod, N, M, cp, taper, cfg, slabs, sents, args, torch, dev)
    cfgs = None
    if world == 1 and not args.no_configs and not args.pipeline:
        cfgs = configs_block(prod, torch, dev, args)
    if rank == 0:
        per = {k: v[0] / max(v[1], 1) for k, v in ser_stats.items()}          # mean ms per launch, kernel alone
        ovl = {k: v[0] / max(v[1], 1) for k, v in ovl_stats.items()}
        blocks0 = slab_blocks[0] if world == 1 else None
        # algorithmic HBM bytes per launch, for one slab of the serial pass: the channelizer
        # moves 12 B per wideband sample (8 read + 4 written); the payload workers read each channel sample of
        # their frames once (4 B per wideband sample) and write 8 B per data symbol plus its soft bits; the scout
        # reads the preamble/header windows only; the decoder reads the soft bits once.
        mean_blocks = float(np.mean(slab_blocks[:args.slabs]))
        nsym_frame = -(-8 * ((args.payload + 4) * 3 // 2) // 2)                      # QPSK symbols of one h128-coded frame
        nframes = cg * args.frames
        share = cg / float(N)
        kbytes = {"channelizer_kernel": B_CHANNELIZER * mean_blocks * K,
                  "payload_kernel": B_SYNC * mean_blocks * K * share + nframes * nsym_frame * (8 + 2),
                  "sync_kernel": B_SYNC * mean_blocks * K * share * (10.0 / 176.0),
                  "decode_kernel": nframes * (nsym_frame * 2 + args.payload),
                  "place_jobs_kernel": nframes * 16.0}
        # the dominant kernel: of the kernels within 10 % of the longest mean launch, the one furthest below its own
        # roofline (two chip-filling kernels take about the same time here; the slower-per-byte one is the one to fix)
        longest = max(per[k] for k in per if k in kbytes)
        fracs = {k: kbytes[k] / (per[k] * 1e-3) / 1e9 / HBM_PEAK_GBS for k in per if k in kbytes and per[k] > 0}
        kname = min((k for k in fracs if per[k] >= 0.9 * longest), key=lambda k: fracs[k])
        kms = per[kname]
        achieved = kbytes[kname] / (kms * 1e-3) / 1e9
        traffic = measured_traffic(kname, N, args.frames, args.payload)
        total = walked + adopted
        out = {
            "metric": "complex Msamples/s through multichannelrx",
            "value": round(value, 3), "unit": "Msamples/s",
            "value_min": round(samples_per_step * args.steps / max(rep_s) / 1e6, 3), "value_max": round(samples_per_step * args.steps / min(rep_s) / 1e6, 3),
            "repetitions": len(rep_s), "timing": "median of %d repetitions of %d steps, each between two fences (barrier + device synchronize)" % (len(rep_s), args.steps),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "512-ch multichannelrx (firpfbch K=2N m=7 + N x ofdmflexframesync), "
                                   "M=64 cp=8 taper=4 QPSK CRC32+Hamming128 %dB payloads, %d frames/ch/slab, "
                                   "%d different slabs per step and GPU, one continuous un-restarted stream"
                                   % (args.payload, args.frames, args.slabs),
                       "channels": N, "subcarriers": M, "samples_per_step": samples_per_step,
                       "receiver": "serial (one stream)" if args.serial else "pipelined (3 internal streams, 3 buffer sets)",
                       "multi_gpu_path": None if pipe is None else ("C-ABI pipeline (mcrx_hip_pipeline_*)" if use_c else
                                                                      "sharding.Pipeline (torch)" + (": " + why_not_c if why_not_c else "")),
                       "rehearsal_on_one_gpu": bool(rehearsal),
                       "receiver_hints_from_the_benchmark": "none (no MCRX_* environment is set by this script)",
                       "runtime_knobs": {"GPU_MAX_HW_QUEUES": "%s (%s)" % (os.environ.get("GPU_MAX_HW_QUEUES", "runtime default"),
                                                                            "exported by the caller" if QUEUES_BEFORE is not None else
                                                                            ("left at the runtime's default for this leg" if os.environ.get("BENCH_DEFAULT_HW_QUEUES") else
                                                                             "set by bench.py before the HIP runtime starts: the handle's streams get a hardware queue each; "
                                                                             "`value_default_hw_queues` is the same loop without it"))},
                       "channel": "noise-free loopback of the GPU transmitter (BASELINE.json's synthetic source).  One stage's cost depends on that: the "
                                  "Hamming(12,8) soft decision forms its neighbour distances only in waves with a non-zero syndrome (exact; "
                                  "decode_kernel 0.115 ms here, 0.138 ms when every wave has one: DESIGN.md section 4.2).  With white noise on the wideband samples "
                                  "the same stream runs at 181.5 / 180.2 / 181.2 / 174.8 Gsample/s at 20 / 10 / 6 / 3 dB against 182.3 clean, "
                                  "all frames valid down to 10 dB, 96.6 % at 6 dB, none at 3 (profiles/r4_noise_probe.jsonl, scratch/noise_probe.py)",
                       "parallelism": ("round-robin time-sharded channelizer -> all-to-all -> %d channels/GPU, %d rounds per step, "
                                       "exchange overlapped" % (cg, args.rounds)) if world > 1 else "single GPU"},
            "spec_hit_rate": round(adopted / total, 4) if total else None,
            "frames_acquired": {"by_scout_walk": walked, "adopted_from_speculation": adopted},
            "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "ms_per_launch": round(kms, 4),
                         "measured": "kernel alone: serial receiver, %d steps, HIP events on the launch stream" % args.serial_steps,
                         "algorithmic_bytes_per_launch": round(kbytes[kname], 0),
                         "traffic_source": traffic_source(),
                         "kernels_ms": {k: round(v, 4) for k, v in per.items()},
                         "kernels_frac_of_peak": {k: round(v, 5) for k, v in fracs.items()},
                         "kernels_ms_overlapped": {k: round(v, 4) for k, v in ovl.items()},
                         "serial_sum_ms_per_slab": round(sum(per.values()), 4),
                         "vector_issue": safely(vector_issue, value, world, N, args.frames, args.payload, samples_per_step / max(1, args.slabs) / world),
                         "pipeline_frac_of_16B_roofline": round(value * 1e6 * 16.0 / (world * HBM_PEAK_GBS * 1e9), 5),
                         "per_gpu": {"Msamples_per_s": round(value / world, 3), "GBps_at_16B_per_sample": round(value / world * 16e-3, 2),
                                     "peak_GBps": HBM_PEAK_GBS},
                         "aggregate": {"Msamples_per_s": round(value, 3), "GBps_at_16B_per_sample": round(value * 16e-3, 2),
                                       "peak_GBps": world * HBM_PEAK_GBS}},
            "verified": {"frames": nfr, "expected": expect, "bit_exact_payloads": n_ok, "ok": verified,
                         "note": "the step after the timed region, same continuing stream"},
            "setup_s": {"iq_generation": round(gen_s, 2)},
        }
        if xchg:
            out["exchange"] = xchg
        if harvest:
            out.update(harvest)
        if aper:
            out.update(aper)
            out["value_aperiodic_over_value"] = round(aper["value_aperiodic"] / value, 4)
        if variants:
            out.update(variants)
        if cfgs:
            out["configs"] = cfgs
        if world == 1 and not args.no_variants and not args.pipeline and not args.serial and not os.environ.get("BENCH_DEFAULT_HW_QUEUES") and QUEUES_BEFORE is None:
            dq = safely(default_queues_leg, args)
            out.update(dq if "error" not in dq else {"value_default_hw_queues": None, "value_default_hw_queues_note": dq["error"]})
        if not args.no_cpu and world == 1:                     # the CPU leg is reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline(ora, prod, slabs[0] if slabs is not None else slabs_keep0, N, M, cp, taper, args.cpu_reps, cfg)
    if rx is not None:
        rx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))
    if not verified:
        sys.exit("rank %d: verification failed (%d/%d frames, %d ok)" % (rank, nfr, expect, n_ok))


def default_queues_leg(args):
    """`value` once more in a process of its own WITHOUT GPU_MAX_HW_QUEUES=8 (the variable is read when the HIP runtime starts): the headline
    depends on a process-level runtime knob, so the line says what it is worth."""
    import subprocess
    env = dict(os.environ); env.pop("GPU_MAX_HW_QUEUES", None); env["BENCH_DEFAULT_HW_QUEUES"] = "1"
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(max(10, args.steps // 2)), "--warmup", str(args.warmup), "--reps", "3",
           "--channels", str(args.channels), "--frames", str(args.frames), "--slabs", str(args.slabs), "--payload", str(args.payload),
           "--no-cpu", "--no-harvest", "--no-aperiodic", "--no-variants", "--no-configs", "--serial-steps", "2"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
    d = json.loads(r.stdout.decode().strip().split("\n")[-1])
    return {"value_default_hw_queues": d["value"], "value_default_hw_queues_note": "the same loop, same library, GPU_MAX_HW_QUEUES unset (%s), a process of its own; verified %s"
            % (d["config"]["runtime_knobs"]["GPU_MAX_HW_QUEUES"], d["verified"]["ok"])}


def config_leg(prod, torch, dev, N, M, cp, frames, plen, mod, fec1, resamp, steps, reps, what, front_end=0):
    """One of the other BASELINE.json configurations as a continuous stream through ONE un-restarted receiver, like `value`:
    two different slabs (seeds, idle tails) pushed alternately, frames dropped on the device, then one more step harvested and
    every frame checked against what was sent.  resamp: the slabs are 2x oversampled (zero stuffing + half-band low-pass, made
    once, untimed) and go through msresamp(0.5) in front of the receiver (src/multichannel_rx.cc:129-138); the unit is then a
    sample entering the resampler, 20 algorithmic bytes each (8 read + 4 written by the resampler, 16 / 2 behind it)."""
    K, taper = 2 * N, 4
    tx = prod.multichanneltx(N, M, cp, taper)
    base = int(prod.lib().mctx_hip_blocks_for(tx._h, frames, plen, mod, 1, fec1))
    slabs, sents = [], []
    for i in range(2):
        d, sent = tx.generate(frames, plen, mod=mod, fec1=fec1, seed=0xBEEF + 7919 * i, nblocks=base + (0, 48)[i], device=dev)
        slabs.append(d); sents.append(frame_index(sent))
    torch.cuda.synchronize()
    tx.close()
    inputs, rs = slabs, None
    if resamp:
        # the antenna stream at twice the rate, made once (untimed) by the transmit applications' own interpolator,
        # msresamp_crcf_create(2.0, 60) (src/flexframe_tx.cc:170) -- the product's GPU build of it, one continuous stream over the slabs
        up = prod.msresamp(2.0, 60.0)
        inputs = [up.execute(d).clone() for d in slabs]
        torch.cuda.synchronize()
        up.close()
        rs = prod.msresamp(0.5, 60.0)
    leg_cfg = dict(LEG_CFG, front_end=front_end) if front_end else LEG_CFG
    rx = prod.multichannelrx(N, M, cp, taper, max_payload_len=plen, max_frames=N * frames + 64, **leg_cfg)
    tile = prod.TILE * K

    # resampler and receiver on ONE caller stream that is not the legacy default stream (work on the NULL stream is a
    # barrier across the handle's blocking streams -- consecutive pushes then run one after the other, 50 instead of 80+ Gsample/s
    # here); its output buffer is allocated under that stream, so the next push's resampler is ordered behind the channelizer that
    # reads it (mcrx_hip_execute_device lets the caller's stream wait until the input has been read)
    # Two such streams, taken in turn (round 6): Execute(y, stream) makes the caller's stream wait until the channelizer has read y, so with
    # ONE stream push p + 1's resampler could not start before push p's channelizer had finished -- resampler and channelizer alternated,
    # 0.83 of every 0.94 ms (profiles/r6_c2_timeline_before.txt); with two, the resampler of the next push runs beside the receiver's
    # work on this one (each stream's output buffer comes from its own pool of the caching allocator)
    nside = int(os.environ.get("BENCH_RESAMP_STREAMS", "2"))
    sides = [torch.cuda.Stream(device=dev) for _ in range(nside)] if rs is not None else []
    turn, rs_done = [0], [None]

    def step(keep=False):
        for x in inputs:
            if rs is not None:
                side = sides[turn[0] % nside]; turn[0] += 1
                with torch.cuda.stream(side):
                    if rs_done[0] is not None:
                        side.wait_event(rs_done[0])             # the resampler is one stream of samples: its filter state passes from call to call
                    y = rs.execute(x, stream=side)
                    rs_done[0] = torch.cuda.Event(); rs_done[0].record(side)
                    assert int(y.numel()) % tile == 0, "the decimated slab is not whole tiles (%d samples)" % int(y.numel())
                    rx.Execute(y, stream=side)
            else:
                rx.Execute(x)
            rx.Poll() if keep else rx.Discard()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    rx.spec_stats(reset=True)
    rep_s = []
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        rep_s.append(time.perf_counter() - t0)
    walked, adopted = rx.spec_stats()
    rx.kernel_timing(True); rx.kernel_stats(reset=True)         # (outside the timed region: a few more steps with the event pairs switched on)
    for _ in range(max(2, steps // 2)):
        step()
    torch.cuda.synchronize()
    ovl = {k: round(v[0] / max(v[1], 1), 4) for k, v in rx.kernel_stats().items()}
    rx.kernel_timing(False)
    rx.Flush(); rx.frames.clear()
    step(keep=True); rx.Flush()
    nfr = len(rx.frames)
    ok = sum(1 for f in rx.frames if f.header_valid and f.payload_valid and
             any(sn[f.channel].get((f.header[0] << 8) | f.header[1]) == (f.header, f.payload) for sn in sents))
    rx.close()
    if rs is not None:
        rs.close()
    n_in = sum(int(x.numel()) for x in inputs)
    roof = None
    if front_end:
        # the front-end kernel alone (a serial receiver: every kernel of a push in order on one stream, HIP events on that stream),
        # against the same 12 algorithmic bytes per wideband sample as the reference's bank: 8 read + 4 written
        ser = prod.multichannelrx(N, M, cp, taper, max_payload_len=plen, max_frames=N * frames + 64, serial=1, **leg_cfg)
        ser.kernel_timing(True)
        for k in range(5):
            if k == 2:
                torch.cuda.synchronize(); ser.kernel_stats(reset=True)
            for x in inputs:
                ser.Execute(x); ser.Discard()
        torch.cuda.synchronize()
        cms, cn = ser.kernel_stats()["channelizer_kernel"]
        ser.close()
        per_launch = n_in / len(inputs)
        ms1 = cms / max(cn, 1)
        ach = B_CHANNELIZER * per_launch / (ms1 * 1e-3) / 1e9
        traffic = front_end_traffic(N)
        roof = {"kernel": "channelizer_kernel<%d,2,%d,28,shift> (oscillator + firpfbch2 analysis + half-band decimator per kept channel, folded into one bank)" % (K, 512 if K >= 512 else 256),
                "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                "kernel_ms_alone": round(ms1, 4), "launches_timed": int(cn), "samples_per_launch": int(per_launch),
                "algorithmic_bytes_per_sample": B_CHANNELIZER,
                "traffic": traffic, "traffic_over_algorithmic": round(traffic / (B_CHANNELIZER * per_launch), 3) if traffic else None,
                "traffic_source": "profiles/r6_pfb2_traffic.json (rocprofv3 --pmc passes of scratch/r6/prof_pfb2.sh: 2 x FETCH_SIZE + WRITE_SIZE per launch)" if traffic else None,
                "unfused_chain_bytes_per_sample": 52.0,
                "note": "front_end = 2 runs the same chain as three kernels (oscillator pass 8 + 8, bank at twice the rate 8 + 16, adapter 8 + 4 bytes per sample)"}
    bytes_per = 20.0 if resamp else 16.0
    med = float(np.median(rep_s))
    val = n_in * steps / med / 1e6
    tot = walked + adopted
    return {"workload": what, "value": round(val, 1), "unit": "Msamples/s" + (" entering the resampler" if resamp else ""),
            "value_min": round(n_in * steps / max(rep_s) / 1e6, 1), "value_max": round(n_in * steps / min(rep_s) / 1e6, 1),
            "repetitions": reps, "steps": steps, "samples_per_step": n_in, "ms_per_step": round(med / steps * 1e3, 4),
            "algorithmic_bytes_per_sample": bytes_per, "frac_of_roofline": round(val * 1e6 * bytes_per / (HBM_PEAK_GBS * 1e9), 5),
            "frames_acquired": {"by_scout_walk": walked, "adopted_from_segment_waves": adopted, "walked_share": round(walked / tot, 5) if tot else None},
            "kernels_ms_overlapped": ovl, **({"roofline": roof} if roof else {}),
            "verified": {"frames": nfr, "expected": 2 * N * frames, "bit_exact_payloads": ok, "ok": nfr == 2 * N * frames and ok == nfr}}


def front_end_traffic(N):
    """HBM bytes per launch of the folded front-end kernel from the committed counter passes (profiles/r6_pfb2_traffic.json), None if absent
    or of another channel count"""
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", "r6_pfb2_traffic.json")))
        return round(prof["hbm_bytes_per_launch"], 0) if int(prof.get("channels", 0)) == N else None
    except (OSError, ValueError, KeyError):
        return None


def configs_block(prod, torch, dev, args):
    """BASELINE.json configs[1], configs[1] with the K = 7 rate-1/2 code, configs[2] and configs[4] on one GPU (configs[3] is the
    headline, configs[0] a CPU plumbing case covered by tests/test_gpu_baseline_shapes.py)."""
    out = {}
    legs = (("8ch", 8, 64, 8, 100, 1200, 40, 6, False, "configs[1]: 8-ch multichannelrx, M=64 cp=8 QPSK CRC32+Hamming128 1200B payloads, 100 frames/ch/slab"),
            ("8ch_long_pushes", 8, 64, 8, 400, 1200, 40, 6, False,
             "configs[1] in pushes four times as long (400 frames/ch/slab, 54 M samples): a push is a chain of latencies -- acquisition, payload workers, decoder, "
             "0.45 ms at 100 frames per channel whatever the sample count -- that eight channels cannot fill the chip beside; longer pushes amortize it"),
            ("8ch_v27", 8, 64, 8, 100, 1200, 40, 11, False, "configs[1] with the K=7 r=1/2 convolutional code (soft Viterbi) as the outer code"),
            ("512ch_m48_reference_app_defaults", 512, 48, 6, 16, 1200, 40, 6, False,
             "the reference applications' own default numerology (src/multichannel_rx.cc:93-95, src/multichannel_tx.cc: M=48 cp=6 taper=4) at 512 channels, "
             "QPSK CRC32+Hamming128 1200B payloads, 16 frames/ch/slab: 48 = 3 x 16 on the lean path (segment waves, one-frame-per-wave workers) since round 5; "
             "rounds 1-4: direct DFTs, every frame walked, 20.4 Gsample/s"),
            ("64ch_m256_qam16_resamp", 64, 256, 32, 32, 1200, 27, 7, True,
             "configs[2]: 64-ch multichannelrx, M=256 cp=32 QAM16 CRC32+Golay(24,12) 1200B payloads, 32 frames/ch/slab, msresamp(0.5) front end"))
    legs = tuple(l + (0,) for l in legs) + (
            ("512ch_pfb2_front_end", 512, 64, 8, 16, 1200, 40, 6, False,
             "the channelizer BASELINE.json's north_star names in front of the headline's receiver: 512-ch multichannelrx with cfg.front_end = 1 -- oscillator, "
             "firpfbch2 analysis bank (1024 channels at twice the channel rate) and the half-band decimator of every kept channel as ONE kernel "
             "(a 28-tap composite bank, csrc/channelizer.hip) -- M=64 cp=8 QPSK CRC32+Hamming128 1200B payloads, 16 frames/ch/slab", 1),)
    for name, N, M, cp, fr, pl, mod, fec1, rsmp, what, fe in legs:
        try:
            out[name] = config_leg(prod, torch, dev, N, M, cp, fr, pl, mod, fec1, rsmp, steps=6, reps=3, what=what, front_end=fe)
        except Exception as e:                                   # a leg that fails says so in the line; the headline stands
            out[name] = {"workload": what, "error": repr(e), "verified": {"ok": False}}
        torch.cuda.empty_cache()
    try:
        import bench_duplex
        d, ok, msg = bench_duplex.measure(prod, torch, dev, 0, 1, None, steps=8, warmup=3)
        out["256ch_duplex_one_gpu"] = {"workload": "configs[4] on ONE GPU: " + d["config"]["workload"], "value": d["value"],
                                       "unit": "Msamples/s transmitted and received", "ms_per_step": d["ms_per_step"], "steps": 8,
                                       "algorithmic_bytes_per_sample": 28.0,
                                       "frac_of_roofline": round(d["value"] * 1e6 * 28.0 / (HBM_PEAK_GBS * 1e9), 5),
                                       "verified": d["verified"],
                                       "note": "28 B = 16 (receive) + 12 (transmit: 4 B of channel-rate granules read + 8 B written per wideband sample)"}
    except Exception as e:
        out["256ch_duplex_one_gpu"] = {"error": repr(e), "verified": {"ok": False}}
    torch.cuda.empty_cache()
    return out


def aperiodic_leg(prod, N, M, cp, taper, slab_blocks, K, args, torch, dev):
    """Worst-case traffic for the frame-position predictor: the recipe of src/multichannel_txrx.cc:227-267 -- every packet its
    own length (uniform in [64, payload] bytes here), 0..3 idle OFDM symbols before each frame and a silence of 16..200 symbols
    once in 8 frames, independently on all channels.  Same receiver, same loop as `value`: different slabs through one
    continuous stream, frames dropped on the device; then one more step harvested and every frame checked against what was sent."""
    tx = prod.multichanneltx(N, M, cp, taper)
    slabs, sents = [], []
    for i, nb in enumerate(slab_blocks):
        d, s, _ = tx.generate_ragged(nb, len_lo=64, len_hi=args.payload, gap_max=3, long_every=8, long_max=184,
                                     seed=0xA9E210 + 104729 * i, device=dev)
        slabs.append(d); sents.append(frame_index(s))
    torch.cuda.synchronize()
    tx.close()
    nfr_slab = [sum(len(c) for c in s) for s in sents]
    cfg = dict(max_payload_len=max(args.payload, 64), max_frames=max(nfr_slab) + 64)
    if args.scout_build:
        cfg["scout_build"] = args.scout_build
    if args.acquisition:
        cfg["acquisition"] = args.acquisition
    if args.worker_build:
        cfg["worker_build"] = args.worker_build
    rx = prod.multichannelrx(N, M, cp, taper, **cfg)

    def step(keep=False):
        for d in slabs:
            rx.Execute(d)
            rx.Poll() if keep else rx.Discard()
    nwarm = max(8, args.warmup)          # (the acquisition policy needs two windows of 8 launches to settle on this traffic)
    for _ in range(nwarm):
        step()
    torch.cuda.synchronize()
    rx.kernel_timing(True)               # (this leg reports the overlapped kernel durations of its own timed steps: ten event packets per 1.4 ms push)
    rx.kernel_stats(reset=True); rx.spec_stats(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.aperiodic_steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    walked, adopted = rx.spec_stats()
    ovl = {k: v[0] / max(v[1], 1) for k, v in rx.kernel_stats().items()}
    rx.Flush(); rx.frames.clear()
    step(keep=True); rx.Flush()
    period = sum(slab_blocks)
    start = np.concatenate([[0], np.cumsum(slab_blocks)])
    base = (nwarm + args.aperiodic_steps) * period
    nfr, ok, wrong, noise = len(rx.frames), 0, 0, 0
    for f in rx.frames:
        si = int(np.searchsorted(start, (f.end_sample - base) % period, side="right") - 1)
        want = sents[min(si, len(sents) - 1)][f.channel].get((f.header[0] << 8) | f.header[1])
        if f.header_valid and f.payload_valid:
            ok += 1 if want == (f.header, f.payload) else 0
            wrong += 0 if want == (f.header, f.payload) else 1
        else:
            noise += 1          # a channel that idles locks onto its neighbours' leakage now and then (the detector is gain-normalised;
                                # the oracle does the same: tests/test_gpu_tx.py); what it decodes there fails the header check
    rx.close()
    samples = sum(int(d.numel()) for d in slabs)
    tot = walked + adopted
    return {"value_aperiodic": round(samples * args.aperiodic_steps / dt / 1e6, 3),
            "value_aperiodic_detail": {"traffic": "payload lengths uniform in [64, %d] B, 0..3 idle symbols before every frame, 16..200 once in 8 "
                                                  "(src/multichannel_txrx.cc:227-267), %d different slabs" % (args.payload, len(slabs)),
                                       "steps": args.aperiodic_steps, "frames_per_step": sum(nfr_slab),
                                       "spec_hit_rate": round(adopted / tot, 4) if tot else None,
                                       "frames_acquired": {"by_scout_walk": walked, "adopted_from_speculation": adopted},
                                       "kernels_ms_overlapped": {k: round(v, 4) for k, v in ovl.items()},
                                       "verified": {"frames": nfr, "sent": sum(nfr_slab), "bit_exact_payloads": ok, "valid_but_different": wrong,
                                                    "failed_header_or_crc": noise,
                                                    "note": "idle channels lock onto neighbour leakage now and then and may miss the frame that "
                                                            "starts underneath (reference behaviour, same in the oracle): ok = nothing valid differs "
                                                            "from what was sent and >= 99.5 % of the sent frames arrive",
                                                    "ok": wrong == 0 and ok >= 0.995 * sum(nfr_slab)}}}


def variant_legs(prod, N, M, cp, taper, cfg, slabs, sent_idx, args, torch, dev):
    """The headline's loop -- the same slabs through one un-restarted receiver, frames dropped on the device -- under two changed
    conditions, each verified on one more harvested step:
    value_awgn30: white noise 30 dB below the wideband signal on every sample (BASELINE.md section 2 / SURVEY 8d: "optional seeded
      AWGN at 30 dB SNR (seed 1)"): the clean channel is the one place where a stage's cost depends on the data (decode_kernel forms
      its Hamming neighbour distances only in waves with a non-zero syndrome), so the driver's line carries the noisy figure too;
    value_without_framesyms: mcrx_hip_config::skip_framesyms = 2 -- the payload workers do not store the equalised symbols (8 B per
      data symbol, a third of what that stage moves): what materialising stats.framesyms for callbacks that never read it costs."""
    res = {}
    samples = sum(int(d.numel()) for d in slabs)
    steps = max(4, args.steps // 2)
    g = torch.Generator(device=dev); g.manual_seed(1)
    noisy = []
    for d in slabs:
        v = torch.view_as_real(d)
        rms = float(torch.sqrt(torch.mean(v * v) * 2.0))                   # rms of the complex samples
        nstd = rms * 10.0 ** (-30.0 / 20.0) / np.sqrt(2.0)
        noisy.append(torch.view_as_complex(v + nstd * torch.randn(v.shape, generator=g, device=dev, dtype=v.dtype)))
    for name, inputs, extra in (("value_awgn30", noisy, {}), ("value_without_framesyms", slabs, {"skip_framesyms": 2})):
        rx = prod.multichannelrx(N, M, cp, taper, **dict(cfg, **extra))

        def step(keep=False):
            for d in inputs:
                rx.Execute(d)
                rx.Poll() if keep else rx.Discard()
        for _ in range(max(2, args.warmup)):
            step()
        torch.cuda.synchronize()
        rx.kernel_timing(True)           # (as in the ragged-traffic leg)
        rx.kernel_stats(reset=True)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ovl = {k: round(v[0] / max(v[1], 1), 4) for k, v in rx.kernel_stats().items()}
        rx.Flush(); rx.frames.clear()
        step(keep=True); rx.Flush()
        nfr = len(rx.frames)
        ok = sum(1 for f in rx.frames if f.header_valid and f.payload_valid and
                 any(sn[f.channel].get((f.header[0] << 8) | f.header[1]) == (f.header, f.payload) for sn in sent_idx))
        rx.close()
        res[name] = round(samples * steps / dt / 1e6, 3)
        res[name + "_detail"] = {"steps": steps, "kernels_ms_overlapped": ovl,
                                 "verified": {"frames": nfr, "expected": N * args.frames * len(slabs), "bit_exact_payloads": ok,
                                              "ok": nfr == N * args.frames * len(slabs) and ok == nfr}}
    del noisy
    torch.cuda.empty_cache()
    return res


def harvest_legs(prod, N, M, cp, taper, cfg, slabs, K, args, torch):
    """The same continuous stream with the frames delivered: push slab k, poll (slab k-1's records + payloads come
    over the host link while slab k is processed), walk them through mcrx_hip_next_frame."""
    res = {}
    samples = sum(int(d.numel()) for d in slabs)
    for name, skip, steps in (("value_with_harvest", 1, args.harvest_steps), ("value_with_full_harvest", 0, max(2, args.harvest_steps // 3))):
        rx = prod.multichannelrx(N, M, cp, taper, skip_framesyms=skip, **cfg)
        for d in slabs:                                   # warm (buffers, pinned arena growth)
            rx.Execute(d); rx.Poll(deliver=False); rx.drain_count()
        rx.Flush(); rx.drain_count()
        torch.cuda.synchronize()
        nfr = nok = nbytes = 0
        t0 = time.perf_counter()
        for _ in range(steps):
            for d in slabs:
                rx.Execute(d)
                rx.Poll(deliver=False)
                a, b, c = rx.drain_count(); nfr += a; nok += b; nbytes += c
        lib_rc = prod.lib().mcrx_hip_flush(rx._h)
        a, b, c = rx.drain_count(); nfr += a; nok += b; nbytes += c
        dt = time.perf_counter() - t0
        res[name] = round(samples * steps / dt / 1e6, 3)
        res[name + "_detail"] = {"steps": steps, "frames_delivered": nfr, "valid": nok, "payload_bytes": nbytes,
                                 "expected_frames": N * args.frames * len(slabs) * steps, "dropped": rx.frames_dropped(),
                                 "equalised_symbols_to_host": not skip, "flush_rc": int(lib_rc)}
        rx.close()
    return res


def traffic_files():
    """the committed counter summaries of the receiver's benchmark command, oldest first (profiles/r*_traffic.json with a `workload`
    line: the transmit side's summaries, profiles/r*_tx_traffic.json, have none)"""
    import glob
    out = []
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json"))):
        try:
            if "multichannelrx" in json.load(open(f)).get("workload", ""):
                out.append(f)
        except (OSError, ValueError):
            pass
    return out


def traffic_source():
    files = traffic_files()
    return ("profiles/" + os.path.basename(files[-1]) + " (rocprofv3 --pmc passes of this command on this workload, 2 x FETCH_SIZE + "
            "WRITE_SIZE per launch; counters cannot be read from inside the timed run)") if files else None


def measured_traffic(kernel, N, frames, payload):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/r*_traffic.json:
    2 x FETCH_SIZE + WRITE_SIZE, collected on this workload by scratch/prof.sh); None when the run's
    configuration is not the profiled one.  Counters cannot be read from inside the timed run."""
    import re
    files = traffic_files()
    if not files:
        return None
    prof = json.load(open(files[-1]))
    m = re.search(r"(\d+)-ch multichannelrx.*?(\d+)B payloads, (\d+) frames/ch/slab", prof.get("workload", ""))
    if not m or (int(m.group(1)), int(m.group(3)), int(m.group(2))) != (N, frames, payload):
        return None                                     # the committed counters are of another workload
    alias = {"payload_kernel": ("payload_lean_kernel", "payload_multi_kernel", "payload_kernel"), "sync_kernel": ("sync_lean_kernel", "sync_kernel")}
    for want in alias.get(kernel, (kernel,)):
        for name, t in prof.get("kernels", {}).items():
            if want in name:
                return round(t["hbm_bytes_per_launch"], 0)
    return None


def safely(fn, *a):
    """(an informational block must never cost the benchmark its line)"""
    try:
        return fn(*a)
    except Exception as e:                                   # noqa: BLE001
        return {"error": repr(e)}


def vector_issue(value, world, N, frames, payload, slab_samples):
    """The other ceiling of this path: wave64 vector instructions per slab (SQ_INSTS_VALU of the receiver's kernels, from the committed
    rocprofv3 --pmc passes of this command: profiles/r*_pmc.csv beside the traffic summary) and the rate the timed run issued them at,
    against 1024 SIMDs x 2.4 GHz / 4 cycles.  DESIGN.md section 4: a kernel of nothing but register arithmetic sustains 330-410 G/s here."""
    import csv
    files = traffic_files()
    if not files or measured_traffic("channelizer_kernel", N, frames, payload) is None:
        return None                                     # (the committed counters are of another workload)
    pmc, stats = files[-1].replace("_traffic.json", "_pmc.csv"), files[-1].replace("_traffic.json", "_kernel_stats.csv")
    if not (os.path.exists(pmc) and os.path.exists(stats)):
        return None
    not_rx = ("syn::", "txsym", "txifft", "txfir", "ilmap", "reset")
    per = {r["kernel"]: float(r["mean_per_dispatch"]) for r in csv.DictReader(open(pmc))
           if r["counter"] == "SQ_INSTS_VALU" and r["kernel"].startswith("mcrx::") and not any(t in r["kernel"] for t in not_rx)}
    calls = {r["Name"]: int(r["Calls"]) for r in csv.DictReader(open(stats))}
    chan = max([c for n, c in calls.items() if "channelizer_kernel" in n] or [0])
    if not per or not chan:
        return None
    table, total = {}, 0.0
    for k, n in per.items():
        c = max([c for nm, c in calls.items() if nm.startswith(k) or nm.startswith("void " + k)] or [chan])
        table[k.replace("mcrx::", "")] = round(n * c / chan / 1e6, 2)          # dispatches per slab = calls relative to the channelizer's
        total += n * c / chan
    rate = total * (value * 1e6 / world) / slab_samples / 1e9
    return {"unit": "wave64 vector instructions", "per_slab_millions": table, "per_slab_total_millions": round(total / 1e6, 1),
            "rate_G_per_s": round(rate, 1), "peak_G_per_s": 614.4, "frac": round(rate / 614.4, 4),
            "peak": "1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction; a kernel of nothing but register arithmetic sustains 330-410 G/s on this chip (DESIGN.md section 4)",
            "source": "profiles/" + os.path.basename(pmc) + " (SQ_INSTS_VALU per dispatch x dispatches per slab) and this run's rate"}


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


def cpu_baseline(ora, prod, d_slab, N, M, cp, taper, reps, cfg):
    """The CPU oracle (a port: liquid-dsp itself is unavailable) on the first slab, one thread (the reference's
    multichannelrx is single threaded: lib/multichannelrx.cc:184); then its frames against the GPU's."""
    base = d_slab.cpu().numpy()
    # one single-thread pass serves the timing AND the parity comparison below: it keeps its frames (a Python callback per
    # frame -- 8192 of them in ~10 s of receiver time, well under 1 % of it)
    rx = ora.MultiChannelRx(N, M, cp, taper)
    chunk = 1 << 22
    t0 = time.perf_counter()
    for _ in range(reps):
        if _:
            rx.frames.clear(); rx.reset()
        for i in range(0, len(base), chunk):
            rx.execute(base[i:i + chunk])
    dt = time.perf_counter() - t0
    n = len(base) * reps
    ok = sum(1 for f in rx.frames if f.payload_valid)
    out = {"value": round(n / dt / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": "port",
           "sample": "the benchmark's first slab x%d (%d samples, %d frames decoded), oracle "
                     "multichannelrx, single thread" % (reps, n, ok),
           "host_cores": os.cpu_count(), "seconds": round(dt, 2)}
    # the same oracle on every host core (analysis banks split over time, synchronizers over channels; same
    # frames bit for bit, tests/test_oracle_dsp.py) -- what the reference's loop would allow, not what it does
    try:
        nthr, quota_note = usable_cores()
        rx3 = ora.MultiChannelRx(N, M, cp, taper, count_only=True)
        t0 = time.perf_counter()
        rx3.execute_parallel(base, nthr)
        dt2 = time.perf_counter() - t0
        out["all_cores"] = {"value": round(len(base) / dt2 / 1e6, 3), "unit": "Msamples/s", "cores": nthr, "seconds": round(dt2, 2),
                            "frames_decoded": rx3.counts()[2],
                            "note": "OpenMP over time blocks (channelizer) and channels (synchronizers); " + quota_note}
    except Exception as e:                                  # the single-thread figure above is the contract
        out["all_cores"] = {"error": repr(e)}
    try:
        # ---- parity on the benchmark's own slab (untimed): the oracle keeping its frames vs a fresh GPU receiver
        # (cold start, like the oracle)
        rx2 = rx                                                # (the timed pass above kept its frames)
        g = prod.multichannelrx(N, M, cp, taper, **cfg)
        g.Execute(d_slab); g.Flush()
        key = lambda f: (f.channel, f.header)
        of = {key(f): f for f in rx2.frames}
        worst, worst_e, bad, cmpd = 0.0, 0.0, 0, 0
        for f in g.frames:
            o = of.get(key(f))
            if o is None or o.payload != f.payload or int(o.payload_valid) != int(f.payload_valid) or len(o.framesyms) != len(f.framesyms):
                bad += 1
                continue
            cmpd += 1
            d, mag = np.abs(f.framesyms - o.framesyms), np.abs(o.framesyms)
            worst = max(worst, float(np.max(d) / np.max(mag)))
            worst_e = max(worst_e, float(np.max(d / np.maximum(mag, 1e-3 * np.max(mag)))))
        out["gpu_vs_oracle_on_this_slab"] = {"gpu_frames": len(g.frames), "oracle_frames": len(rx2.frames), "compared": cmpd,
                                            "mismatched_or_missing": bad + abs(len(g.frames) - len(rx2.frames)),
                                            "framesyms_max_rel_err": worst, "tolerance": 1e-5,
                                            "framesyms_element_wise_rel_err": worst_e,
                                            "note": "max_rel_err = max|gpu-oracle| / max|oracle| per frame (the 1e-5 bar); element-wise = "
                                                    "max over symbols of |gpu-oracle| / |oracle| (symbols above 1e-3 of full scale)",
                                            "ok": bad == 0 and len(g.frames) == len(rx2.frames) and worst <= 1e-5}
        g.close()
    except Exception as e:
        out["gpu_vs_oracle_on_this_slab"] = {"error": repr(e), "ok": False}
    return out


if __name__ == "__main__":
    main()
