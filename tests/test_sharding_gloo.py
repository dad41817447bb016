"""world_size-2 test of the multi-GPU orchestration on CPU (gloo): time-sharded channelizer
output -> all-to-all -> channel-sharded synchronizers.  The compute stages are played by the
CPU oracle; what is under test is the layout, the exchange and the shard bookkeeping that
bench.py --gpus N runs over RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleBackend(object):
    """CPU stand-in with the product's stage-level interface (channelize / sync / restart)."""

    def __init__(self, oracle, N, M, cp, taper, full_iq, rank, world):
        self.O, self.N, self.M, self.cp, self.taper = oracle, N, M, cp, taper
        self.full_iq, self.rank, self.world = full_iq, rank, world
        self.frames = []

    def restart(self, stream=None):
        self.frames = []

    def channelize(self, iq, nblocks, first_sample, out, groups=1, d_halo=None, stream=None):
        from liquid_usrp_amd import sharding
        K = 2 * self.N
        # the stand-in recomputes the stream head so that NCO phase and FIR history are right
        upto = first_sample + nblocks * K
        assert np.array_equal(self.full_iq[first_sample:upto], iq.numpy().view(np.complex64))
        ch = self.O.MultiChannelRx(self.N, self.M, self.cp, self.taper).channelize(self.full_iq[:upto])
        mine = ch[first_sample // K:]
        out.copy_(torch.from_numpy(sharding.pack_groups(mine, groups).reshape(-1).view(np.float32)))

    def sync(self, chan, first_sample, nsamples, stream=None):
        from liquid_usrp_amd import sharding
        c0, cg = sharding.shard_of(self.rank, self.world, self.N)
        streams = sharding.unpack_shard(chan.numpy().view(np.complex64), self.world, cg)
        assert streams.shape == (cg, nsamples)
        for c in range(cg):
            fs = self.O.FlexFrameSync(self.M, self.cp, self.taper)
            fs.execute(streams[c])
            for f in fs.frames:
                f.channel = c0 + c
            self.frames += fs.frames


class StreamingOracleBackend(OracleBackend):
    """... for sharding.Pipeline: the synchronizers persist across rounds, every sync call carries `hist_tiles`
    tiles of already-consumed samples in front (first_sample < 0 in the first round)."""
    hist_tiles = 11

    def __init__(self, *a):
        OracleBackend.__init__(self, *a)
        self.fs, self.consumed, self.launch = None, 0, 0

    def channelize(self, iq, nblocks, first_sample, out, groups=1, d_halo=None, stream=None):
        from liquid_usrp_amd import sharding
        K = 2 * self.N
        upto = first_sample + nblocks * K
        assert np.array_equal(self.full_iq[first_sample:upto], iq.numpy())
        ch = self.O.MultiChannelRx(self.N, self.M, self.cp, self.taper).channelize(self.full_iq[:upto])
        out.copy_(torch.from_numpy(sharding.pack_groups(ch[first_sample // K:], groups).reshape(-1)))

    def sync(self, chan, first_sample, nsamples, stream=None):
        from liquid_usrp_amd import sharding
        c0, cg = sharding.shard_of(self.rank, self.world, self.N)
        streams = np.ascontiguousarray(chan.numpy().view(np.complex64).reshape(-1, cg, sharding.TILE).transpose(1, 0, 2)).reshape(cg, -1)
        assert streams.shape == (cg, nsamples)
        TS = sharding.TILE
        assert first_sample + self.hist_tiles * TS == self.consumed            # new samples continue the stream
        if self.fs is None:
            self.fs = [self.O.FlexFrameSync(self.M, self.cp, self.taper) for _ in range(cg)]
        if self.consumed:                                                     # the history really is the previous tail
            assert np.array_equal(streams[:, :self.hist_tiles * TS], self.tail)
        new = streams[:, self.hist_tiles * TS:]
        for c in range(cg):
            n0 = len(self.fs[c].frames)
            self.fs[c].execute(new[c])
            for f in self.fs[c].frames[n0:]:
                f.channel = c0 + c
                self.frames.append(f)
        self.tail = new[:, -self.hist_tiles * TS:].copy()
        self.consumed += new.shape[1]
        self.launch += 1
        return self.launch - 1

    def stream_wait(self, stream=None, launch=None):
        pass


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from __graft_entry__ import load_product
    load_product()
    import oracle as O
    from liquid_usrp_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, M, cp, tp = 4, 64, 8, 4
    K = 2 * N
    iq, sent = O.synth_traffic(N, M, cp, tp, 4, payload_len=60, seed=99)
    nb = len(iq) // K // (sharding.TILE * world) * (sharding.TILE * world)
    iq = iq[:nb * K]
    T = nb // world
    mine = torch.from_numpy(iq[rank * T * K:(rank + 1) * T * K].copy().view(np.float32))
    cg = N // world
    out = torch.zeros(world * T * cg * 2, dtype=torch.float32)
    recv = torch.zeros_like(out)
    be = OracleBackend(O, N, M, cp, tp, iq, rank, world)
    sharding.step(be, mine, T, rank, world, dist, out, recv)
    res = sorted((f.channel, f.header, f.payload, int(f.payload_valid)) for f in be.frames)
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def _pipeline_worker(rank, world, port, q, rounds, nbuf=3):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from __graft_entry__ import load_product
    load_product()
    import oracle as O
    from liquid_usrp_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, M, cp, tp = 4, 64, 8, 4
    K = 2 * N
    iq, sent = O.synth_traffic(N, M, cp, tp, 4, payload_len=60, seed=99)
    unit = 16 * world * rounds
    nb = len(iq) // K // unit * unit
    iq = iq[:nb * K]
    Tc = nb // (world * rounds)
    be = StreamingOracleBackend(O, N, M, cp, tp, iq, rank, world)
    pipe = sharding.Pipeline(be, rank, world, dist, N, Tc, be.hist_tiles, device=None, nbuf=nbuf)
    for c in range(rounds):
        u = c * world + rank
        sub = torch.from_numpy(iq[u * Tc * K:(u + 1) * Tc * K].copy())
        assert pipe.first_sample() == u * Tc * K
        pipe.push(sub, None)
    res = sorted((f.channel, f.header, f.payload, int(f.payload_valid)) for f in be.frames)
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


# (nbuf: rotating buffer sets -- 2 wraps inside three rounds, 5 is what bench.py runs)
@pytest.mark.parametrize("world,rounds,nbuf", [(2, 3, 3), (4, 2, 3), (2, 3, 2), (2, 3, 5)])
def test_round_robin_pipeline_reproduces_single_process_result(oracle, world, rounds, nbuf):
    """sharding.Pipeline (what bench.py --gpus N runs): sub-slabs round robin over the ranks, one all-to-all per
    round, synchronizers continuing from round to round with their history tiles in front -- same frames as one
    process over the whole stream, each rank delivering exactly its channel shard."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 27000 + (os.getpid() % 2000) + 10 * world + nbuf
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, q, rounds, nbuf)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    N, M, cp, tp = 4, 64, 8, 4
    iq, sent = oracle.synth_traffic(N, M, cp, tp, 4, payload_len=60, seed=99)
    unit = 16 * world * rounds
    nb = len(iq) // (2 * N) // unit * unit
    ref = oracle.MultiChannelRx(N, M, cp, tp)
    ref.execute(iq[:nb * 2 * N])
    want = sorted((f.channel, f.header, f.payload, int(f.payload_valid)) for f in ref.frames)
    assert len(want) >= 3 * N
    assert sorted(sum((got[r] for r in range(world)), [])) == want
    cg = N // world
    for r in range(world):
        assert {c for c, *_ in got[r]} == set(range(r * cg, (r + 1) * cg))


def test_two_rank_exchange_reproduces_single_process_result(oracle):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference over the same stream
    N, M, cp, tp = 4, 64, 8, 4
    iq, sent = oracle.synth_traffic(N, M, cp, tp, 4, payload_len=60, seed=99)
    nb = len(iq) // (2 * N) // 32 * 32
    ref = oracle.MultiChannelRx(N, M, cp, tp)
    ref.execute(iq[:nb * 2 * N])
    want = sorted((f.channel, f.header, f.payload, int(f.payload_valid)) for f in ref.frames)
    assert len(want) >= 3 * N
    merged = sorted(got[0] + got[1])
    assert merged == want
    assert {c for c, *_ in got[0]} == {0, 1} and {c for c, *_ in got[1]} == {2, 3}


def test_layout_helpers_roundtrip(product):
    from liquid_usrp_amd import sharding
    rng = np.random.RandomState(0)
    blocks = (rng.randn(32, 8) + 1j * rng.randn(32, 8)).astype(np.complex64)
    for world in (1, 2, 4):
        g = sharding.pack_groups(blocks, world)
        assert g.shape == (world, 32 // sharding.TILE, 8 // world, sharding.TILE)
        for r in range(world):
            c0, cg = sharding.shard_of(r, world, 8)
            # a rank that received only its own chunk from a single slab sees its channels in time order
            got = sharding.unpack_shard(g[r], 1, cg)
            assert np.array_equal(got, blocks[:, c0:c0 + cg].T)
    assert sharding.slab_first_sample(3, 1000, 512) == 3 * 1000 * 1024


def _complex_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_product
    load_product()
    from liquid_usrp_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = (torch.arange(8, dtype=torch.float32) + 100 * rank).to(torch.complex64) * (1 + 2j)
    recv = torch.zeros_like(out)
    got = sharding.exchange(out, recv, world, dist)
    q.put((rank, got.numpy().copy()))
    dist.destroy_process_group()


def test_exchange_moves_complex_tensors_as_float_pairs():
    """bench.py hands complex64 granules to the exchange; RCCL has no complex type, so they travel as floats."""
    world, port = 2, 29000 + os.getpid() % 1000 + 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_complex_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps: p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in ps: p.join(60)
    base = lambda r: (np.arange(8, dtype=np.float32) + 100 * r).astype(np.complex64) * (1 + 2j)
    for r in range(world):
        want = np.concatenate([base(s)[r * 4:(r + 1) * 4] for s in range(world)])
        assert np.array_equal(res[r], want)


# ---- transmit side: channel-sharded frame generators -> all-to-all -> time-sharded synthesis (sharding.TxPipeline)
class OracleTraffic(object):
    """CPU stand-in for the product's TxTraffic: the frames of a channel shard (traffic recipe of
    oracle.synth_traffic, one frame generator per channel, frames back to back) as channel-rate streams."""

    def __init__(self, O, N, M, cp, tp, ch_first, ch_count, nframes, plen, seed):
        self.sent, streams = [], []
        for ch in range(ch_first, ch_first + ch_count):
            rng = np.random.RandomState((seed + ch) & 0x7FFFFFFF)
            fg = O.FlexFrameGen(M, cp, tp)
            s, sent = [], []
            for pid in range(nframes):
                hdr = bytes([(pid >> 8) & 0xff, pid & 0xff, ch & 0xff]) + bytes(rng.randint(0, 256, 5).astype(np.uint8))
                pl = bytes(rng.randint(0, 256, plen).astype(np.uint8))
                s.append(fg.frame(hdr, pl)); sent.append((hdr, pl))
            streams.append(np.concatenate(s)); self.sent.append(sent)
        self.streams = np.stack(streams)                    # [c][block]

    def tiles(self, first_block, nblocks, out, stream=None):
        cg, T = self.streams.shape
        x = np.zeros((cg, nblocks), np.complex64)
        lo, hi = max(first_block, 0), min(first_block + nblocks, T)
        if hi > lo:
            x[:, lo - first_block:hi - first_block] = self.streams[:, lo:hi]
        out.copy_(torch.from_numpy(np.ascontiguousarray(x.reshape(cg, nblocks // 8, 8).transpose(1, 0, 2)).reshape(-1)))


class OracleSynth(object):
    """CPU stand-in for multichanneltx.synthesize: the oracle's synthesis bank started `lead` blocks early from a
    zero state (its memory is 25 blocks), oscillator phase set from the absolute sample index."""

    def __init__(self, O, N):
        self.O, self.N, self.K = O, N, 2 * N
        f = np.float32(-0.5) * np.float32(N - 1) / np.float32(N)
        p = float(np.float32(float(f) * np.pi)) / (2 * np.pi)
        self.dtheta = int(np.rint((p - np.floor(p)) * 2.0 ** 32)) & 0xFFFFFFFF

    def synthesize(self, tiles, groups, first_block, nblocks, lead, keep, gain=None, out=None, stream=None):
        K, N, cg = self.K, self.N, self.N // groups
        tot = lead + nblocks
        t = tiles.numpy().reshape(groups, tot // 8, cg, 8)                       # [g][tile][c][t]
        X = np.zeros((tot, K), np.complex64)
        X[:, :N] = t.transpose(1, 3, 0, 2).reshape(tot, N)
        y = self.O.Channelizer(self.O.SYNTHESIZER, K, 13).synthesize(X).reshape(-1)
        n = (np.arange(tot * K, dtype=np.uint64) + np.uint64((first_block - lead) * K % (1 << 32))) * np.uint64(self.dtheta)
        th = (n & np.uint64(0xFFFFFFFF)).astype(np.float64) * (2 * np.pi / 2.0 ** 32)
        y = (y.astype(np.complex128) * np.exp(1j * th)).astype(np.complex64) * np.float32(1.0 / N if gain is None else gain)
        out.copy_(torch.from_numpy(y[(lead - keep) * K:]))
        return out


def _tx_worker(rank, world, port, q, rounds, Tc):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from __graft_entry__ import load_product
    load_product()
    import oracle as O
    from liquid_usrp_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, M, cp, tp = 4, 64, 8, 4
    c0, cg = sharding.shard_of(rank, world, N)
    tr = OracleTraffic(O, N, M, cp, tp, c0, cg, 2, 60, 99)
    pipe = sharding.TxPipeline(OracleSynth(O, N), tr, rank, world, dist, N, Tc, lead_blocks=48, keep_blocks=16, device=None)
    slabs = []
    for c in range(rounds):
        iq, _ = pipe.push()
        slabs.append(iq.numpy().copy())
    q.put((rank, slabs, tr.sent))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_transmitter_reproduces_the_single_process_stream(oracle, world):
    """sharding.TxPipeline under gloo: every rank modulates its channel shard, one all-to-all per round turns channel
    shards into time shards, every rank synthesizes its sub-slab (+16 blocks in front = the receiver's halo).
    The sub-slabs, put back in round-robin order, are the oracle multichanneltx's stream over all channels."""
    N, M, cp, tp, Tc = 4, 64, 8, 4, 128
    K = 2 * N
    iq, sent = oracle.synth_traffic(N, M, cp, tp, 2, payload_len=60, seed=99)
    rounds = (len(iq) // K + world * Tc - 1) // (world * Tc)
    ref = np.concatenate([iq, np.zeros(rounds * world * Tc * K - len(iq), np.complex64)])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 25000 + (os.getpid() % 2000) + 10 * world
    procs = [ctx.Process(target=_tx_worker, args=(r, world, port, q, rounds, Tc)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r, slabs, s = q.get(timeout=180)
        got[r] = (slabs, s)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cg = N // world
    scale = float(np.max(np.abs(ref)))
    worst = 0.0
    for r in range(world):
        slabs, s = got[r]
        assert s == sent[r * cg:(r + 1) * cg]
        for c, y in enumerate(slabs):
            u = c * world + r
            want = ref[(u * Tc - 16) * K:(u + 1) * Tc * K] if u else np.concatenate([np.zeros(16 * K, np.complex64), ref[:Tc * K]])
            worst = max(worst, float(np.max(np.abs(y - want))) / scale)
    assert worst <= 1e-5, worst
