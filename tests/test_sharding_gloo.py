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
        streams = np.ascontiguousarray(chan.numpy().view(np.complex64).reshape(-1, cg, 8).transpose(1, 0, 2)).reshape(cg, -1)
        assert streams.shape == (cg, nsamples)
        assert first_sample + self.hist_tiles * 8 == self.consumed            # new samples continue the stream
        if self.fs is None:
            self.fs = [self.O.FlexFrameSync(self.M, self.cp, self.taper) for _ in range(cg)]
        if self.consumed:                                                     # the history really is the previous tail
            assert np.array_equal(streams[:, :self.hist_tiles * 8], self.tail)
        new = streams[:, self.hist_tiles * 8:]
        for c in range(cg):
            n0 = len(self.fs[c].frames)
            self.fs[c].execute(new[c])
            for f in self.fs[c].frames[n0:]:
                f.channel = c0 + c
                self.frames.append(f)
        self.tail = new[:, -self.hist_tiles * 8:].copy()
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
    nb = len(iq) // K // (8 * world) * (8 * world)
    iq = iq[:nb * K]
    T = nb // world
    mine = torch.from_numpy(iq[rank * T * K:(rank + 1) * T * K].copy().view(np.float32))
    cg = N // world
    out = torch.zeros(world * (T // 8) * cg * 8 * 2, dtype=torch.float32)
    recv = torch.zeros_like(out)
    be = OracleBackend(O, N, M, cp, tp, iq, rank, world)
    sharding.step(be, mine, T, rank, world, dist, out, recv)
    res = sorted((f.channel, f.header, f.payload, int(f.payload_valid)) for f in be.frames)
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def _pipeline_worker(rank, world, port, q, rounds):
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
    unit = 8 * world * rounds
    nb = len(iq) // K // unit * unit
    iq = iq[:nb * K]
    Tc = nb // (world * rounds)
    be = StreamingOracleBackend(O, N, M, cp, tp, iq, rank, world)
    pipe = sharding.Pipeline(be, rank, world, dist, N, Tc, be.hist_tiles, device=None)
    for c in range(rounds):
        u = c * world + rank
        sub = torch.from_numpy(iq[u * Tc * K:(u + 1) * Tc * K].copy())
        assert pipe.first_sample() == u * Tc * K
        pipe.push(sub, None)
    res = sorted((f.channel, f.header, f.payload, int(f.payload_valid)) for f in be.frames)
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,rounds", [(2, 3), (4, 2)])
def test_round_robin_pipeline_reproduces_single_process_result(oracle, world, rounds):
    """sharding.Pipeline (what bench.py --gpus N runs): sub-slabs round robin over the ranks, one all-to-all per
    round, synchronizers continuing from round to round with their history tiles in front -- same frames as one
    process over the whole stream, each rank delivering exactly its channel shard."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 27000 + (os.getpid() % 2000) + 10 * world
    procs = [ctx.Process(target=_pipeline_worker, args=(r, world, port, q, rounds)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    N, M, cp, tp = 4, 64, 8, 4
    iq, sent = oracle.synth_traffic(N, M, cp, tp, 4, payload_len=60, seed=99)
    unit = 8 * world * rounds
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
    nb = len(iq) // (2 * N) // 16 * 16
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
        assert g.shape == (world, 4, 8 // world, 8)
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
