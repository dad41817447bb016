"""Multi-GPU sharding of the multichannel receiver (one process per GPU, RCCL over xGMI).

The polyphase analysis bank couples all channels inside one K-point FFT but is independent
across time blocks (13 blocks of FIR halo); the synchronizers are independent per channel but
serial in time.  So the stream is cut into sub-slabs of `sub_blocks` blocks that go round robin
to the ranks -- sub-slab u to rank u % G -- and every round of G sub-slabs ends in one all-to-all
that turns the time-sharded channelizer output into channel shards:

    channelize:  out[g][tile][c][16]  g = destination rank, channel = g*Cg + c   (rank r, sub-slab c*G + r)
    all-to-all:  chunk g of rank r  ->  chunk r of rank g
    sync:        chan[s][tile][c][16] s = source rank  ==  [tile of the round][c][16], contiguous in time

Rounds are pipelined with rotating buffers:

    channelize(c+1)  ||  all_to_all(c)  ||  synchronizers(c-1)

on two streams of the pipeline's own -- one for the channelizer, one for the exchange and the launch of the synchronizer stage,
which overlaps its acquisition and payload kernels on the handle's internal streams (a third stream for that launch, as in
rounds 2-3, cost the one-GPU run 11 %: scratch/r4an.sh).
No collective other than this exchange is on the data path.
The backend object supplies the two compute stages (the HIP library in production; tests
inject a CPU stand-in so the orchestration runs under gloo without a GPU).
"""
import numpy as np

TILE = 16         # receive side: channel-rate samples per (channel, tile) granule = MCRX_TILE (one channel per 128-B line)
TX_TILE = 8       # transmit side: granules of the frame generators' output (mctx_hip_traffic_tiles / synthesize_tiles)


def shard_of(rank, world, num_channels):
    """(first channel, channel count) synchronized by `rank`."""
    assert num_channels % world == 0, "channels must divide evenly across ranks"
    cg = num_channels // world
    return rank * cg, cg


def slab_first_sample(rank, slab_blocks, num_channels):
    """Absolute wideband sample index of the first sample of rank's time slab (NCO phase)."""
    return rank * slab_blocks * 2 * num_channels


def exchange(out, recv, world, dist):
    """Time shards -> channel shards.  `out`/`recv` are flat tensors of world equal chunks.

    Device tensors under a process group without device collectives (gloo: the one-GPU rehearsal of the multi-rank job,
    bench.py --rehearse-on-one-gpu, where RCCL refuses two ranks on one device) are staged through host memory -- the same
    chunks to the same ranks, only the transport differs."""
    if world == 1:
        return out
    if getattr(out, "is_cuda", False) and dist.get_backend() == "gloo":
        import torch
        o = torch.view_as_real(out).reshape(-1).cpu() if out.is_complex() else out.cpu()       # (waits for the producing stream)
        r = torch.empty_like(o)
        dist.all_to_all_single(r, o)
        if recv.is_complex():
            torch.view_as_real(recv).reshape(-1).copy_(r, non_blocking=False)
        else:
            recv.copy_(r, non_blocking=False)
        return recv
    if out.is_complex():
        # RCCL (like NCCL) has no complex element type: exchange the same bytes as float pairs
        import torch
        dist.all_to_all_single(torch.view_as_real(recv).reshape(-1), torch.view_as_real(out).reshape(-1))
    else:
        dist.all_to_all_single(recv, out)
    return recv


def step(backend, iq, slab_blocks, rank, world, dist, out, recv, halo=None, stream=None):
    """Unpipelined form, one slab per rank: restart, channelize, exchange, synchronize (each stage after the
    other on `stream`; kept for the stage-level tests -- bench.py uses Pipeline)."""
    backend.restart(stream)
    first = slab_first_sample(rank, slab_blocks, backend.N)
    backend.channelize(iq, slab_blocks, first, out, groups=world, d_halo=halo, stream=stream)
    chan = exchange(out, recv, world, dist)
    backend.sync(chan, 0, world * slab_blocks, stream=stream)
    return chan


class Pipeline(object):
    """Round-robin time sharding with the exchange overlapped (see the module docstring).

    push(iq_sub, halo) runs one round: this rank's sub-slab of the round (`sub_blocks` blocks, the 13 blocks that
    precede it in the stream as `halo`, None = zeros), the all-to-all, and the synchronizers of the rank's channel
    shard over the round's G*sub_blocks blocks.  Buffers rotate over `nbuf` rounds; nothing waits on the host.
    Tensors are complex64 on the backend's device (CUDA: three torch streams; CPU/gloo: everything in order).
    """

    def __init__(self, backend, rank, world, dist, num_channels, sub_blocks, hist_tiles, device=None, nbuf=3, streams=2):
        import torch
        assert sub_blocks % TILE == 0 and num_channels % world == 0
        self.be, self.rank, self.world, self.dist = backend, rank, world, dist
        self.N, self.K, self.Tc, self.hist, self.nbuf = num_channels, 2 * num_channels, sub_blocks, hist_tiles, nbuf
        self.cg = num_channels // world
        self.tiles = sub_blocks // TILE
        self.cuda = device is not None and torch.device(device).type == "cuda"
        per = self.tiles * self.cg * TILE
        self.out = [torch.zeros(world * per, dtype=torch.complex64, device=device) for _ in range(nbuf)]
        self.recv = [torch.zeros((hist_tiles * self.cg * TILE) + world * per, dtype=torch.complex64, device=device)
                     for _ in range(nbuf)]
        self.hist_elems = hist_tiles * self.cg * TILE
        # one rank on a GPU: there is nothing to exchange, the channelizer writes where the synchronizers read (behind the
        # history tiles) -- what the C pipeline does too (csrc/pipeline.hip: "out[i] IS recv[i]")
        self.alias = world == 1 and self.cuda
        if self.alias:
            self.out = [r[self.hist_elems:] for r in self.recv]
        self.rounds = 0
        self.tickets = [None] * nbuf
        self.time_exchange = False                  # bench.py: HIP events around every exchange (exchange_ms)
        self._xev = []
        if self.cuda:
            nstr = int(streams)          # 2: channelizer | exchange + launch of the synchronizer stage (3 = a stream per stage, rounds 2-3: slower)
            self.sA = torch.cuda.Stream(device=device)
            self.sB = torch.cuda.Stream(device=device) if nstr >= 2 else self.sA
            self.sC = torch.cuda.Stream(device=device) if nstr >= 3 else self.sB     # (module docstring: the synchronizer stage is launched from the exchange's stream)
            # the buffers above were zeroed on the current stream; round 0 relies on those zeros (no history copy yet)
            for s in (self.sA, self.sB, self.sC):
                s.wait_stream(torch.cuda.current_stream(device))
            self.evA = [torch.cuda.Event() for _ in range(nbuf)]
            self.evB = [torch.cuda.Event() for _ in range(nbuf)]
            self.evC = [torch.cuda.Event() for _ in range(nbuf)]
        else:
            self.sA = self.sB = self.sC = None

    def first_sample(self, rnd=None):
        rnd = self.rounds if rnd is None else rnd
        return (rnd * self.world + self.rank) * self.Tc * self.K

    def push(self, iq_sub, halo=None, after=None):
        import torch
        c, i, nb = self.rounds, self.rounds % self.nbuf, self.nbuf
        out, recv = self.out[i], self.recv[i]
        new = recv[self.hist_elems:]
        # ---- A: channelize this rank's sub-slab into per-destination groups
        if self.cuda:
            if after is not None:
                self.sA.wait_event(after)                       # whoever produced iq_sub (e.g. TxPipeline.push)
            if c >= nb and not self.alias:
                self.sA.wait_event(self.evB[i])                 # the exchange that last read out[i]
            if c >= nb and self.alias:                          # out[i] is recv[i]: its last readers instead
                self.be.stream_wait(self.sA, launch=self.tickets[i])
                self.sA.wait_event(self.evC[(i + 1) % nb])
            with torch.cuda.stream(self.sA):
                self.be.channelize(iq_sub, self.Tc, self.first_sample(c), out, groups=self.world, d_halo=halo, stream=self.sA)
                self.evA[i].record(self.sA)
        else:
            self.be.channelize(iq_sub, self.Tc, self.first_sample(c), out, groups=self.world, d_halo=halo, stream=None)
        # ---- B: time shards -> channel shards
        if self.cuda:
            self.sB.wait_event(self.evA[i])
            if c >= nb:
                self.be.stream_wait(self.sB, launch=self.tickets[i])   # the synchronizers that last read recv[i]
                self.sB.wait_event(self.evC[(i + 1) % nb])             # ... and the history copy that read its tail
            with torch.cuda.stream(self.sB):
                if self.time_exchange:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(self.sB)
                if self.alias:
                    pass
                elif self.world == 1:
                    new.copy_(out, non_blocking=True)
                else:
                    exchange(out, new, self.world, self.dist)
                if self.time_exchange:
                    e1.record(self.sB)
                    self._xev.append((e0, e1))
                self.evB[i].record(self.sB)
        else:
            if self.world == 1:
                new.copy_(out)
            else:
                exchange(out, new, self.world, self.dist)
        # ---- C: synchronizer history in front (tail of the previous round), then the bank over the round
        first_chan = c * self.world * self.Tc - self.hist * TILE
        nsamp = self.hist * TILE + self.world * self.Tc
        if self.cuda:
            self.sC.wait_event(self.evB[i])
            with torch.cuda.stream(self.sC):
                if c > 0:
                    recv[:self.hist_elems].copy_(self.recv[(c - 1) % nb][-self.hist_elems:], non_blocking=True)
                self.evC[i].record(self.sC)
                self.tickets[i] = self.be.sync(recv, first_chan, nsamp, stream=self.sC)
        else:
            if c > 0:
                recv[:self.hist_elems].copy_(self.recv[(c - 1) % nb][-self.hist_elems:])
            self.tickets[i] = self.be.sync(recv, first_chan, nsamp, stream=None)
        self.rounds += 1


    def exchange_ms(self, reset=True):
        """(total ms, rounds) of the exchanges timed since time_exchange was set: event pairs on the exchange stream, i.e.
        the all-to-all as the device saw it (queueing behind the previous round's exchange excluded, waiting for a slow
        peer included)."""
        tot = 0.0
        for e0, e1 in self._xev:
            e1.synchronize()
            tot += e0.elapsed_time(e1)
        n = len(self._xev)
        if reset:
            self._xev = []
        return tot, n

    def bytes_sent_per_round(self):
        """Bytes this rank's all-to-all sends to OTHER ranks in one round (4 B per wideband sample x (G-1)/G)."""
        return self.tiles * self.cg * TILE * 8 * (self.world - 1)


class TxPipeline(object):
    """The transmit side of the full-duplex loop sharded the other way round (src/multichannel_txrx.cc:105-259 run
    over G GPUs): frame generators are per channel, the synthesis bank couples all channels of one block, so

        frames:      rank r modulates the frames of ITS channel shard once        (tx.traffic)
        tiles:       out[g][tile][c][8] = their channel-rate samples over the blocks rank g will synthesize:
                     sub-slab c*G + g plus `lead` blocks of filter history in front  (traffic.tiles, per destination)
        all-to-all:  chunk g of rank r -> chunk r of rank g : channel shards -> time shards
        synthesize:  recv[s][tile][c][8] -> inverse FFT over all N channels, synthesis FIR, oscillator -> the wideband
                     samples of sub-slab c*G + r, with `keep` (>= 13) blocks in front of it

    push() returns (iq, event): iq[keep*K:] is exactly the sub-slab the receiver's Pipeline.push wants from this rank
    in the same round and iq[(keep-13)*K:keep*K] its halo, so in the full-duplex job no wideband sample ever leaves
    the GPU that made it.  Three streams, `nbuf` rotating buffer sets, like Pipeline.
    """

    def __init__(self, tx, traffic, rank, world, dist, num_channels, sub_blocks, lead_blocks=48, keep_blocks=16,
                 device=None, nbuf=3, gain=None, streams=0):
        import torch
        assert sub_blocks % TX_TILE == 0 and lead_blocks % TX_TILE == 0 and num_channels % world == 0
        assert lead_blocks >= 25 + keep_blocks, "the synthesis filter remembers 25 blocks"
        self.tx, self.traffic, self.rank, self.world, self.dist = tx, traffic, rank, world, dist
        self.N, self.K, self.Tc, self.lead, self.keep, self.nbuf = num_channels, 2 * num_channels, sub_blocks, lead_blocks, keep_blocks, nbuf
        self.cg = num_channels // world
        self.gain = gain
        self.per = (lead_blocks + sub_blocks) * self.cg
        self.cuda = device is not None and torch.device(device).type == "cuda"
        self.out = [torch.zeros(world * self.per, dtype=torch.complex64, device=device) for _ in range(nbuf)]
        self.recv = [torch.zeros(world * self.per, dtype=torch.complex64, device=device) for _ in range(nbuf)]
        self.iq = [torch.zeros((keep_blocks + sub_blocks) * self.K, dtype=torch.complex64, device=device) for _ in range(nbuf)]
        self.rounds = 0
        self.alias = world == 1 and self.cuda                   # one rank: the granules are made where the synthesis reads them
        if self.alias:
            self.out = self.recv
        if self.cuda:
            # (streams are not free: every one past the hardware queues shares a queue with a busy one -- Pipeline above.  Here stage C is
            #  a kernel of its own: several ranks run it beside the exchange, which shares the tile generator's stream; one rank has
            #  no exchange and runs everything in order -- full duplex on one GPU 62.8 / 68.4 / 74.3 Gsample/s with 3 / 2 / 1 streams)
            nstr = int(streams) or (1 if self.alias else 2)
            self.sA = torch.cuda.Stream(device=device)
            self.sB = torch.cuda.Stream(device=device) if nstr >= 3 else self.sA
            self.sC = torch.cuda.Stream(device=device) if nstr >= 2 else self.sA
            for s in (self.sA, self.sB, self.sC):               # the zeroed lead tiles above were written on the current stream
                s.wait_stream(torch.cuda.current_stream(device))
            self.evA = [torch.cuda.Event() for _ in range(nbuf)]
            self.evB = [torch.cuda.Event() for _ in range(nbuf)]
            self.evC = [torch.cuda.Event() for _ in range(nbuf)]

    def push(self, consumed=None):
        """One round.  `consumed`: event after which iq buffer of round c - nbuf is free again (None: caller's problem)."""
        import torch
        c, i, nb = self.rounds, self.rounds % self.nbuf, self.nbuf
        out, recv, iq = self.out[i], self.recv[i], self.iq[i]
        nblk = self.lead + self.Tc

        def stage_a(st):
            for g in range(self.world):
                self.traffic.tiles((c * self.world + g) * self.Tc - self.lead, nblk, out[g * self.per:(g + 1) * self.per], stream=st)

        def stage_c(st):
            self.tx.synthesize(recv, self.world, (c * self.world + self.rank) * self.Tc, self.Tc, self.lead, self.keep,
                               gain=self.gain, out=iq, stream=st)
        if not self.cuda:
            stage_a(None)
            if self.world == 1:
                recv.copy_(out)
            else:
                exchange(out, recv, self.world, self.dist)
            stage_c(None)
            self.rounds += 1
            return iq, None
        if c >= nb:
            self.sA.wait_event(self.evC[i] if self.alias else self.evB[i])   # the exchange that last read out[i] (aliased: the synthesis)
        with torch.cuda.stream(self.sA):
            stage_a(self.sA)
            self.evA[i].record(self.sA)
        self.sB.wait_event(self.evA[i])
        if c >= nb:
            self.sB.wait_event(self.evC[i])                     # the synthesis that last read recv[i]
        with torch.cuda.stream(self.sB):
            if self.alias:
                pass
            elif self.world == 1:
                recv.copy_(out, non_blocking=True)
            else:
                exchange(out, recv, self.world, self.dist)
            self.evB[i].record(self.sB)
        self.sC.wait_event(self.evB[i])
        if consumed is not None:
            self.sC.wait_event(consumed)
        with torch.cuda.stream(self.sC):
            stage_c(self.sC)
            self.evC[i].record(self.sC)
        self.rounds += 1
        return iq, self.evC[i]


def pack_groups(blocks, world):
    """[block][channel] -> the channelizer's grouped tile layout [g][tile][c][TILE] (numpy helper)."""
    nb, n = blocks.shape
    cg = n // world
    a = blocks.reshape(nb // TILE, TILE, world, cg)         # [tile][t][g][c]
    return np.ascontiguousarray(a.transpose(2, 0, 3, 1))    # [g][tile][c][t]


def unpack_shard(chan, world, cg):
    """[s][tile][c][TILE] as received -> [c][time] for the rank's channel shard (numpy helper)."""
    a = np.asarray(chan).reshape(-1, cg, TILE)              # [global tile][c][t]
    return np.ascontiguousarray(a.transpose(1, 0, 2)).reshape(cg, -1)
