"""GPU tests of the sharded transmit side (BASELINE configs[4]/[5], src/multichannel_txrx.cc over several GPUs):
channel-sharded frame generators -> all-to-all -> time-sharded synthesis bank, all ranks' handles in one process
and the all-to-all played by slicing.  The sharded stream must be the unsharded generator's, bit for bit; fed
straight into the round-robin sharded receiver it must give back every frame that was sent."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _sharded_tx_rounds(product, torch, N, M, cp, world, Tc, rounds, nf, plen, seed, lead=48, keep=16, **kw):
    """Yields (round, rank, iq) with iq = blocks [u*Tc - keep, (u+1)*Tc) of the stream, u = round*world + rank,
    produced by the stage calls sharding.TxPipeline makes; also returns what was sent."""
    from liquid_usrp_amd import sharding
    cg = N // world
    txs, trs = [], []
    sent = [None] * N
    for r in range(world):
        tx = product.multichanneltx(N, M, cp, 4)
        c0, cnt = sharding.shard_of(r, world, N)
        tr = tx.traffic(c0, cnt, nf, plen, seed=seed, **kw)
        for c in range(cnt):
            sent[c0 + c] = tr.sent[c]
        txs.append(tx); trs.append(tr)
    per = (lead + Tc) * cg
    slabs = []
    for c in range(rounds):
        outs = []
        for r in range(world):
            o = torch.empty(world * per, dtype=torch.complex64, device="cuda")
            for g in range(world):
                trs[r].tiles((c * world + g) * Tc - lead, lead + Tc, o[g * per:(g + 1) * per])
            outs.append(o)
        torch.cuda.synchronize()
        for r in range(world):
            recv = torch.cat([outs[s][r * per:(r + 1) * per] for s in range(world)])          # all_to_all_single
            iq = txs[r].synthesize(recv, world, (c * world + r) * Tc, Tc, lead, keep)
            slabs.append((c, r, iq))
        torch.cuda.synchronize()
    for tr in trs:
        tr.close()
    for tx in txs:
        tx.close()
    return slabs, sent


@pytest.mark.parametrize("N,M,cp,world,Tc", [(64, 64, 8, 4, 256), (16, 256, 32, 2, 512), (256, 64, 8, 8, 128)])
def test_sharded_transmitter_is_the_unsharded_stream_bit_for_bit(product, N, M, cp, world, Tc):
    torch = _torch()
    nf, plen, seed = 2, 120, 4242
    tx = product.multichanneltx(N, M, cp, 4)
    need = int(product.lib().mctx_hip_blocks_for(tx._h, nf, plen, 40, 1, 6))
    rounds = (need + world * Tc - 1) // (world * Tc)
    full, sent_full = tx.generate(nf, plen, seed=seed, nblocks=rounds * world * Tc)
    tx.close()
    K, keep = 2 * N, 16
    slabs, sent = _sharded_tx_rounds(product, torch, N, M, cp, world, Tc, rounds, nf, plen, seed, keep=keep)
    assert sent == sent_full
    ref = full[:rounds * world * Tc * K]
    assert float(ref.abs().max()) > 0
    for c, r, iq in slabs:
        u = c * world + r
        assert torch.equal(iq[keep * K:], ref[u * Tc * K:(u + 1) * Tc * K]), (c, r)
        if u > 0:
            assert torch.equal(iq[:keep * K], ref[(u * Tc - keep) * K:u * Tc * K]), (c, r)      # the receiver's halo
        else:
            assert float(iq[:keep * K].abs().max()) == 0.0


def test_config5_full_duplex_eight_rank_emulation(product):
    """BASELINE configs[5] as an 8-GPU job, all ranks in one process: 256 channels, M=64, QPSK + Hamming(12,8).
    Sharded transmitter -> (each rank keeps the slab it synthesized) -> round-robin sharded receiver; every frame
    that was sent comes back from the rank that owns its channel, payload intact."""
    torch = _torch()
    from liquid_usrp_amd import sharding
    N, M, cp, world, nf, plen = 256, 64, 8, 8, 3, 600
    K, cg, keep = 2 * N, N // world, 16
    tx = product.multichanneltx(N, M, cp, 4)
    need = int(product.lib().mctx_hip_blocks_for(tx._h, nf, plen, 40, 1, 6))
    tx.close()
    Tc = 512
    rounds = (need + world * Tc - 1) // (world * Tc)
    slabs, sent = _sharded_tx_rounds(product, torch, N, M, cp, world, Tc, rounds, nf, plen, 99, keep=keep)
    rxs = []
    for r in range(world):
        c0, cnt = sharding.shard_of(r, world, N)
        rxs.append(product.multichannelrx(N, M, cp, 4, max_payload_len=plen, channel_first=c0, channel_count=cnt))
    H = rxs[0].hist_tiles
    TS = product.TILE
    tiles = Tc // TS
    per = tiles * cg * TS
    prev = [None] * world
    it = iter(slabs)
    for c in range(rounds):
        outs = []
        for r in range(world):
            cc, rr, iq = next(it)
            assert (cc, rr) == (c, r)
            u = c * world + r
            o = torch.empty(world * per, dtype=torch.complex64, device="cuda")
            halo = iq[(keep - 13) * K:keep * K] if u > 0 else None
            rxs[r].channelize(iq[keep * K:], Tc, u * Tc * K, o, groups=world, d_halo=halo)
            outs.append(o)
        torch.cuda.synchronize()
        for r in range(world):
            new = torch.cat([outs[s][r * per:(r + 1) * per] for s in range(world)])
            hist = prev[r][-H * cg * TS:] if prev[r] is not None else torch.zeros(H * cg * TS, dtype=torch.complex64, device="cuda")
            buf = torch.cat([hist, new])
            rxs[r].sync(buf, c * world * Tc - H * TS, H * TS + world * Tc)
            prev[r] = buf
        torch.cuda.synchronize()
    got = []
    for r in range(world):
        rxs[r].Flush()
        c0, cnt = sharding.shard_of(r, world, N)
        assert all(c0 <= f.channel < c0 + cnt for f in rxs[r].frames)
        got += rxs[r].frames
        rxs[r].close()
    assert len(got) == nf * N
    for f in got:
        assert f.header_valid and f.payload_valid
        assert sent[f.channel][(f.header[0] << 8) | f.header[1]] == (f.header, f.payload)


def test_tx_and_rx_pipelines_back_to_back_on_one_gpu(product):
    """The production objects themselves (world = 1: the exchange is a copy): TxPipeline.push feeds Pipeline.push
    round after round with events only; frames = what the traffic object says was sent."""
    torch = _torch()
    from liquid_usrp_amd import sharding
    N, M, cp, nf, plen, Tc = 32, 64, 8, 4, 300, 1024
    dev = torch.device("cuda", 0)
    tx = product.multichanneltx(N, M, cp, 4)
    tr = tx.traffic(0, N, nf, plen, seed=7)
    rounds = (tr.blocks + Tc - 1) // Tc + 1
    rx = product.multichannelrx(N, M, cp, 4, max_payload_len=plen, defer_samples=4096)
    txp = sharding.TxPipeline(tx, tr, 0, 1, None, N, Tc, device=dev)
    rxp = sharding.Pipeline(rx, 0, 1, None, N, Tc, rx.hist_tiles, device=dev)
    K, keep = 2 * N, txp.keep
    for c in range(rounds):
        consumed = rxp.evA[(c - txp.nbuf) % rxp.nbuf] if c >= txp.nbuf else None
        iq, ev = txp.push(consumed=consumed)
        rxp.push(iq[keep * K:], halo=iq[(keep - 13) * K:keep * K] if c > 0 else None, after=ev)
    torch.cuda.synchronize()
    rx.Flush()
    assert len(rx.frames) == nf * N
    for f in rx.frames:
        assert f.payload_valid and tr.sent[f.channel][(f.header[0] << 8) | f.header[1]] == (f.header, f.payload)
    rx.close(); tr.close(); tx.close()
