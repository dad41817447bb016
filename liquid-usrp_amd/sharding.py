"""Multi-GPU sharding of the multichannel receiver (one process per GPU, RCCL over xGMI).

The polyphase analysis bank couples all channels inside one K-point FFT but is independent
across time blocks (13 blocks of FIR halo); the synchronizers are independent per channel but
serial in time.  So rank r channelizes time slab r, and one all-to-all turns the time-sharded
channelizer output into channel shards:

    channelize:  out[g][tile][c][8]   g = destination rank, channel = g*Cg + c   (rank r, slab r)
    all-to-all:  chunk g of rank r  ->  chunk r of rank g
    sync:        chan[s][tile][c][8]  s = source rank = time slab  ==  [global tile][c][8]

No collective other than this exchange is on the data path.
The backend object supplies the two compute stages (the HIP library in production; tests
inject a CPU stand-in so the orchestration runs under gloo without a GPU).
"""
import numpy as np

TILE = 8


def shard_of(rank, world, num_channels):
    """(first channel, channel count) synchronized by `rank`."""
    assert num_channels % world == 0, "channels must divide evenly across ranks"
    cg = num_channels // world
    return rank * cg, cg


def slab_first_sample(rank, slab_blocks, num_channels):
    """Absolute wideband sample index of the first sample of rank's time slab (NCO phase)."""
    return rank * slab_blocks * 2 * num_channels


def exchange(out, recv, world, dist):
    """Time shards -> channel shards.  `out`/`recv` are flat tensors of world equal chunks."""
    if world == 1:
        return out
    if out.is_complex():
        # RCCL (like NCCL) has no complex element type: exchange the same bytes as float pairs
        import torch
        dist.all_to_all_single(torch.view_as_real(recv).reshape(-1), torch.view_as_real(out).reshape(-1))
    else:
        dist.all_to_all_single(recv, out)
    return recv


def step(backend, iq, slab_blocks, rank, world, dist, out, recv, halo=None, stream=None):
    """One pass over this rank's slab: restart, channelize, exchange, synchronize."""
    backend.restart(stream)
    first = slab_first_sample(rank, slab_blocks, backend.N)
    backend.channelize(iq, slab_blocks, first, out, groups=world, d_halo=halo, stream=stream)
    chan = exchange(out, recv, world, dist)
    backend.sync(chan, 0, world * slab_blocks, stream=stream)
    return chan


def pack_groups(blocks, world):
    """[block][channel] -> the channelizer's grouped tile layout [g][tile][c][8] (numpy helper)."""
    nb, n = blocks.shape
    cg = n // world
    a = blocks.reshape(nb // TILE, TILE, world, cg)         # [tile][t][g][c]
    return np.ascontiguousarray(a.transpose(2, 0, 3, 1))    # [g][tile][c][t]


def unpack_shard(chan, world, cg):
    """[s][tile][c][8] as received -> [c][time] for the rank's channel shard (numpy helper)."""
    a = np.asarray(chan).reshape(-1, cg, TILE)              # [global tile][c][t]
    return np.ascontiguousarray(a.transpose(1, 0, 2)).reshape(cg, -1)
