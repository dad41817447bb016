"""The C-ABI multi-GPU pipeline (mcrx_hip_pipeline_*, csrc/pipeline.hip) at world = 1 -- the same code a multi-rank job runs,
minus the RCCL exchange (one GPU per lease: ncclSend / ncclRecv between ranks has never run on hardware, and the docs say so) --
against the oracle and against sharding.Pipeline, the Python mirror the gloo tests drive."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_c_pipeline_world1_matches_oracle_and_python_mirror(oracle, product):
    import torch
    from liquid_usrp_amd import sharding
    from test_gpu_parity import check_frames
    N, M, cp = 64, 64, 8
    K = 2 * N
    tx = product.multichanneltx(N, M, cp, 4)
    iq, sent = tx.generate(24, 100, seed=11)
    tx.close()
    Tc = 8192                                                       # blocks per round (~5 frames): a frame straddles every round boundary
    rounds = int(iq.numel()) // K // Tc
    x = iq[:rounds * Tc * K]
    ora = oracle.MultiChannelRx(N, M, cp, 4)
    ora.execute(x.cpu().numpy())
    results = []
    for which in ("c", "python"):
        rx = product.multichannelrx(N, M, cp, 4, max_payload_len=300, defer_samples=16384)
        pipe = product.pipeline(rx, 0, 1, Tc) if which == "c" else sharding.Pipeline(rx, 0, 1, None, N, Tc, rx.hist_tiles, device=x.device)
        for c in range(rounds):
            sub = x[c * Tc * K:(c + 1) * Tc * K]
            pipe.push(sub, None if c == 0 else x[(c * Tc - 13) * K:c * Tc * K])
            if c % 2:
                rx.Poll()
        if which == "c":
            pipe.wait()
        torch.cuda.synchronize()
        rx.Flush()
        frames = list(rx.frames)
        # frames that end inside the stream's last, unfinished symbol may be pending: compare what the oracle completed too
        check_frames(frames, [f for f in ora.frames][:len(frames)] if len(frames) < len(ora.frames) else ora.frames)
        results.append([(f.channel, f.header, f.payload) for f in frames])
        if which == "c":
            assert pipe.bytes_sent_per_round() == 0 and pipe.exchange_ms()[1] == 0
            pipe.close()
        rx.close()
    assert results[0] == results[1] and len(results[0]) >= 20 * N


def test_c_pipeline_with_the_oversampled_front_end(oracle, product):
    """The multi-GPU code path (time-sharded bank, per-destination groups, deferred frames at round boundaries) with cfg.front_end = 1:
    the folded firpfbch2 + half-band bank needs 27 blocks of history in front of every sub-slab instead of 13
    (mcrx_hip_history_blocks); frames against the oracle's stage-by-stage chain."""
    import torch
    from test_gpu_parity import check_frames
    N, M, cp = 32, 64, 8
    K = 2 * N
    tx = product.multichanneltx(N, M, cp, 4)
    iq, sent = tx.generate(12, 100, seed=12)
    tx.close()
    Tc = 4096
    rounds = int(iq.numel()) // K // Tc
    x = iq[:rounds * Tc * K]
    ora = oracle.MultiChannelRx(N, M, cp, 4, front_end=1)
    ora.execute(x.cpu().numpy())
    rx = product.multichannelrx(N, M, cp, 4, max_payload_len=300, defer_samples=16384, front_end=1)
    H = rx.history_blocks()
    assert H == 27
    pipe = product.pipeline(rx, 0, 1, Tc)
    for c in range(rounds):
        pipe.push(x[c * Tc * K:(c + 1) * Tc * K], None if c == 0 else x[(c * Tc - H) * K:c * Tc * K])
        if c % 2:
            rx.Poll()
    pipe.wait(); torch.cuda.synchronize(); rx.Flush()
    frames = list(rx.frames)
    assert len(frames) >= 10 * N
    check_frames(frames, ora.frames[:len(frames)] if len(frames) < len(ora.frames) else ora.frames)
    pipe.close(); rx.close()


def test_c_pipeline_argument_errors(product):
    rx = product.multichannelrx(8, 64, 8, 4)
    with pytest.raises(product.McrxError):
        product.pipeline(rx, 0, 3, 64)                  # 3 ranks do not divide 8 channels
    with pytest.raises(product.McrxError):
        product.pipeline(rx, 0, 2, 64)                  # world > 1 without rank 0's ncclUniqueId
    with pytest.raises(product.McrxError):
        product.pipeline(rx, 0, 1, 60)                  # not whole tiles
    rx.close()
