"""The reference's unchanged src/multichannel_rx.cc (binary built in the build container from
/root/reference, see liquid-usrp_amd/host/Makefile `refapp`) running on the GPU library with
the synthetic-IQ UHD shim."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "liquid-usrp_amd", "lib", "multichannel_rx_ref")


@pytest.mark.skipif(not os.path.exists(EXE), reason="reference app binary not built")
def test_unchanged_reference_app_decodes_synthetic_traffic(oracle, tmp_path):
    N, M, cp, tp = 4, 64, 8, 4
    iq, sent = oracle.synth_traffic(N, M, cp, tp, 3, payload_len=120)
    f = tmp_path / "iq.bin"
    iq.astype(np.complex64).tofile(f)
    # the class library is plain g++ code over the C-ABI: (re)build it where the test runs
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "liquid-usrp_amd", "host"), "-s"])
    env = dict(os.environ, MCRX_IQ_FILE=str(f), MCRX_IQ_PACKET="4096", MCRX_DEBUG_DIR=str(tmp_path))
    out = subprocess.run([EXE, "-n", str(N), "-M", str(M), "-C", str(cp), "-T", str(tp), "-t", "0.5", "-v"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = re.findall(r"channel: (\d+) rx packet id:\s+(\d+)\n", out.stdout)
    assert len(lines) >= 3 * N, out.stdout[-2000:]
    assert {int(c) for c, _ in lines} == set(range(N))
    # the stand-in for the reference's BST_DEBUG dump (lib/multichannelrx.cc:118-122): one framesync_channel%u.m per channel
    for c in range(N):
        txt = (tmp_path / ("framesync_channel%u.m" % c)).read_text()
        assert "framesyms = [" in txt and "frames received" in txt
    assert "PAYLOAD INVALID" not in out.stdout.split("usrp data transfer started")[1][:2000]


@pytest.mark.skipif(not os.path.exists(EXE), reason="reference app binary not built")
def test_unchanged_reference_app_on_the_oversampled_front_end(oracle, tmp_path):
    """The unchanged src/multichannel_rx.cc with the channelizer BASELINE.json's north_star names in front of the synchronizers:
    MCRX_FRONT_END=1 switches the receiver class to the firpfbch2 bank + half-band decimators (one folded kernel, include/mcrx_hip.h:
    front_end) from outside -- the class interface has no argument for it.  Same packets as with the reference's own bank."""
    N, M, cp, tp = 4, 64, 8, 4
    iq, sent = oracle.synth_traffic(N, M, cp, tp, 4, payload_len=120)
    f = tmp_path / "iq.bin"
    iq.astype(np.complex64).tofile(f)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "liquid-usrp_amd", "host"), "-s"])
    got = {}
    for mode, extra in (("firpfbch", {}), ("firpfbch2", {"MCRX_FRONT_END": "1"})):
        env = dict(os.environ, MCRX_IQ_FILE=str(f), MCRX_IQ_PACKET="4096", **extra)
        out = subprocess.run([EXE, "-n", str(N), "-M", str(M), "-C", str(cp), "-T", str(tp), "-t", "0.5", "-v"],
                             env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr[-2000:]
        assert "INVALID" not in out.stdout.split("usrp data transfer started")[1]
        got[mode] = set(re.findall(r"channel: (\d+) rx packet id:\s+(\d+)\n", out.stdout))
    assert len(got["firpfbch"]) >= 4 * N and got["firpfbch2"] == got["firpfbch"]
    env = dict(os.environ, MCRX_IQ_FILE=str(f), MCRX_FRONT_END="1", MCRX_WORLD="1", MCRX_RANK="0")     # not with the sharded class
    out = subprocess.run([EXE, "-n", str(N), "-M", str(M), "-C", str(cp), "-T", str(tp), "-t", "0.1"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "MCRX_FRONT_END" in out.stderr


@pytest.mark.skipif(not os.path.exists(EXE), reason="reference app binary not built")
def test_unchanged_reference_app_through_the_sharded_receiver(oracle, tmp_path):
    """The same unchanged application with the receiver class switched to its multi-GPU form from outside (MCRX_WORLD / MCRX_RANK /
    MCRX_SUB_BLOCKS: host/multichannelrx.cc over mcrx_hip_pipeline_*), world = 1 -- the one size a one-GPU lease can run: rounds of
    512 blocks fed from host memory, frames cut by round boundaries re-acquired by the next round.  Same packets as the plain receiver."""
    N, M, cp, tp = 4, 64, 8, 4
    iq, sent = oracle.synth_traffic(N, M, cp, tp, 6, payload_len=120)
    f = tmp_path / "iq.bin"
    iq.astype(np.complex64).tofile(f)
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "liquid-usrp_amd", "host"), "-s"])
    got = {}
    for mode, extra in (("plain", {}), ("sharded", {"MCRX_WORLD": "1", "MCRX_RANK": "0", "MCRX_SUB_BLOCKS": "512"})):
        env = dict(os.environ, MCRX_IQ_FILE=str(f), MCRX_IQ_PACKET="4096", **extra)
        out = subprocess.run([EXE, "-n", str(N), "-M", str(M), "-C", str(cp), "-T", str(tp), "-t", "0.5", "-v"],
                             env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr[-2000:]
        assert "INVALID" not in out.stdout.split("usrp data transfer started")[1]
        got[mode] = set(re.findall(r"channel: (\d+) rx packet id:\s+(\d+)\n", out.stdout))
    assert len(got["plain"]) >= 6 * N and got["sharded"] == got["plain"]
    env = dict(os.environ, MCRX_IQ_FILE=str(f), MCRX_WORLD="3", MCRX_RANK="0")          # 3 ranks do not divide 4 channels: the constructor's error path
    out = subprocess.run([EXE, "-n", str(N), "-M", str(M), "-C", str(cp), "-T", str(tp), "-t", "0.1"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "MCRX_WORLD" in out.stderr


def test_sharded_class_reset_in_mid_stream_and_end_of_stream(oracle, tmp_path):
    """Reset() on the sharded class (it used to reset the handle only and leave the pipeline's round counter, halo and
    partial round behind) and the samples of an unfinished round at destruction (they used to be lost).  host/shard_test.cc feeds the
    class bulk pieces of uneven size, calls Reset() in the middle of a frame, and destroys the object right behind the last frame --
    a point that is not a round boundary.  Plain class and sharded class (world = 1, rounds of 256 and of 4096 blocks: the second is
    longer than the whole stream, so every frame is delivered by the reset's / the destructor's completion of the round) print the
    same frames, and they are the frames the oracle receiver gets from the same calls."""
    N, M, cp, tp = 4, 64, 8, 4
    iq, sent = oracle.synth_traffic(N, M, cp, tp, 8, payload_len=150, seed=3)
    iq = iq.astype(np.complex64)
    f = tmp_path / "iq.bin"
    iq.tofile(f)
    host = os.path.join(ROOT, "liquid-usrp_amd", "host")
    subprocess.check_call(["make", "-C", host, "-s"])
    subprocess.check_call(["make", "-C", host, "-s", "shard_test"])
    K = 2 * N
    frame = len(iq) // 8                                      # roughly one frame period of wideband samples
    reset_at = (3 * frame + frame // 2) // K * K + 3          # in the middle of the fourth frame, not on a block boundary
    ora = oracle.MultiChannelRx(N, M, cp, tp)
    ora.execute(iq[:reset_at]); ora.reset(); ora.execute(iq[reset_at:])
    want = sorted((f_.channel, (f_.header[0] << 8) | f_.header[1], f_.payload_valid, len(f_.payload)) for f_ in ora.frames)
    assert len(want) >= 6 * N and all(w[2] for w in want)
    got = {}
    for mode, extra in (("plain", {}), ("sharded256", {"MCRX_WORLD": "1", "MCRX_RANK": "0", "MCRX_SUB_BLOCKS": "256"}),
                        ("sharded4096", {"MCRX_WORLD": "1", "MCRX_RANK": "0", "MCRX_SUB_BLOCKS": "4096"})):
        out = subprocess.run([os.path.join(ROOT, "liquid-usrp_amd", "lib", "shard_test"), str(f), str(N), str(M), str(cp), str(tp), "10007", str(reset_at)],
                             env=dict(os.environ, **extra), capture_output=True, text=True, timeout=180)
        assert out.returncode == 0 and "done" in out.stdout, out.stderr[-2000:]
        got[mode] = sorted((int(a), int(b), int(d), int(e)) for a, b, c, d, e in
                           re.findall(r"frame ch (\d+) pid (\d+) hv (\d+) pv (\d+) len (\d+)", out.stdout))
        sums = sorted(re.findall(r"frame (ch \d+ pid \d+) .* sum (\d+)", out.stdout))
        got[mode + "_sums"] = sums
    assert got["plain"] == want, (got["plain"], want)
    assert got["sharded256"] == want and got["sharded4096"] == want
    assert got["sharded256_sums"] == got["plain_sums"] == got["sharded4096_sums"]


def test_sharded_class_reset_inside_a_payload_delivers_nothing_from_the_padding(oracle, product, tmp_path):
    """the sharded class completes the unfinished round with zeros when the stream stops (Reset, destruction).  A frame whose
    payload is in progress at that point used to be finished on the zeros and handed to the callback as an invalid frame; the reference
    drops it silently (ofdmflexframesync_reset, lib/multichannelrx.cc:139-140) and its destructor synchronizes nothing further.  Here the
    Reset() comes 40 channel samples before the end of every channel's fourth frame (the position is taken from the GPU receiver's own
    end_sample of a run without reset) and the stream is cut the same way inside the last frame; rounds of 4096 blocks hold the whole
    stream, so all the padding behind both points is synchronized.  Plain class, sharded class and oracle deliver the same frames."""
    import torch
    N, M, cp, tp = 4, 64, 8, 4
    iq, sent = oracle.synth_traffic(N, M, cp, tp, 8, payload_len=150, seed=5)
    iq = iq.astype(np.complex64)
    K = 2 * N
    rx = product.multichannelrx(N, M, cp, tp, max_payload_len=256)
    rx.Execute(torch.from_numpy(iq[:len(iq) // K * K]).cuda()); rx.Flush()
    ends = sorted(f.end_sample for f in rx.frames if f.channel == 0 and f.payload_valid)
    rx.close()
    assert len(ends) >= 6
    reset_at = (ends[3] - 40) * K + 3                        # 40 channel samples short of the fourth frame's last one, not on a block boundary
    stop_at = (ends[-1] - 40) * K + 5                        # ... and the stream ends inside the last frame
    cut = iq[:stop_at]
    f = tmp_path / "iq.bin"
    cut.tofile(f)
    ora = oracle.MultiChannelRx(N, M, cp, tp)
    ora.execute(cut[:reset_at]); ora.reset(); ora.execute(cut[reset_at:])
    want = sorted((f_.channel, (f_.header[0] << 8) | f_.header[1], f_.payload_valid, len(f_.payload)) for f_ in ora.frames)
    assert len(want) >= 4 * N and all(w[2] for w in want)
    host = os.path.join(ROOT, "liquid-usrp_amd", "host")
    subprocess.check_call(["make", "-C", host, "-s"])
    subprocess.check_call(["make", "-C", host, "-s", "shard_test"])
    for mode, extra in (("plain", {}), ("sharded256", {"MCRX_WORLD": "1", "MCRX_RANK": "0", "MCRX_SUB_BLOCKS": "256"}),
                        ("sharded4096", {"MCRX_WORLD": "1", "MCRX_RANK": "0", "MCRX_SUB_BLOCKS": "4096"})):
        out = subprocess.run([os.path.join(ROOT, "liquid-usrp_amd", "lib", "shard_test"), str(f), str(N), str(M), str(cp), str(tp), "10007", str(reset_at)],
                             env=dict(os.environ, **extra), capture_output=True, text=True, timeout=180)
        assert out.returncode == 0 and "done" in out.stdout, out.stderr[-2000:]
        got = sorted((int(a), int(b), int(d), int(e)) for a, b, c, d, e in
                     re.findall(r"frame ch (\d+) pid (\d+) hv (\d+) pv (\d+) len (\d+)", out.stdout))
        assert got == want, (mode, [g for g in got if g not in want], [w for w in want if w not in got])


TXEXE = os.path.join(ROOT, "liquid-usrp_amd", "lib", "multichannel_tx_ref")


@pytest.mark.skipif(not os.path.exists(TXEXE), reason="reference app binary not built")
def test_unchanged_reference_tx_app_feeds_both_receivers(oracle, product, tmp_path):
    """src/multichannel_tx.cc (unchanged) on the GPU multichanneltx class, its samples captured by the UHD
    shim; the GPU receiver and the oracle receiver must then report the same frames with the app's headers."""
    import torch
    N, M, cp, tp, P = 4, 64, 8, 4, 120
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "liquid-usrp_amd", "host"), "-s"])
    f = tmp_path / "tx.bin"
    nsamp = 32 * N * 5000
    env = dict(os.environ, MCTX_IQ_FILE=str(f), MCTX_IQ_SAMPLES=str(nsamp))
    out = subprocess.run([TXEXE, "-n", str(N), "-M", str(M), "-C", str(cp), "-T", str(tp), "-P", str(P), "-m", "qpsk",
                          "-c", "none", "-k", "h128", "-g", "0"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "warning" not in out.stderr                      # the app only updates channels that report ready
    iq = np.fromfile(f, np.complex64)
    assert len(iq) == nsamp
    rx = product.multichannelrx(N, M, cp, tp)
    rx.Execute(torch.from_numpy(iq).cuda()); rx.Flush()
    orx = oracle.MultiChannelRx(N, M, cp, tp)
    orx.execute(iq)
    key = lambda fr: (fr.channel, fr.header, fr.payload, fr.header_valid, fr.payload_valid)
    assert sorted(map(key, rx.frames)) == sorted(map(key, orx.frames))
    per_ch = {c: sorted((fr.header[0] << 8) | fr.header[1] for fr in rx.frames if fr.channel == c) for c in range(N)}
    for c in range(N):
        assert len(per_ch[c]) >= 5 and per_ch[c] == list(range(1, len(per_ch[c]) + 1)), per_ch      # pid counts up from 1
    assert all(fr.payload_valid and fr.header[2] == fr.channel and len(fr.payload) == P for fr in rx.frames)
    rx.close()


OTX = os.path.join(ROOT, "liquid-usrp_amd", "lib", "ofdmflexframe_tx_ref")
ORX = os.path.join(ROOT, "liquid-usrp_amd", "lib", "ofdmflexframe_rx_ref")


@pytest.mark.skipif(not (os.path.exists(OTX) and os.path.exists(ORX)), reason="reference app binaries not built")
def test_unchanged_apps_with_the_convolutional_code(oracle, product, tmp_path):
    """The same two applications with `-k v27` (liquid's r = 1/2, K = 7 convolutional code as the outer code, QAM16):
    liquid_getopt_str2fec knows the name, the GPU transmitter encodes it, the receiver application's Viterbi decoder
    returns every packet; the capture is also decoded by the oracle."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "liquid-usrp_amd", "host"), "-s"])
    f = tmp_path / "tx.bin"
    nframes, P = 8, 500
    env = dict(os.environ, MCTX_IQ_FILE=str(f), MCTX_IQ_SAMPLES=str(1 << 30))
    out = subprocess.run([OTX, "-N", str(nframes), "-P", str(P), "-m", "qam16", "-k", "v27"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    iq = np.fromfile(f, np.complex64)
    ora = oracle.FlexFrameSync(48, 6, 4)
    ora.execute(np.concatenate([iq, np.zeros(200, np.complex64)]))
    assert [((fr.header[0] << 8) | fr.header[1], fr.payload_valid, len(fr.payload)) for fr in ora.frames] == [(i, 1, P) for i in range(nframes)]
    assert all((fr.mod_scheme, fr.fec1) == (27, 11) for fr in ora.frames)
    np.concatenate([iq, np.zeros(4096, np.complex64)]).tofile(f)
    env = dict(os.environ, MCRX_IQ_FILE=str(f), MCRX_IQ_PACKET="4096")
    out = subprocess.run([ORX, "-t", "0.5"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    ids = [int(x) for x in re.findall(r"rx packet id:\s+(\d+)\n", out.stdout)]
    assert len(ids) >= nframes and set(ids) == set(range(nframes)), out.stdout[-3000:]
    assert "INVALID" not in out.stdout


@pytest.mark.skipif(not (os.path.exists(OTX) and os.path.exists(ORX)), reason="reference app binaries not built")
def test_unchanged_ofdmflexframe_tx_and_rx_apps_loop_back(oracle, product, tmp_path):
    """BASELINE configs[0] as the reference runs it: src/ofdmflexframe_tx.cc -> (file instead of a radio) ->
    src/ofdmflexframe_rx.cc, both unchanged, on the GPU ofdmtxrx class, at the applications' default
    numerology (M=48, cp=6, taper=4, QPSK, crc32 + Golay(24,12)).  The capture is also decoded by the oracle."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "liquid-usrp_amd", "host"), "-s"])
    f = tmp_path / "tx.bin"
    nframes, P = 12, 300
    env = dict(os.environ, MCTX_IQ_FILE=str(f), MCTX_IQ_SAMPLES=str(1 << 30))
    out = subprocess.run([OTX, "-N", str(nframes), "-P", str(P)], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert len(re.findall(r"tx packet id:", out.stdout)) == nframes
    iq = np.fromfile(f, np.complex64)
    ora = oracle.FlexFrameSync(48, 6, 4)
    ora.execute(np.concatenate([iq, np.zeros(200, np.complex64)]))
    assert [((fr.header[0] << 8) | fr.header[1], fr.payload_valid, len(fr.payload)) for fr in ora.frames] == \
        [(i, 1, P) for i in range(nframes)]
    assert all((fr.mod_scheme, fr.fec0, fr.fec1) == (40, 1, 7) for fr in ora.frames)
    # the receiver application replays the capture in a loop for one second
    np.concatenate([iq, np.zeros(4096, np.complex64)]).tofile(f)
    env = dict(os.environ, MCRX_IQ_FILE=str(f), MCRX_IQ_PACKET="4096")
    out = subprocess.run([ORX, "-t", "1.0"], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    ids = [int(x) for x in re.findall(r"rx packet id:\s+(\d+)\n", out.stdout)]
    assert len(ids) >= nframes and set(ids) == set(range(nframes)), out.stdout[-3000:]
    assert "INVALID" not in out.stdout
    m = re.search(r"frames detected\s+:\s+(\d+)\n\s+valid headers\s+:\s+(\d+).*\n\s+valid packets\s+:\s+(\d+)", out.stdout)
    assert m and m.group(1) == m.group(2) == m.group(3) and int(m.group(1)) == len(ids)


TXRX = os.path.join(ROOT, "liquid-usrp_amd", "lib", "multichannel_txrx_ref")


@pytest.mark.skipif(not os.path.exists(TXRX), reason="reference app binary not built")
def test_unchanged_multichannel_txrx_app_hears_its_own_bursts(oracle, product, tmp_path):
    """src/multichannel_txrx.cc (unchanged; its run time is fixed at 30 s) on the GPU multichanneltxrx class with
    the UHD stand-in looping transmit back into receive and recording what went over the air.  What the
    application's callbacks report must be exactly what the oracle receiver decodes from the recording.
    Nearly every packet comes back; the exceptions are the first few of a burst, which the application hands
    over while the transmit worker is still in its start-of-burst Reset() (lib/multichanneltxrx.cc:449 -- the
    reference has the same race), and the odd frame that starts while its idle channel is locked onto a
    neighbour's -60 dB leakage (liquid's detector is gain-normalised)."""
    N = 4
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "liquid-usrp_amd", "host"), "-s"])
    tee = tmp_path / "air.bin"
    env = dict(os.environ, MCTX_LOOPBACK="1", MCTX_TEE_FILE=str(tee))
    out = subprocess.run([TXRX, "-n", str(N), "-M", "64", "-C", "8", "-T", "4", "-P", "400"], env=env, capture_output=True,
                         text=True, timeout=180)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "frame pool exhausted" not in out.stderr
    sent = [(int(p) & 0xffff, int(n)) for p, n, c in
            re.findall(r"transmitting packet\s+(\d+) \(\s*(\d+) bytes\) on channel\s+(\d+)", out.stdout)]
    got = [(int(p), int(n)) for p, n in re.findall(r"header:pass, payload\[\s*(\d+),\s*(\d+) bytes\]:pass", out.stdout)]
    nfail = len(re.findall(r"header:FAIL", out.stdout))
    assert len(sent) >= 1000 and not re.search(r"header:pass.*:FAIL", out.stdout)
    iq = np.fromfile(tee, np.complex64)
    iq = iq[:len(iq) // (32 * N) * (32 * N)]
    orx = oracle.MultiChannelRx(N, 64, 8, 4)
    orx.execute(iq)
    want = [((f.header[0] << 8) | f.header[1], len(f.payload)) for f in orx.frames if f.header_valid]
    assert all(f.payload_valid for f in orx.frames if f.header_valid)
    # parity on what went over the air: the recording (45 M samples, ~5000 frames of random length in eleven bursts)
    # through the GPU receiver in 256-sample packets against the oracle's frame list, per channel in stream order
    rx = product.multichannelrx(N, 64, 8, 4)
    for i in range(0, len(iq), 256 * 64):
        rx.Execute(iq[i:i + 256 * 64])
    rx.Flush()
    # Idle channels of this application lock onto a neighbour's -60 dB leakage (the detector is gain-normalised); what
    # such a lock decodes is decisions on noise (EVM +19 .. +84 dB), and whether the real frame that starts underneath it
    # is still caught is decided by comparisons that flip when the ORACLE's own input is perturbed by 1e-6 of full scale
    # (scratch/refapp_hunt.py cuts such windows out; two were examined).  So: per channel, in stream order, the two frame
    # lists are equal except inside episodes that touch a CRC-failed header in one of them -- a handful in ~5000 frames.
    import difflib
    key = lambda f: (((f.header[0] << 8) | f.header[1]) if f.header_valid else -1, len(f.payload), int(f.header_valid), int(f.payload_valid))
    episodes = 0
    for c in range(N):
        a, b = [key(f) for f in rx.frames if f.channel == c], [key(f) for f in orx.frames if f.channel == c]
        for tag, i1, i2, j1, j2 in difflib.SequenceMatcher(None, a, b, autojunk=False).get_opcodes():
            if tag == "equal":
                continue
            near = a[max(i1 - 1, 0):i2 + 1] + b[max(j1 - 1, 0):j2 + 1]
            assert any(k[2] == 0 for k in near) and (i2 - i1) <= 3 and (j2 - j1) <= 3, (c, a[i1:i2], b[j1:j2])
            episodes += 1
    assert episodes <= 8 and abs(len(rx.frames) - len(orx.frames)) <= 8, episodes
    rx.close()
    # ... and exactly, with the leakage masked: a seeded noise floor 40 dB under the bursts (20 dB over a neighbour's leakage)
    # keeps idle channels from locking onto it, and on the first 12 M samples of that recording the two receivers then report
    # the same frames -- ids, lengths, flags, order, every payload byte -- with no episode to excuse.  (One is tolerated: the
    # recording is new in every run, and a detection within 1e-6 of its threshold can always exist.)
    n2 = min(len(iq), 12_000_000) // (32 * N) * (32 * N)
    act = np.abs(iq[:n2]) > 0
    sigma = float(np.sqrt(np.mean(np.abs(iq[:n2][act]) ** 2))) * 10 ** (-40 / 20) / np.sqrt(2.0)
    rng = np.random.default_rng(20260930)
    x = (iq[:n2] + sigma * (rng.standard_normal(n2, dtype=np.float32) + 1j * rng.standard_normal(n2, dtype=np.float32))).astype(np.complex64)
    orx2 = oracle.MultiChannelRx(N, 64, 8, 4)
    orx2.execute(x)
    rx2 = product.multichannelrx(N, 64, 8, 4)
    for i in range(0, len(x), 256 * 64):
        rx2.Execute(x[i:i + 256 * 64])
    rx2.Flush()
    full = lambda f: (f.header, f.payload, int(f.header_valid), int(f.payload_valid))
    differ = 0
    for c in range(N):
        a, b = [full(f) for f in rx2.frames if f.channel == c], [full(f) for f in orx2.frames if f.channel == c]
        if a != b:
            ops = [o for o in difflib.SequenceMatcher(None, [k[0] for k in a], [k[0] for k in b], autojunk=False).get_opcodes() if o[0] != "equal"]
            differ += max(1, len(ops))
    print("txrx recording, leakage masked: %d frames, %d differing episodes" % (len(orx2.frames), differ))
    assert len(orx2.frames) >= 500 and differ <= 1, differ
    rx2.close()
    # the live callbacks: the same list up to what two free-running worker threads and a wall clock do to the stand-in's
    # air (a receive worker that lags is handed the stream with a sample gap; the recording has none)
    sg, sw = set(got), set(want)
    assert len(sg ^ sw) <= 4 and abs(nfail - sum(1 for f in orx.frames if not f.header_valid)) <= 2, (sorted(sg - sw), sorted(sw - sg))
    assert len(got) >= 0.98 * len(sent) and set(got) <= set(sent)


FDX = os.path.join(ROOT, "liquid-usrp_amd", "lib", "fullduplex_txrx_ref")


@pytest.mark.skipif(not os.path.exists(FDX), reason="reference app binary not built")
def test_unchanged_fullduplex_app_receives_while_transmitting():
    """src/fullduplex_txrx.cc (unchanged): the receiver of one ofdmtxrx object runs while its transmitter sends
    200 frames; with the stand-in looped back every frame must come out of the callback, valid and in order."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "liquid-usrp_amd", "host"), "-s"])
    env = dict(os.environ, MCTX_LOOPBACK="1")
    # Two threads and a wall clock: the stand-in treats the air as continuous inside a burst, but a transmitter
    # thread that is descheduled long enough mid-frame still tears that frame (seen once in ~100 runs; the same
    # receiver code loses nothing in 27 000 frames of scratch/gap_hunt.py).  One repeat is allowed for that.
    for attempt in range(3):
        out = subprocess.run([FDX, "-N", "200", "-P", "500", "-m", "qam16", "-c", "h128", "-k", "none"], env=env,
                             capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr[-2000:]
        assert len(re.findall(r"tx packet id:", out.stdout)) == 200
        ids = [int(x) for x in re.findall(r"rx packet id:\s+(\d+)\n", out.stdout)]
        assert "INVALID" not in out.stdout and ids == sorted(set(ids)) and len(ids) >= 198
        if ids == list(range(200)):
            break
    assert ids == list(range(200)), (len(ids), out.stdout[-1500:])


def test_ofdmtxrx_blocking_receiver_worker_and_debug_dump(tmp_path):
    """The reference's second ofdmtxrx constructor selects ofdmtxrx_rx_worker_blocking (lib/ofdmtxrx.cc:642-739): another thread
    edits every received packet between rx_buffer_filled_cond and rx_buffer_modified_cond.  host/blocking_test.cc turns the
    samples by 180 degrees (every frame still arrives) and then zeroes them (nothing arrives); with MCRX_DEBUG_DIR set,
    debug_enable() leaves the received frames' equalised symbols in a .m file at destruction."""
    host = os.path.join(ROOT, "liquid-usrp_amd", "host")
    subprocess.check_call(["make", "-C", host, "-s"])
    subprocess.check_call(["make", "-C", host, "-s", "blocking_test"])
    env = dict(os.environ, MCTX_LOOPBACK="1", MCRX_DEBUG_DIR=str(tmp_path))
    out = subprocess.run([os.path.join(ROOT, "liquid-usrp_amd", "lib", "blocking_test")], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    m = re.search(r"phase1 (\d+) (\d+)  phase2 (\d+) (\d+)  packets_edited (\d+)", out.stdout)
    assert m, out.stdout
    sent1, got1, sent2, got2, edited = map(int, m.groups())
    assert got1 >= sent1 - 1 and got2 == 0 and edited > 0, out.stdout
    dump = tmp_path / "ofdmtxrx_framesyms.m"
    assert dump.exists() and "framesyms{1}" in dump.read_text()
