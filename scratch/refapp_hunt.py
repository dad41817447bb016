"""hunt for a multichannel_txrx air recording on which the GPU receiver and the oracle disagree; cut the window out"""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_product, load_oracle
prod, ora = load_product(), load_oracle()
N, K = 4, 8
TXRX = os.path.join(ROOT, "liquid-usrp_amd", "lib", "multichannel_txrx_ref")
key = lambda f: (((f.header[0] << 8) | f.header[1]) if f.header_valid else -1, len(f.payload), f.channel, int(f.header_valid), int(f.payload_valid))


def gpu(iq, step=256 * 64, **env):
    for k, v in env.items():
        os.environ[k] = v
    rx = prod.multichannelrx(N, 64, 8, 4)
    for k in env:
        del os.environ[k]
    for i in range(0, len(iq), step):
        rx.Execute(iq[i:i + step])
    rx.Flush()
    fr = list(rx.frames)
    rx.close()
    return fr


def oracle(iq):
    o = ora.MultiChannelRx(N, 64, 8, 4); o.execute(iq)
    return list(o.frames)


def per_ch(fr):
    d = {c: [] for c in range(N)}
    for f in fr:
        d[f.channel].append(f)
    return d


tee = "/tmp/air.bin"
for attempt in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    if os.path.exists(tee):
        os.remove(tee)
    env = dict(os.environ, MCTX_LOOPBACK="1", MCTX_TEE_FILE=tee)
    subprocess.run([TXRX, "-n", "4", "-M", "64", "-C", "8", "-T", "4", "-P", "400"], env=env, capture_output=True, text=True, timeout=180)
    iq = np.fromfile(tee, np.complex64)
    iq = iq[:len(iq) // (16 * N) * (16 * N)]
    of, gf = oracle(iq), gpu(iq)
    same = all([key(f) for f in of if f.channel == c] == [key(f) for f in gf if f.channel == c] for c in range(N))
    print("attempt", attempt, "samples", len(iq), "oracle", len(of), "gpu", len(gf), "equal", same, flush=True)
    if same:
        continue
    og, gg = per_ch(of), per_ch(gf)
    for c in range(N):
        a, b = [key(f) for f in og[c]], [key(f) for f in gg[c]]
        if a == b:
            continue
        i = 0
        while i < min(len(a), len(b)) and a[i] == b[i]:
            i += 1
        e = gg[c][i - 1].end_sample if i else 0
        print(" channel", c, "diverges at frame", i, "after channel-rate sample", e, "\n   oracle", a[i:i + 3], "\n   gpu   ", b[i:i + 3],
              "gpu ends", [f.end_sample for f in gg[c][i:i + 3]], flush=True)
        for name, kw in (("nospec", dict(MCRX_NO_SPEC="1")), ("serial", dict(MCRX_SERIAL="1")), ("rounds1", dict(MCRX_SCOUT_ROUNDS="1"))):
            g2 = per_ch(gpu(iq, **kw))
            print("   mode", name, "equal on this channel:", [key(f) for f in g2[c]] == a, flush=True)
        g2 = per_ch(gpu(iq, step=1 << 20))
        print("   mode 1M-sample pushes equal on this channel:", [key(f) for f in g2[c]] == a, flush=True)
        for back in (40000, 400000):
            lo = max(0, (e - back) * K) // (16 * N) * (16 * N)
            hi = min(len(iq), (e + 60000) * K)
            w = iq[lo:hi]
            w = w[:len(w) // (16 * N) * (16 * N)]
            ow, gw = per_ch(oracle(w)), per_ch(gpu(w))
            eq = [key(f) for f in ow[c]] == [key(f) for f in gw[c]]
            print("   window back", back, "samples", len(w), "oracle", len(ow[c]), "gpu", len(gw[c]), "equal", eq, flush=True)
            if not eq:
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                w.tofile(os.path.join(ROOT, "gpurun_out", "air_window_ch%d.bin" % c))
                print("   saved window", flush=True)
                break
        break
    break
