"""replay the air recording of the multichannel_txrx app through the GPU receiver in several modes and compare with the oracle"""
import os, sys, subprocess, re
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_product, load_oracle
prod, ora = load_product(), load_oracle()
tee = "/tmp/air.bin"
if not os.path.exists(tee):
    TXRX = os.path.join(ROOT, "liquid-usrp_amd", "lib", "multichannel_txrx_ref")
    env = dict(os.environ, MCTX_LOOPBACK="1", MCTX_TEE_FILE=tee)
    subprocess.run([TXRX, "-n", "4", "-M", "64", "-C", "8", "-T", "4", "-P", "400"], env=env, capture_output=True, text=True, timeout=180)
N = 4
iq = np.fromfile(tee, np.complex64)
iq = iq[:len(iq) // (16 * N) * (16 * N)]
o = ora.MultiChannelRx(N, 64, 8, 4); o.execute(iq)
want = sorted(((f.header[0] << 8) | f.header[1], len(f.payload), f.channel, int(f.header_valid), int(f.payload_valid)) for f in o.frames)
print("oracle frames", len(want), "samples", len(iq))
mode = sys.argv[1] if len(sys.argv) > 1 else "host256"
rx = prod.multichannelrx(N, 64, 8, 4)
if mode == "host256":
    for i in range(0, len(iq), 256 * 64):
        rx.Execute(iq[i:i + 256 * 64])
elif mode == "bulk":
    rx.Execute(iq)
else:
    import torch
    d = torch.from_numpy(iq).cuda()
    step = int(mode)
    for i in range(0, len(iq), step):
        rx.Execute(d[i:i + step])
rx.Flush()
got = sorted(((f.header[0] << 8) | f.header[1], len(f.payload), f.channel, int(f.header_valid), int(f.payload_valid)) for f in rx.frames)
print(mode, os.environ.get("MCRX_SCOUT_ROUNDS"), os.environ.get("MCRX_NO_SPEC"), os.environ.get("MCRX_SERIAL"), "gpu frames", len(got), "equal", got == want)
if got != want:
    sg, sw = set(got), set(want)
    print("  only gpu:", sorted(sg - sw)[:6], " only oracle:", sorted(sw - sg)[:6])
    ends = {((f.header[0] << 8) | f.header[1], f.channel): f.end_sample for f in rx.frames}
    for k in sorted(sg - sw)[:3]:
        print("   gpu-only frame end_sample", ends.get((k[0], k[2])))
rx.close()
