"""Why does the loopback decode nothing with liquid's "timing backoff correction" G *= B applied (D6 = 1)?  CPU oracle only.
Hypothesis: with G *= B the equaliser no longer carries the data windows' phase ramp (2 pi backoff / M per subcarrier), the pilot line fit has to
remove it every symbol, and the fit's slope is SMOOTHED -- p1 = 0.3 p1 + 0.7 p1' with p1' = 0 at the first symbol -- so the first header
symbol keeps 70 % of the ramp: +-3.6 rad at the band edge for M = 64, and the BPSK header fails.  Test: the same build with the smoothing
off (alpha = 1)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASE = r'''
import sys, json; sys.path.insert(0, %r)
import numpy as np, oracle as O
out = {}
for (N, M, cp, mod, fec1, plen) in ((2, 64, 8, 40, 6, 300), (2, 256, 32, 27, 7, 300), (2, 48, 6, 39, 1, 100)):
    iq, sent = O.synth_traffic(N, M, cp, 4, 4, payload_len=plen, mod=mod, fec1=fec1, seed=5)
    rx = O.MultiChannelRx(N, M, cp, 4); rx.execute(iq)
    out["M%%d" %% M] = [len(rx.frames), sum(1 for f in rx.frames if f.header_valid), sum(1 for f in rx.frames if f.payload_valid)]
print(json.dumps(out))
''' % os.path.join(ROOT, "oracle")
res = {}
for tag, lib in (("default (no B, alpha 0.3)", None), ("D6: G *= B, alpha 0.3", "/tmp/liboracle_d6.so"), ("D6: G *= B, alpha 1.0 (no slope smoothing)", "/tmp/liboracle_d6a1.so"), ("no B, alpha 1.0", "/tmp/liboracle_a1.so")):
    env = dict(os.environ)
    if lib: env["LL_ORACLE_LIB"] = lib
    r = subprocess.run([sys.executable, "-c", CASE], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    res[tag] = json.loads(r.stdout.decode().strip().split("\n")[-1]) if r.returncode == 0 else r.stderr.decode()[-300:]
print(json.dumps({"frames [detected, header valid, payload valid] of 8 sent per case": res}, indent=1))
