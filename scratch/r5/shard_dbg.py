"""debug: shard_test plain vs sharded, print frame multiset differences"""
import os, re, subprocess, sys, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_oracle
oracle = load_oracle()
N, M, cp, tp = 4, 64, 8, 4
iq, sent = oracle.synth_traffic(N, M, cp, tp, 8, payload_len=150, seed=3)
iq = iq.astype(np.complex64)
f = "/tmp/iq.bin"; iq.tofile(f)
K = 2 * N
frame = len(iq) // 8
reset_at = (3 * frame + frame // 2) // K * K + 3
print("samples", len(iq), "blocks", len(iq) // K, "reset_at", reset_at, reset_at // K)
for reset in (0, reset_at):
    for mode, extra in (("plain", {}), ("s256", {"MCRX_WORLD": "1", "MCRX_RANK": "0", "MCRX_SUB_BLOCKS": "256"}),
                        ("s512", {"MCRX_WORLD": "1", "MCRX_RANK": "0", "MCRX_SUB_BLOCKS": "512"}),
                        ("s4096", {"MCRX_WORLD": "1", "MCRX_RANK": "0", "MCRX_SUB_BLOCKS": "4096"})):
        out = subprocess.run([os.path.join(ROOT, *(sys.argv[1:2] or ["liquid-usrp_amd/lib"])[0].split("/"), "shard_test"), f, str(N), str(M), str(cp), str(tp), "10007", str(reset)],
                             env=dict(os.environ, **extra), capture_output=True, text=True, timeout=180)
        fr = re.findall(r"frame ch (\d+) pid (\d+) hv (\d+) pv (\d+) len (\d+)", out.stdout)
        cnt = collections.Counter(fr)
        print(reset, mode, "rc", out.returncode, "frames", len(fr), "distinct", len(cnt), "dups", sorted(k for k, v in cnt.items() if v > 1)[:6], out.stderr[-300:])
