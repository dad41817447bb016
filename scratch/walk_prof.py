"""cycle profile of the walking scout on a small ragged stream (library built with -DSY_PROFILE=1, MCRX_DEBUG=2)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from __graft_entry__ import load_product
prod = load_product()
N, M, cp = 64, 64, 8
tx = prod.multichanneltx(N, M, cp, 4)
d, s, _ = tx.generate_ragged(40000, len_lo=64, len_hi=1200, gap_max=3, long_every=8, long_max=184, seed=77)
tx.close()
rx = prod.multichannelrx(N, M, cp, 4, serial=1, max_payload_len=1200, max_frames=4096)
rx.Execute(d); rx.Flush()
print("frames", len(rx.frames), "sent ch0", len(s[0]))
rx.close()
