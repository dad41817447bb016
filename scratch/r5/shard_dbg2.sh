#!/bin/bash
# debug prints of one sharded run (MCRX_DEBUG: 4 = per-channel adoption counts, 16 = emits of channel 0, 64 = segment waves of channel 0)
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_oracle
import numpy as np
o = load_oracle()
iq, sent = o.synth_traffic(4, 64, 8, 4, 8, payload_len=150, seed=3)
iq.astype(np.complex64).tofile("/tmp/iq.bin")
PY
MCRX_DEBUG=${DBG:-84} MCRX_WORLD=1 MCRX_RANK=0 MCRX_SUB_BLOCKS=${SUB:-512} liquid-usrp_amd/lib/shard_test /tmp/iq.bin 4 64 8 4 10007 0 2>&1 | grep -v "^frame ch [123]" | head -${LINES_MAX:-400}
