#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_chan; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "channelizer or full_chain" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_baseline_shapes.py -q -x 2>&1 | tail -3
B="python bench.py --no-cpu --no-harvest"
timeout 300 $B > $O/b_v2.json 2>$O/b_v2.err; echo rc=$?
MCRX_ABLATE=256 timeout 300 $B > $O/b_v1.json 2>/dev/null
for f in v2 v1; do python - <<PY
import json
d=json.load(open("$O/b_$f.json"))
print("$f", d["value"], d["ms_per_step"], d["roofline"]["kernels_ms"], d["roofline"]["frac"], d["verified"]["ok"])
PY
done
tail -3 $O/b_v2.err
