#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_half; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "channelizer or full_chain" 2>&1 | tail -2
B="python bench.py --no-cpu --no-harvest"
timeout 300 $B > $O/b_half2.json 2>/dev/null
MCRX_LIB=$PWD/liquid-usrp_amd/lib/libmcrx_hip_b1.so timeout 300 $B > $O/b_half1.json 2>/dev/null
MCRX_ABLATE=256 timeout 300 $B > $O/b_full.json 2>/dev/null
for f in half2 half1 full; do python - <<PY
import json
d=json.load(open("$O/b_$f.json"))
print("$f", d["value"], d["ms_per_step"], d["roofline"]["kernels_ms"]["channelizer_kernel"], d["roofline"]["kernels_ms_overlapped"], d["verified"]["ok"])
PY
done
