#!/bin/bash
# round 4, first call: FETCH_SIZE calibration + the r3 build's bench line on this session's box (baseline)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4a; mkdir -p $O
bash scratch/fetchcal.sh > $O/fetchcal.txt 2>&1
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --no-cpu 2>$O/bench.err | tail -1 > $O/bench.json
python - <<PY
import json
d = json.loads(open("$O/bench.json").read())
print("value", d["value"], "harvest", d.get("value_with_harvest"), "aper", d.get("value_aperiodic"), d["roofline"]["kernels_ms"], d["roofline"]["kernels_ms_overlapped"])
PY
tail -30 $O/fetchcal.txt
