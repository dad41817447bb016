#!/bin/bash
# round 4: segment-parallel acquisition -- GPU tests, bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4c; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "pytest rc $?" >> $O/tests.txt
tail -25 $O/tests.txt
timeout 600 python bench.py --no-cpu 2>$O/bench.err | tail -1 > $O/bench.json
python - <<PY
import json
d = json.loads(open("$O/bench.json").read())
print("value", d["value"], "harvest", d.get("value_with_harvest"), "aper", d.get("value_aperiodic"), d["roofline"]["kernels_ms"], d["roofline"]["kernels_ms_overlapped"], d["verified"], d["frames_acquired"])
print(d.get("value_aperiodic_detail"))
PY
tail -5 $O/bench.err
