#!/bin/bash
# round 2, second GPU pass: new tests + speculation rounds / stream priority experiments
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_second; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_refapp.py > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
B="python bench.py --no-cpu --no-harvest"
timeout 300 $B > $O/b_default.json 2> $O/b_default.err; echo "default rc=$?"
MCRX_SCOUT_ROUNDS=1 timeout 300 $B > $O/b_rounds1.json 2>/dev/null
MCRX_SCOUT_ROUNDS=3 timeout 300 $B > $O/b_rounds3.json 2>/dev/null
MCRX_NO_PRIO=1 timeout 300 $B > $O/b_noprio.json 2>/dev/null
timeout 300 $B --chunk-blocks 25344 > $O/b_chunk4.json 2>/dev/null
for f in default rounds1 rounds3 noprio chunk4; do echo "== $f"; python - <<PY
import json
d=json.load(open("$O/b_$f.json"))
print(d["value"], d["ms_per_step"], d["spec_hit_rate"], d["roofline"]["kernels_ms"], d["roofline"]["kernels_ms_overlapped"], d["verified"]["ok"])
PY
done
tail -3 $O/b_default.err
