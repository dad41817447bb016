#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_fifth; mkdir -p $O
for s in 11 12 13; do python scratch/soak_dbg.py $s 2>&1 | grep -E "bad" | awk '{s+=$4} END {print "seed bad total", s}'; done
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
B="python bench.py --no-cpu --no-harvest"
timeout 300 $B > $O/b_default.json 2> $O/b_default.err; echo "default rc=$?"
python - <<PY
import json
d=json.load(open("$O/b_default.json"))
print(d["value"], d["ms_per_step"], d["spec_hit_rate"], d["roofline"]["kernels_ms"], d["roofline"]["kernels_ms_overlapped"], d["verified"]["ok"])
PY
python bench.py --pipeline --no-cpu --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipeline path', d['value'], d['spec_hit_rate'], d['verified']['ok'])"
