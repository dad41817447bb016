#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_depth; mkdir -p $O
B="python bench.py --no-cpu --no-harvest"
for d in 3 4 5 6 8; do
MCRX_SLOTS=$d timeout 300 $B > $O/b_$d.json 2>/dev/null
python - <<PY
import json
d=json.load(open("$O/b_$d.json"))
print("slots $d", d["value"], d["ms_per_step"], d["spec_hit_rate"], d["roofline"]["kernels_ms_overlapped"], d["verified"]["ok"])
PY
done
MCRX_SLOTS=5 MCRX_NO_PRIO=1 timeout 300 $B > $O/b_np.json 2>/dev/null; python -c "import json; d=json.load(open('$O/b_np.json')); print('slots 5 noprio', d['value'])"
MCRX_SLOTS=5 MCRX_SCOUT_ROUNDS=3 timeout 300 $B > $O/b_r3.json 2>/dev/null; python -c "import json; d=json.load(open('$O/b_r3.json')); print('slots 5 rounds3', d['value'])"
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_refapp.py > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -3 $O/pytest.log
for s in 11 12; do python scratch/soak_dbg.py $s 2>&1 | grep -E "bad" | awk '{s+=$4} END {print "seed bad total", s}'; done
