#!/bin/bash
O=gpurun_out/r3o; mkdir -p $O
python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee $O/tests.log
B="python bench.py --no-cpu --no-harvest --no-aperiodic --steps 30 --warmup 8 --serial-steps 3"
$B 2>/dev/null | tail -1 > $O/bench_map.json
MCRX_NO_ILMAP=1 $B 2>/dev/null | tail -1 > $O/bench_nomap.json
python - <<'PY'
import json
for n in ("map", "nomap"):
    d = json.load(open("gpurun_out/r3o/bench_%s.json" % n)); r = d["roofline"]
    print(n, d["value"], r["kernels_ms"], d["verified"]["ok"])
PY
