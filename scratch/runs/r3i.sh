#!/bin/bash
O=gpurun_out/r3i; mkdir -p $O
python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | tee $O/tests.log
python scratch/aper_probe.py 0 2>&1 | grep -v amdgpu.ids | tee $O/probe.log
MCRX_SEEK_BURST=0 python scratch/aper_probe.py 0 2>&1 | grep -v amdgpu.ids | tee -a $O/probe.log
python bench.py --no-cpu --no-harvest --steps 30 --warmup 8 --serial-steps 3 2>$O/err.log | tail -1 > $O/bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3i/bench.json"))
print("value", d["value"], "hit", d["spec_hit_rate"], "aperiodic", d.get("value_aperiodic"), json.dumps(d.get("value_aperiodic_detail"))[290:900])
PY
