#!/bin/bash
O=gpurun_out/r3h; mkdir -p $O
python -m pytest tests/test_gpu_tx.py -q -x -k "ragged" 2>&1 | tail -5 | tee $O/tests.log
python scratch/aper_probe.py 1 2>&1 | grep -v amdgpu.ids | tee $O/probe.log
python scratch/aper_probe.py 0 2>&1 | grep -v amdgpu.ids | tee -a $O/probe.log
python bench.py --no-cpu --no-harvest --steps 30 --warmup 8 --serial-steps 3 2>$O/err.log | tail -1 > $O/bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3h/bench.json"))
print("value", d["value"], "hit", d["spec_hit_rate"], "aperiodic", d.get("value_aperiodic"), json.dumps(d.get("value_aperiodic_detail"))[:1200])
PY
