#!/bin/bash
O=gpurun_out/r3g; mkdir -p $O
python -m pytest tests/test_gpu_tx.py -q -x -k "ragged" 2>&1 | tail -15 | tee $O/tests.log
python bench.py --no-cpu --no-harvest --steps 30 --warmup 8 --serial-steps 3 2>$O/err.log | tail -1 > $O/bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3g/bench.json"))
print("value", d["value"], "aperiodic", d.get("value_aperiodic"), json.dumps(d.get("value_aperiodic_detail"), indent=0)[:1500])
PY
tail -5 $O/err.log
