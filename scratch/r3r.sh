#!/bin/bash
python -m pytest tests/test_gpu_stream.py tests/test_gpu_tx.py tests/test_gpu_soak.py -q -x -k "policy or ragged or soak" 2>&1 | tail -2
( time python bench.py > gpurun_out/bench_r3_last.json 2>/dev/null ) 2>&1 | grep real
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_r3_last.json").read().strip().splitlines()[-1])
print(d["value"], d["value_aperiodic"], d["value_aperiodic_over_value"], d["value_aperiodic_detail"]["verified"]["ok"], d["verified"]["ok"], d["value_with_harvest"])
PY
