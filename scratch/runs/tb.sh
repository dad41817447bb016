#!/bin/bash
# tests + bench line summary
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python bench.py --no-cpu $BENCH_ARGS 2>&1 | tail -1 > gpurun_out/bq.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/bq.json"))
r=d["roofline"]
print("value", d["value"], "ms/step", d["ms_per_step"], r["kernel"], r["frac"], r["traffic"], r["kernels_ms"], d["verified"]["ok"])
PY
