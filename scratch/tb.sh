#!/bin/bash
# tests + bench line summary
python -m pytest tests -m gpu -q -x 2>&1 | tail -3
python bench.py --no-cpu 2>&1 | tail -1 > gpurun_out/bq.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/bq.json"))
r=d["roofline"]
print("value", d["value"], "ms/step", d["ms_per_step"], "chan", r["channelizer_ms"], "scout", r["scout_ms"], "payload", r["payload_ms"], d["verified"])
PY
