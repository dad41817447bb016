#!/bin/bash
for v in 0 2 6 8; do
MCRX_NO_FAST=$v python bench.py --no-cpu 2>/dev/null | tail -1 > gpurun_out/bq.json
python - <<PY
import json
d=json.load(open("gpurun_out/bq.json"))
r=d["roofline"]
print("no_fast=$v value", d["value"], "chan", r["channelizer_ms"], "scout", r["scout_ms"], "payload", r["payload_ms"], d["verified"]["ok"])
PY
done
