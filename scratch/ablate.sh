#!/bin/bash
# usage: ablate.sh <list of MCRX_ABLATE values>
for ab in "$@"; do
  MCRX_ABLATE=$ab python bench.py --steps 100 --warmup 20 --no-cpu 2>/dev/null | tail -1 > /tmp/ab.json
  python - <<PY
import json
d=json.load(open('/tmp/ab.json'))
print("ablate", $ab, "channelizer_ms", d["roofline"]["channelizer_ms"], "sync_ms", d["roofline"]["sync_ms"], "value", d["value"])
PY
done
