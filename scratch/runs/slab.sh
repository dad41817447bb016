#!/bin/bash
for sb in "$@"; do
  python bench.py --steps 5 --warmup 2 --no-cpu --slab-blocks $sb 2>/dev/null | tail -1 > /tmp/ab.json
  python - <<PY
import json
d=json.load(open('/tmp/ab.json'))
print("slab", $sb, "channelizer_ms", d["roofline"]["channelizer_ms"], "sync_ms", d["roofline"]["sync_ms"], "value", d["value"])
PY
done
