#!/bin/bash
python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^  File\|^Extension" | tail -4
for f in 2 16; do
  echo "== frames $f"
  python bench.py --frames $f --no-aperiodic --no-cpu --no-harvest 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['verified']['ok'], 'alone', r['kernels_ms'], 'ovl', r['kernels_ms_overlapped'])
"
done
