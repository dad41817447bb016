#!/bin/bash
for f in 2 12 16; do
  echo "== frames $f"
  python bench.py --frames $f --no-aperiodic --no-cpu --no-harvest 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['verified']['ok'], 'alone', r['kernels_ms']['payload_kernel'], 'ovl', r['kernels_ms_overlapped'])
"
done
