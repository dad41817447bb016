#!/bin/bash
for p in 13312 17000 20000 26000; do
  echo "== pad $p"
  MCRX_WALK_LDS_PAD=$p python bench.py --no-cpu --no-harvest --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], 'aper', d['value_aperiodic'], d['value_aperiodic_detail']['verified']['ok'], d['value_aperiodic_detail']['kernels_ms_overlapped'])
"
done
