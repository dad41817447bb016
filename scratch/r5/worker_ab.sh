#!/bin/bash
# lean payload workers: butterfly exchanges through the LDS crossbar (default) against DPP / permlane swaps on the VALU (--worker-build 1)
cd $GRAFT_REPO_ROOT
for i in 1 2; do for wb in 0 1; do
  python bench.py --no-cpu --no-variants --no-harvest --no-aperiodic --worker-build $wb 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('worker_build $wb:', d['value'], {k:v.get('value') for k,v in d['configs'].items()}, 'payload alone', d['roofline']['kernels_ms']['payload_kernel'])"
done; done
