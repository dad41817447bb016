#!/bin/bash
# the anchor carried over from the previous push (default) against the anchor phase of rounds 4-5 (--acquisition 3), alternating on one box
cd $GRAFT_REPO_ROOT
for i in 1 2; do for aq in 0 3; do
  python bench.py --no-cpu --no-variants --no-harvest --acquisition $aq 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('acquisition $aq:', d['value'], 'aperiodic', d['value_aperiodic'], {k:v.get('value') for k,v in d['configs'].items()}, 'sync alone', d['roofline']['kernels_ms']['sync_kernel'])"
done; done
for aq in 0 3; do python bench.py --pipeline --no-cpu --steps 20 --warmup 5 --acquisition $aq 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipeline, acquisition $aq:', d['value'])"; done
