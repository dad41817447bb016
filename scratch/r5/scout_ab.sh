#!/bin/bash
# the general state machine's segment waves (default) against the lean ones (--scout-build 2), alternating on one box
cd $GRAFT_REPO_ROOT
for i in 1 2; do for sb in 0 2; do
  python bench.py --no-cpu --no-variants --no-harvest --no-configs --scout-build $sb 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('scout_build $sb:', d['value'], 'aperiodic', d['value_aperiodic'], 'sync alone', d['roofline']['kernels_ms']['sync_kernel'])"
done; done
