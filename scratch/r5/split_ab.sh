#!/bin/bash
# the normally empty launches on a stream of their own (this tree's library) against the library of the commit before
# (scratch/r5/bisect/HEAD.so, built from a worktree), alternating on one box
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
for i in 1 2 3; do for t in new inline; do
  lib=$R/liquid-usrp_amd/lib/libmcrx_hip.so; [ $t = inline ] && lib=$R/scratch/r5/bisect/HEAD.so
  MCRX_LIB=$lib python bench.py --no-cpu --no-variants --no-harvest 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$t:', d['value'], 'aperiodic', d['value_aperiodic'], {k:v.get('value') for k,v in d['configs'].items()})"
done; done
for t in new inline; do
  lib=$R/liquid-usrp_amd/lib/libmcrx_hip.so; [ $t = inline ] && lib=$R/scratch/r5/bisect/HEAD.so
  MCRX_LIB=$lib python bench.py --pipeline --no-cpu --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipeline $t:', d['value'])"
done
