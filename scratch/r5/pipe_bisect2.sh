#!/bin/bash
# bench.py --pipeline of round 4's tree with the library of a series of this round's commits (scratch/r5/bisect/<sha>.so, built by hand from
# git worktrees), twice round the list
cd $GRAFT_REPO_ROOT/scratch/r5/oldtree
for i in 1 2; do
  echo "round4-lib $(python bench.py --pipeline --no-cpu --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])")"
  for f in $GRAFT_REPO_ROOT/scratch/r5/bisect/*.so $GRAFT_REPO_ROOT/liquid-usrp_amd/lib/libmcrx_hip.so; do
    echo "$(basename $f) $(MCRX_LIB=$f python bench.py --pipeline --no-cpu --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])")"
  done
done
