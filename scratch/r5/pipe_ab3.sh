#!/bin/bash
# bench.py --pipeline: this round's tree and round 4's, alternating, four times each on one box
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for t in new old; do
  d=.; [ $t = old ] && d=scratch/r5/oldtree
  echo "$t $(cd $d && python bench.py --pipeline --no-cpu --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('value_min'), d.get('value_max'))")"
done; done
