#!/bin/bash
# bench.py --pipeline through the torch mirror of the schedule (--exchange torch) and through the C pipeline (--exchange c): this round's tree
# against round 4's, alternating
cd $GRAFT_REPO_ROOT
for i in 1 2; do for x in torch c; do for t in new old; do
  d=.; [ $t = old ] && d=scratch/r5/oldtree
  echo "$x $t $(cd $d && python bench.py --pipeline --exchange $x --no-cpu --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])")"
done; done; done
