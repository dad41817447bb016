#!/bin/bash
# hardware queues per process (GPU_MAX_HW_QUEUES 8 = bench.py's default, 12, 16) on the legs with the most streams
cd $GRAFT_REPO_ROOT
for i in 1 2; do for q in 8 12 16; do
  echo "queues $q: pipeline $(GPU_MAX_HW_QUEUES=$q python bench.py --pipeline --no-cpu --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])") duplex $(GPU_MAX_HW_QUEUES=$q python bench_duplex.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])")"
done; done
