#!/bin/bash
# do the pipeline paths' extra streams share hardware queues with the receiver's? (GPU_MAX_HW_QUEUES: 4 by default)
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu --no-harvest --no-aperiodic --no-configs --steps 20 --warmup 5 --reps 3 --serial-steps 2 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  value', d['value'], d['value_min'], d['value_max'], d['verified']['ok'])"; }
for q in 4 8 16; do
export GPU_MAX_HW_QUEUES=$q
echo "== GPU_MAX_HW_QUEUES=$q direct"; run
echo "== GPU_MAX_HW_QUEUES=$q pipeline"; run --pipeline
done
