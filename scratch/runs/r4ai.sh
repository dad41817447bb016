#!/bin/bash
# where the pipeline path's distance to the direct one comes from: rounds that cut the stream between frames (equal slabs) against rounds that cut it anywhere
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu --no-harvest --no-aperiodic --no-configs --steps 20 --warmup 5 --reps 3 --serial-steps 2 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  value', d['value'], d['value_min'], d['value_max'], d['verified']['ok'], d['frames_acquired'])"; }
echo "== direct"; run
echo "== pipeline"; run --pipeline
export BENCH_EQUAL_PADS=1
echo "== direct, equal slabs"; run
echo "== pipeline, equal slabs (cuts between frames)"; run --pipeline
echo "== pipeline, equal slabs, defer 0"; run --pipeline --defer 0
