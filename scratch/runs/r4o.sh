#!/bin/bash
cd $GRAFT_REPO_ROOT
for t in 0 1 3 0 3; do echo "== MCRX_EXPERIMENT_SKIP_RARE=$t"; MCRX_EXPERIMENT_SKIP_RARE=$t python bench.py --no-cpu --no-harvest --no-aperiodic --no-configs --steps 20 --warmup 5 --reps 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  value', d['value'], d['value_min'], d['value_max'], 'overlapped', d['roofline']['kernels_ms_overlapped'], d['verified']['ok'])"; done
