#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
VARIANTS="new cadence" bash scratch/r4g.sh
for ns in 2 4 8 12 16; do echo "== MCRX_NSEG=$ns"; MCRX_NSEG=$ns python bench.py --no-cpu --no-harvest --no-aperiodic --no-configs --steps 20 --warmup 5 --reps 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  value', d['value'], d['value_min'], d['value_max'], 'sync alone', d['roofline']['kernels_ms'].get('sync_kernel'), 'overlapped', d['roofline']['kernels_ms_overlapped'].get('sync_kernel'), d['frames_acquired'], d['verified']['ok'])"; done
