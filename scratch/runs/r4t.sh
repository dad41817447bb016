#!/bin/bash
cd $GRAFT_REPO_ROOT
for k in 1 2 0 2; do echo "== MCRX_POLL_KEEP=$k"; MCRX_POLL_KEEP=$k python bench.py --no-cpu --no-aperiodic --no-configs --steps 20 --warmup 5 --reps 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  value', d['value'], 'harvest', d['value_with_harvest'], round(d['value_with_harvest']/d['value'],3), d['value_with_harvest_detail']['frames_delivered'], d['value_with_harvest_detail']['expected_frames'])"; done
