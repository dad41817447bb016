#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
VARIANTS="new" bash scratch/r4g.sh
echo "== single launch (MCRX_ACQ_MODE=1)"
MCRX_ACQ_MODE=1 python bench.py --no-cpu --no-harvest --no-configs --steps 20 --warmup 5 --reps 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  value', d['value'], 'aper', d.get('value_aperiodic'), 'sync alone', d['roofline']['kernels_ms'].get('sync_kernel'), d['frames_acquired'], d['verified']['ok'])"
MCRX_ACQ_MODE=1 python scratch/mg8_stream.py 2 2>&1 | tail -1 | cut -c1-400
for sf in 2 3 6; do echo "== MCRX_SEG_FRAMES=$sf"; MCRX_SEG_FRAMES=$sf python bench.py --no-cpu --no-harvest --no-configs --steps 20 --warmup 5 --reps 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  value', d['value'], 'aper', d.get('value_aperiodic'), 'sync alone', d['roofline']['kernels_ms'].get('sync_kernel'), d['frames_acquired'], d['verified']['ok'])"; done
