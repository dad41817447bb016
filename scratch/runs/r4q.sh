#!/bin/bash
cd $GRAFT_REPO_ROOT
j() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  value', d['value'], d['value_min'], d['value_max'], d['verified']['ok'], d['roofline']['kernels_ms'], d['roofline']['kernels_ms_overlapped'])"; }
B="--no-cpu --no-harvest --no-aperiodic --no-configs --steps 20 --warmup 5 --reps 3"
echo "== direct"; python bench.py $B 2>/dev/null | tail -1 | j
echo "== direct, defer 16384"; python bench.py $B --defer 16384 2>/dev/null | tail -1 | j
echo "== pipeline C, defer 0"; python bench.py --pipeline $B --defer 0 2>/dev/null | tail -1 | j
echo "== pipeline C, defer 4096"; python bench.py --pipeline $B --defer 4096 2>/dev/null | tail -1 | j
echo "== pipeline C"; python bench.py --pipeline $B 2>/dev/null | tail -1 | j
