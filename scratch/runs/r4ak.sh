#!/bin/bash
# the pipeline path's distance to the direct one: pool sizes?  (the pipeline paths close one result generation per step of three rounds, the direct path one per slab)
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu --no-harvest --no-aperiodic --no-configs --steps 20 --warmup 5 --reps 3 --serial-steps 2 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  value', d['value'], d['ms_per_step'], d['roofline'].get('kernels_ms_overlapped'), d['verified']['ok'])"; }
echo "== direct"; run
echo "== direct, pools for three slabs, one generation per slab"; BENCH_MAXFRAMES_X=3 run
echo "== direct, pools for three slabs, one generation per step"; BENCH_MAXFRAMES_X=3 BENCH_DISCARD_PER_STEP=1 run
echo "== pipeline"; run --pipeline
