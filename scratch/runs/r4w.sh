#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu --no-harvest --no-aperiodic --no-configs --steps 20 --warmup 5 --reps 3 --serial-steps 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  value', d['value'], d['value_min'], d['value_max'], d['roofline']['kernels_ms'], d['verified']['ok'])"; }
for v in prev default prev default; do
  if [ $v = default ]; then unset MCRX_LIB; else export MCRX_LIB=$GRAFT_REPO_ROOT/scratch/libs/libmcrx_$v.so; fi
  echo "== $v"; run
done
unset MCRX_LIB

