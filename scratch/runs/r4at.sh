#!/bin/bash
# payload workers with two symbol windows requested ahead (libmcrx_pf2.so) against one (default build)
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu --no-harvest --no-aperiodic --no-configs --steps 20 --warmup 5 --reps 3 --serial-steps 6 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  value', d['value'], d['value_min'], d['value_max'], d['roofline']['kernels_ms'], d['verified']['ok'])"; }
for v in default pf2 default pf2; do
  if [ $v = default ]; then unset MCRX_LIB; else export MCRX_LIB=$GRAFT_REPO_ROOT/scratch/libs/libmcrx_$v.so; fi
  echo "== $v"; run
  echo "   8ch:"; python scratch/cfg_probe.py 8ch 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ', d['value'], d['kernels_ms_overlapped'], d['verified']['ok'])"
done
