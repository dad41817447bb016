#!/bin/bash
# decode_kernel held to 5 / 6 waves per SIMD (96 / 80 registers, small spills) against 4 (120 registers): libmcrx_dk5.so, libmcrx_dk6.so, libmcrx_eo.so
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu --no-harvest --no-aperiodic --no-configs --steps 20 --warmup 5 --reps 3 --serial-steps 6 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  value', d['value'], d['value_min'], d['value_max'], d['roofline']['kernels_ms'], d['verified']['ok'])"; }
for v in eo dk5 dk6 eo dk5; do
  export MCRX_LIB=$GRAFT_REPO_ROOT/scratch/libs/libmcrx_$v.so
  echo "== $v"; run
done
