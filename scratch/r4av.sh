#!/bin/bash
# Hamming(12,8) soft decision with the zero-syndrome early-out (default build) against libmcrx_rank.so (the build before); GPU suite + two soak seeds first
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for s in 51 52; do timeout 280 python scratch/soak.py $s 10 2>&1 | tail -1; done
run() { python bench.py --no-cpu --no-harvest --no-configs --steps 20 --warmup 5 --reps 3 --serial-steps 6 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  value', d['value'], d['value_min'], d['value_max'], d['roofline']['kernels_ms'], d['verified']['ok'], 'aperiodic', d['value_aperiodic'])"; }
for v in rank default rank default; do
  if [ $v = default ]; then unset MCRX_LIB; else export MCRX_LIB=$GRAFT_REPO_ROOT/scratch/libs/libmcrx_$v.so; fi
  echo "== $v"; run
done
