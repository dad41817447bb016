#!/bin/bash
# radix-8 LDS stages in the synthesis bank (K = 1024, 512, 128) and in the channelizer's K = 512 / 128 plans (default build) against libmcrx_fz.so
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for v in fz default fz default; do
  if [ $v = default ]; then unset MCRX_LIB; else export MCRX_LIB=$GRAFT_REPO_ROOT/scratch/libs/libmcrx_$v.so; fi
  echo "== $v"; python bench_duplex.py --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  duplex', d['value'], {k: d[k] for k in d if 'ms' in k or 'kernel' in k})"
done
unset MCRX_LIB
