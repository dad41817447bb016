#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in default synnt default synnt; do
  if [ $v = default ]; then unset MCRX_LIB; else export MCRX_LIB=$GRAFT_REPO_ROOT/scratch/libs/libmcrx_$v.so; fi
  echo "== $v"; python bench_duplex.py --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  duplex', d['value'], d['ms_per_step'], d['verified']['ok'])"
  python scratch/tx_time.py 512 2>/dev/null | tail -2 | cut -c1-300
done
