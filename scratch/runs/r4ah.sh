#!/bin/bash
# pipeline paths with the history copy ahead of the wait for the exchange (default build) against libmcrx_win2.so; direct path beside them
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "pipeline or shard or launch or refapp_through or sharded" 2>&1 | tail -3
run() { python bench.py --no-cpu --no-harvest --no-aperiodic --no-configs --steps 20 --warmup 5 --reps 3 --serial-steps 2 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  value', d['value'], d['value_min'], d['value_max'], d['verified']['ok'])"; }
for v in win2 default win2 default; do
  if [ $v = default ]; then unset MCRX_LIB; else export MCRX_LIB=$GRAFT_REPO_ROOT/scratch/libs/libmcrx_$v.so; fi
  echo "== $v --pipeline (C-ABI)"; run --pipeline
done
unset MCRX_LIB
echo "== default --pipeline --exchange torch"; run --pipeline --exchange torch
echo "== default direct"; run
