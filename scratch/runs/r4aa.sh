#!/bin/bash
# the synchronizers' history copied right behind the channelizer by a small-workgroup kernel (default) against hipMemcpyAsync in front of the next one (MCRX_NO_PREFILL=1)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "stream or soak or launch or refapp or pipeline" 2>&1 | tail -3
run() { python bench.py --no-cpu --no-harvest --no-aperiodic --no-configs --steps 20 --warmup 5 --reps 3 --serial-steps 8 $EXTRA 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  value', d['value'], d['value_min'], d['value_max'], d['verified']['ok'])"; }
for v in old new old new; do
  if [ $v = new ]; then unset MCRX_NO_PREFILL; else export MCRX_NO_PREFILL=1; fi
  echo "== $v"; run
done
unset MCRX_NO_PREFILL
