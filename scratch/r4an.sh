#!/bin/bash
# the pipelines on one GPU with their stages on 2 streams (default since this run) / 3 (rounds 2-3) / 1; GPU tests of the sharded paths first
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "pipeline or shard or launch or sharded or txshard" 2>&1 | tail -3
run() { python bench.py --no-cpu --no-harvest --no-aperiodic --no-configs --steps 20 --warmup 5 --reps 3 --serial-steps 2 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  value', d['value'], d['value_min'], d['value_max'], d['verified']['ok'])"; }
echo "== direct"; run
for n in 2 3 2 3 1; do echo "== C-ABI pipeline, MCRX_PIPE_STREAMS=$n"; MCRX_PIPE_STREAMS=$n run --pipeline; done
echo "== torch pipeline (two streams)"; run --pipeline --exchange torch
echo "== duplex"; python bench_duplex.py --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  duplex', d['value'], d['ms_per_step'])"
