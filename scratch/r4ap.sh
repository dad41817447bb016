#!/bin/bash
# does the direct path gain from fewer streams too?  (payload workers + decoder on the acquisition's stream / on the channelizer's)
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu --no-harvest --no-aperiodic --no-configs --steps 20 --warmup 5 --reps 3 --serial-steps 2 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  value', d['value'], d['value_min'], d['value_max'], d['verified']['ok'])"; }
echo "== default (channelizer | acquisition | workers)"; run
echo "== workers on the acquisition's stream"; MCRX_X_WORK_ON_SCOUT=1 run
echo "== workers on the channelizer's stream"; MCRX_X_WORK_ON_MAIN=1 run
echo "== default"; run
echo "== duplex (transmit pipeline on one stream)"; python bench_duplex.py --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  duplex', d['value'], d['ms_per_step'])"
