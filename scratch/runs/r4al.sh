#!/bin/bash
# harvest against the closed generations a poll leaves for later (MCRX_POLL_KEEP: 1 = two pushes in flight)
cd $GRAFT_REPO_ROOT
for k in 1 2 3 1 2; do
echo "== MCRX_POLL_KEEP=$k"; MCRX_POLL_KEEP=$k python bench.py --no-cpu --no-aperiodic --no-configs --serial-steps 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ', d['value'], d['value_with_harvest'], round(d['value_with_harvest']/d['value'],3), d['value_with_harvest_detail']['frames_delivered'])"
done
