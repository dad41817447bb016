#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for sb in 0 1056 1584 400 528; do echo "== slab_blocks $sb"; python bench.py --no-configs --no-cpu --no-harvest --no-aperiodic --no-variants --steps 30 --warmup 5 --reps 3 --slab-blocks $sb 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['value_min'], d['value_max'], d['roofline']['kernels_ms'].get('channelizer_kernel'), d['verified']['ok'])"; done
