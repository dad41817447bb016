#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
timeout 600 python bench.py --gpus 2 --rehearse-on-one-gpu --steps 4 --warmup 2 --reps 1 --no-cpu 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('rehearsal 2 ranks:', d['value'], d['n_gpus'], d['verified'], d.get('exchange',{}).get('path'))"
timeout 600 python bench.py --pipeline --steps 10 --warmup 3 --reps 2 --no-cpu 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pipeline 1 gpu:', d['value'], d['verified']['ok'], d.get('exchange',{}).get('path'))"
