#!/bin/bash
# full duplex on one GPU against the transmit pipeline's stream count
cd $GRAFT_REPO_ROOT
for n in 3 2 1 3 2; do echo "== MCTX_PIPE_STREAMS=$n"; MCTX_PIPE_STREAMS=$n python bench_duplex.py --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  duplex', d['value'], d['ms_per_step'], d.get('verified'))"; done
