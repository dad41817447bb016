#!/bin/bash
# full duplex on one GPU against the pipelines' stream counts (transmit: MCTX_PIPE_STREAMS, receive: MCRX_PIPE_STREAMS)
cd $GRAFT_REPO_ROOT
for n in "1 2" "1 1" "1 2" "1 1" "1 3"; do set -- $n; echo "== transmit $1 receive $2"; MCTX_PIPE_STREAMS=$1 MCRX_PIPE_STREAMS=$2 python bench_duplex.py --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  duplex', d['value'], d['ms_per_step'], d['verified']['ok'])"; done
