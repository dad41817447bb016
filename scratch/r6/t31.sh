#!/bin/bash
# the K = 7 decoder's launch on the fourth stream (gen_side): parity for the code, then the 8-channel legs
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "v27 or viterbi or conv or k7 or K7 or surprises or soak" 2>&1 | tail -3
for i in 1 2; do
  for leg in 8ch_v27 8ch; do
    python scratch/r6/leg.py $leg 6 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=list(d)[0]; v=d[k]; print(k, v['value'], v['value_min'], v['value_max'], v.get('kernels_ms_overlapped'), v['verified']['ok'])"
  done
done
