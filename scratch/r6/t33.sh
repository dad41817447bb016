#!/bin/bash
# the general decoder's 64-register build for its normally-empty launches: parity (K = 7, hard decisions, surprises), then the legs
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -k "v27 or viterbi or conv or surprises or general or hard or soak or payload_soft or codes or every_modem" 2>&1 | tail -3
for i in 1 2; do
  for leg in 64ch_m256_qam16_resamp 8ch_v27 8ch 512ch; do
    python scratch/r6/leg.py $leg 6 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=list(d)[0]; v=d[k]; print(k, v['value'], v['value_min'], v['value_max'], v.get('kernels_ms_overlapped'), v['verified']['ok'])"
  done
done
