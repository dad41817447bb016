#!/bin/bash
# after the timing switch: the few-channel legs and the configs[2] / headline legs
cd $GRAFT_REPO_ROOT
for leg in 8ch 8ch_v27 8ch_long_pushes 64ch_m256_qam16_resamp 512ch; do
  python scratch/r6/leg.py $leg 6 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=list(d)[0]; v=d[k]; print(k, v['value'], v['value_min'], v['value_max'], v.get('kernels_ms_overlapped'), v['verified']['ok'])"
done
