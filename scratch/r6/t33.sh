#!/bin/bash
# wide symbols without the two tail launches: parity at M >= 256, then the 64-channel legs
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -k "256 or wide or soak or config2 or config3 or baseline or straddl or oversize or defer or low_snr" 2>&1 | tail -3
for i in 1 2; do
  for leg in 64ch_m256_qam16_resamp 64ch_m256_qam16 64ch_m256_qpsk; do
    python scratch/r6/leg.py $leg 6 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=list(d)[0]; v=d[k]; print(k, v['value'], v['value_min'], v['value_max'], v.get('kernels_ms_overlapped'), v['verified']['ok'])"
  done
done
