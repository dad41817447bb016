#!/bin/bash
# alternating A/B on the development build: wide symbols with (MCRX_WIDE_TAILS=1, the form until now) and without their two tail launches
cd $GRAFT_REPO_ROOT
export MCRX_LIB=$GRAFT_REPO_ROOT/liquid-usrp_amd/lib/libmcrx_hip_devel.so
for i in 1 2 3 4; do
  for t in "" 1; do
    MCRX_WIDE_TAILS=$t python scratch/r6/leg.py 64ch_m256_qam16_resamp 6 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=list(d)[0]; v=d[k]; print('tails=[$t]', k, v['value'], v['value_min'], v['value_max'], v['verified']['ok'])"
  done
done
