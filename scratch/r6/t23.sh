#!/bin/bash
# configs[1]: the lean segment waves (cfg.scout_build = 2: chains of one frame) against the default
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for cfg in "" "scout_build=2"; do
    for leg in 8ch 8ch_v27 8ch_long_pushes; do
      LEG_CFG=$cfg python scratch/r6/leg.py $leg 6 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=list(d)[0]; v=d[k]; print('$cfg', k, v['value'], v['value_min'], v['value_max'], v.get('kernels_ms_overlapped'), v['verified']['ok'])"
    done
  done
done
