#!/bin/bash
# configs[1]: a smaller footprint for the K = 16 channelizer (larger slabs -> fewer workgroups) beside the latency-bound stages
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for cfg in "" "slab_blocks=160" "slab_blocks=320" "slab_blocks=640"; do
      LEG_CFG=$cfg python scratch/r6/leg.py 8ch 6 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=list(d)[0]; v=d[k]; print('$cfg', k, v['value'], v['value_min'], v['value_max'], v.get('kernels_ms_overlapped'), v['verified']['ok'])"
  done
done
