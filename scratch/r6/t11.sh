#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
export MCRX_LIB=$GRAFT_REPO_ROOT/scratch/r6/exp/libs/libexp_devel.so
for v in "" "MCRX_NSEG=32" "MCRX_NSEG=8" "MCRX_NSEG=64" "MCRX_SEG_FRAMES=1" "MCRX_ACQ_MODE=5"; do
  echo "== $v"; env $v python scratch/r6/leg.py 64ch_m256_qam16_resamp 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); v=list(d.values())[0]; print(v['value'], v['value_min'], v['value_max'], v['kernels_ms_overlapped'], v['verified']['ok'])"; done
