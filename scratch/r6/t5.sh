#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for rep in 1 2; do for lib in product side32; do
  if [ $lib = product ]; then unset MCRX_LIB; else export MCRX_LIB=$GRAFT_REPO_ROOT/scratch/r6/exp/libs/libexp_$lib.so; fi
  python scratch/r6/leg.py 64ch_m256_qam16_resamp 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); v=list(d.values())[0]; print('$lib', v['value'], v['value_min'], v['value_max'], v['verified']['ok'])"; done; done
