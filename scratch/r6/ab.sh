#!/bin/bash
# alternating A/B of a leg between the product library and variant libraries: scratch/r6/ab.sh <leg> <reps> <tag>...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
leg=$1; reps=$2; shift; shift
for rep in $(seq 1 $reps); do for lib in product "$@"; do
  if [ $lib = product ]; then unset MCRX_LIB; else export MCRX_LIB=$GRAFT_REPO_ROOT/scratch/r6/exp/libs/libexp_$lib.so; fi
  python scratch/r6/leg.py $leg 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); v=list(d.values())[0]; print('$lib', v['value'], v['value_min'], v['value_max'], v['kernels_ms_overlapped'], v['verified']['ok'])"; done; done
