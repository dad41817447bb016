#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for t in product pf2 product pf2; do
  if [ $t = product ]; then unset MCRX_LIB; else export MCRX_LIB=$GRAFT_REPO_ROOT/scratch/r6/exp/libs/libexp_$t.so; fi
  python scratch/r6/exp/time_chan.py 0 2>&1 | grep -v amdgpu.ids
done
export MCRX_LIB=$GRAFT_REPO_ROOT/scratch/r6/exp/libs/libexp_pf2.so
python -m pytest tests/test_gpu_parity.py -x -q -k "channelizer_matches or any_channel or full_chain_bit" 2>&1 | tail -2
