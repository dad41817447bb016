#!/bin/bash
# RB=16 channelizer experiment: parity under the variant library, then alternating timings against the product
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
L=$GRAFT_REPO_ROOT/scratch/r6/exp/libs/libexp_rb16.so
MCRX_LIB=$L timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "channelizer or any_channel or full_chain_bit or split" 2>&1 | tail -5
for i in 1 2 3; do
  python scratch/r6/exp/time_chan.py 0 2>&1 | grep -v amdgpu.ids
  MCRX_LIB=$L python scratch/r6/exp/time_chan.py 0 2>&1 | grep -v amdgpu.ids
done
