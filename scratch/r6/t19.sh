#!/bin/bash
# channelizer variants: parity under the variant library, then alternating timings against the product
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in "$@"; do
L=$GRAFT_REPO_ROOT/scratch/r6/exp/libs/libexp_$v.so
#MCRX_LIB=$L timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py -x -q -k "channelizer or any_channel or full_chain_bit or split or config4 or 512" 2>&1 | tail -3
for i in 1 2 3; do
  python scratch/r6/exp/time_chan.py 0 2>&1 | grep -v amdgpu.ids
  MCRX_LIB=$L python scratch/r6/exp/time_chan.py 0 2>&1 | grep -v amdgpu.ids
done
done
