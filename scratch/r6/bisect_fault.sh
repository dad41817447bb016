#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for leg in 8ch 8ch_long_pushes 8ch_v27 512ch_m48 64ch_m256_qam16_resamp 512ch_pfb2_front_end; do
  echo "== $leg"; timeout 300 python scratch/r6/leg.py $leg 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tail -2
done
echo "== headline only"; timeout 600 python bench.py --no-configs --no-cpu 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tail -2
