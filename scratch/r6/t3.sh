#!/bin/bash
# configs[2] baseline: leg value + kernel trace
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r6_t3
python scratch/r6/leg.py 64ch_m256_qam16_resamp 2>/dev/null | tee gpurun_out/r6_t3/leg.json | python -c "
import json,sys; d=json.load(sys.stdin); v=list(d.values())[0]; print(v['value'], v['ms_per_step'], v['kernels_ms_overlapped'], v['frames_acquired'], v['verified'])"
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r6_t3/prof -o c2 -- python $GRAFT_REPO_ROOT/scratch/r6/leg.py 64ch_m256_qam16_resamp 6 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/r6_t3/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cp {} gpurun_out/r6_t3/kernel_stats.csv; head -25 {}'
find gpurun_out/r6_t3/prof -name "*.db" -delete; find gpurun_out/r6_t3/prof -size +5M -delete
