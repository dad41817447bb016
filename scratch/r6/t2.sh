#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r6_t2
python scratch/r6/split_dbg.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6_t2/split.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "oversampled or channelizer_matches or any_channel_count" 2>&1 | tail -8 | tee gpurun_out/r6_t2/tests.log
for leg in 512ch_pfb2_front_end 512ch; do timeout 600 python scratch/r6/leg.py $leg > gpurun_out/r6_t2/$leg.json 2> gpurun_out/r6_t2/$leg.err; python -c "
import json,sys; d=json.load(open('gpurun_out/r6_t2/$leg.json')); v=list(d.values())[0]; print('$leg', v['value'], v.get('roofline',{}).get('kernel_ms_alone'), v['kernels_ms_overlapped'], v['verified']['ok'])"; done
