#!/bin/bash
# round 6, first GPU contact: the folded front end (parity, speed, kernel trace)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r6_t1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "oversampled or channelizer_matches or any_channel_count" 2>&1 | tail -15 > gpurun_out/r6_t1/tests.log
cat gpurun_out/r6_t1/tests.log
for leg in 512ch_pfb2_front_end 512ch 512ch_pfb2_chain; do timeout 600 python scratch/r6/leg.py $leg > gpurun_out/r6_t1/$leg.json 2> gpurun_out/r6_t1/$leg.err; tail -c 1500 gpurun_out/r6_t1/$leg.json; tail -3 gpurun_out/r6_t1/$leg.err; done
