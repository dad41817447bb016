#!/bin/bash
# the pushes' true timelines (development build, MCRX_EVT_DUMP; scratch/r6/evt_timeline.py) of the three receiver shapes -> profiles/r6_evt_timelines.txt
cd $GRAFT_REPO_ROOT
export MCRX_LIB=$GRAFT_REPO_ROOT/liquid-usrp_amd/lib/libmcrx_hip_devel.so
out=gpurun_out/r6_evt_timelines.txt
echo "# five timed stages of the last 14 of 72 pushes: start [us from the first event] + duration [us]; 'period' = from one push's acquisition to the next's" > $out
echo "# == configs[1]: 8 channels, M = 64, QPSK + Hamming(12,8), 100 frames per channel and push" >> $out
MCRX_EVT_DUMP=1 python scratch/r6/host_bound.py 8 100 6 2>&1 | python scratch/r6/evt_timeline.py >> $out
echo "# == configs[1] with the K = 7 code (its trellis on the fourth stream: 'decode' ends there)" >> $out
MCRX_EVT_DUMP=1 python scratch/r6/host_bound.py 8 100 11 2>&1 | python scratch/r6/evt_timeline.py >> $out
echo "# == the headline's receiver: 512 channels, M = 64, 16 frames per channel and push" >> $out
MCRX_EVT_DUMP=1 python scratch/r6/host_bound.py 512 16 6 2>&1 | python scratch/r6/evt_timeline.py >> $out
echo "# == configs[2]'s receiver (no resampler in front): 64 channels, M = 256, QAM16 + Golay, 32 frames per channel and push" >> $out
HB_M=256 HB_CP=32 HB_MOD=27 MCRX_EVT_DUMP=1 python scratch/r6/host_bound.py 64 32 7 2>&1 | python scratch/r6/evt_timeline.py >> $out
