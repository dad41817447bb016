#!/bin/bash
# configs[1] latency: what bounds the device period?  devel build
cd $GRAFT_REPO_ROOT
export MCRX_LIB=$GRAFT_REPO_ROOT/liquid-usrp_amd/lib/libmcrx_hip_devel.so
run() { echo "== $*"; env "$@" python scratch/r6/host_bound.py 2>&1 | grep -v amdgpu.ids | grep "60 pushes"; }
run X=1
run MCRX_NO_EVT=1
run MCRX_NO_EVT=1 MCRX_FREE_RUN=1
run GPU_MAX_HW_QUEUES=12
run GPU_MAX_HW_QUEUES=16
run GPU_MAX_HW_QUEUES=16 MCRX_NO_EVT=1
run MCRX_NO_PRIO=1
