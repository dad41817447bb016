#!/bin/bash
# the pushes' true timeline (development build, MCRX_EVT_DUMP): scratch/r6/t23.sh <channels> <frames> <fec1>
cd $GRAFT_REPO_ROOT
export MCRX_LIB=$GRAFT_REPO_ROOT/liquid-usrp_amd/lib/libmcrx_hip_devel.so
MCRX_EVT_DUMP=1 python scratch/r6/host_bound.py $1 $2 $3 2>&1 | python scratch/r6/evt_timeline.py
