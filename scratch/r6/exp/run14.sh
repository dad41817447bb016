#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for t in "$@"; do MCRX_LIB=$GRAFT_REPO_ROOT/scratch/r6/exp/libs/libexp_$t.so python scratch/r6/exp/time_chan.py 0 2>&1 | grep -v amdgpu.ids; done
