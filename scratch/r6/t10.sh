#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for m in 0 1; do echo "== conv_scratch=$m"; LEG_CFG=conv_scratch=$m timeout 300 python scratch/r6/leg.py 8ch_v27 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tail -2; done
