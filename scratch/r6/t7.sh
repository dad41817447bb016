#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for leg in 64ch_m256_qpsk 64ch_m256_qam16 64ch_m256_qam64; do LEG_CFG=serial=1 python scratch/r6/leg.py $leg 4 1 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); v=list(d.values())[0]; print('$leg', v['value'], v['samples_per_step'], v['kernels_ms_overlapped'], v['verified'])"; done
