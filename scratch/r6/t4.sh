#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for n in 1 2 3; do BENCH_RESAMP_STREAMS=$n python scratch/r6/leg.py 64ch_m256_qam16_resamp 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); v=list(d.values())[0]; print('streams $n', v['value'], v['value_min'], v['value_max'], v['ms_per_step'], v['kernels_ms_overlapped'], v['verified'])"; done
