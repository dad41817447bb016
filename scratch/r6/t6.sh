#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "wide or full_chain_bit_exact" 2>&1 | tail -12
timeout 600 python -m pytest tests/test_gpu_baseline_shapes.py -x -q 2>&1 | tail -4
python scratch/r6/leg.py 64ch_m256_qam16_resamp 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); v=list(d.values())[0]; print(v['value'], v['value_min'], v['value_max'], v['ms_per_step'], v['kernels_ms_overlapped'], v['verified'])"
