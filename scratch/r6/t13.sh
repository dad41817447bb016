#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for cfg in "" "serial=1"; do for leg in 8ch 8ch_v27 8ch_long_pushes; do echo "== $leg [$cfg]"; LEG_CFG=$cfg python scratch/r6/leg.py $leg 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); v=list(d.values())[0]; print(v['value'], v['value_min'], v['value_max'], v['ms_per_step'], v['kernels_ms_overlapped'])"; done; done
