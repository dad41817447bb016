#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_shapes.py -x -q -k "msresamp or resampled or config" 2>&1 | tail -4
python scratch/rs_prof.py rs0.5 rs0.8 rs0.37 rs2.0 2>&1 | grep msresamp
[ "$1" = prof ] && bash scratch/prof_rs.sh r6rs 2>&1 | tail -3
