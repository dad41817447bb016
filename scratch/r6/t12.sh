#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_refapp.py tests/test_gpu_pipeline.py tests/test_gpu_parity.py -x -q -k "oversampled or scratch_is_allocated or pipeline" 2>&1 | tail -6
