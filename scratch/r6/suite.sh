#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r6_suite
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/r6_suite/gpu_tests.log
