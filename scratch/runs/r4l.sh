#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4l
timeout 1500 python -X faulthandler -m pytest tests -m gpu -x -v > gpurun_out/r4l/tests.txt 2>&1
grep -n "PASSED\|FAILED\|ERROR" gpurun_out/r4l/tests.txt | tail -5
grep -n "Fatal\|fault\|File \"/" gpurun_out/r4l/tests.txt | head -30
