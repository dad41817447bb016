#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4n
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r4n/tests.txt 2>&1
tail -30 gpurun_out/r4n/tests.txt | cut -c1-300
