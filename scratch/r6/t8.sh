#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_liquid_interop.py tests/test_gpu_soak.py -m gpu -x -q -s -rs 2>&1 | grep -v "^$" | tail -25
