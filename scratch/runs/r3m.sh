#!/bin/bash
for env in "X=1" "MCRX_SEEK_BURST=0" "MCRX_ACQ_MODE=2" "MCRX_ACQ_MODE=2 MCRX_SEEK_BURST=0" "MCRX_ACQ_MODE=1" "MCRX_NO_SPEC=1" "MCRX_NO_SPEC=1 MCRX_SEEK_BURST=0"; do
echo "== $env"; env $env python -m pytest tests/test_gpu_stream.py -q -x -k policy 2>&1 | grep -E "AssertionError:|passed|failed" | head -3
done
