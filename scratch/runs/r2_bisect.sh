#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_bisect; mkdir -p $O
for v in "X=1" "MCRX_SCOUT_ROUNDS=1" "MCRX_SERIAL=1" "MCRX_NO_SPEC=1" "MCRX_NO_PRIO=1"; do
  echo "== $v"
  env $v timeout 600 python -m pytest tests/test_gpu_soak.py -q -x -k "11" 2>&1 | grep -E "passed|failed|iter . rep . " | awk '{print $1,$2,$3,$4,$5,$6,$7}' | sort | uniq -c | head -12
done
