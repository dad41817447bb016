#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_third; mkdir -p $O
for s in 11 12 13 14; do python scratch/soak_dbg.py $s 2>&1 | grep -E "bad" | awk '{s+=$4} END {print "seed bad total", s}'; done
MCRX_SERIAL=1 python scratch/soak_dbg.py 11 2>&1 | grep -E "bad" | awk '{s+=$4} END {print "serial bad total", s}'
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_refapp.py > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
