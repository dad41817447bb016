#!/bin/bash
# decode_kernel's phases in cycles (SY_PROFILE build of part 0, MCRX_DEBUG=2)
cd $GRAFT_REPO_ROOT
export MCRX_LIB=$GRAFT_REPO_ROOT/scratch/libs/libmcrx_prof0.so
MCRX_DEBUG=2 python bench.py --no-cpu --no-harvest --no-aperiodic --no-configs --steps 3 --warmup 1 --reps 1 --serial-steps 2 2>&1 | grep "\[prof\] decode" | sort | uniq -c | sort -rn | head -20
