#!/bin/bash
# steady-state timeline of the headline loop
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6_headline_trace; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o hl -- python $R/bench.py --no-cpu --no-harvest --no-aperiodic --no-configs --no-variants --steps 6 --warmup 3 --reps 1 > $O/hl.log 2>&1
f=$(find $O -name "hl_kernel_trace.csv" | head -1)
python $R/scratch/r6/timeline.py $f channelizer 3
tail -1 $O/hl.log | cut -c1-200
