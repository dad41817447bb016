#!/bin/bash
# configs[1] (8 channels, 100 frames per push): per-kernel durations of the pipelined receiver (scratch/r5/host_bound.py under rocprofv3)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/small_prof; rm -rf $O; mkdir -p $O
(cd $R; FRAMES=${FRAMES:-100} rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python scratch/r5/host_bound.py > $O/s.log 2> $O/s.err)
tail -1 $O/s.log
python3 - $O/s_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(f"  {r['Name'][:64]:64s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1000:8.1f} us  {r['Percentage']}%")
PY
