#!/bin/bash
# per-kernel durations of the serial receiver on the headline stream (every kernel alone on one stream): scratch/r5/serial_stats.sh [bench.py flags]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/serial_stats; rm -rf $O; mkdir -p $O
(cd $R; rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python bench.py --serial --no-cpu --no-harvest --no-aperiodic --no-configs --no-variants --steps 6 --warmup 2 --reps 1 "$@" > $O/s.json 2> $O/s.err)
python3 - $O/s_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print(f"  {r['Name'][:64]:64s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1000:8.1f} us  {r['Percentage']}%")
PY
