#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_duplex
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats -- python $R/bench_duplex.py --steps 10 --warmup 3 > $O/stats.log 2>&1
tail -2 $O/stats.log | cut -c1-300
python - <<PY
import csv
for r in csv.DictReader(open("$O/stats_kernel_stats.csv")):
    print("%-70s calls %5s avg_us %9.1f  pct %5s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
