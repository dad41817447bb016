#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/stats_q
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python $R/bench.py --no-cpu --steps 5 --warmup 2 > $O/log 2>&1
grep "mcrx::" $O/s_kernel_stats.csv | awk -F, '{n=split($0,a,"\""); print a[2], $(NF-6), $(NF-4)}' | cut -c1-60,100-
python - <<PY
import csv
for r in csv.DictReader(open("$O/s_kernel_stats.csv")):
    if "mcrx::" in r["Name"]: print("%-60s calls %3s avg_us %9.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
