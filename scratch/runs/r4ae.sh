#!/bin/bash
# kernel durations of the 8-channel leg at long pushes (FRAMES per channel and push)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4ae; mkdir -p $O
for f in ${FR:-400}; do
FRAMES=$f STEPS=4 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o f$f -- python $R/scratch/cfg_probe.py 8ch > $O/f$f.log 2>&1
echo "== $f"; python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/f${f}_kernel_stats.csv")))
for r in rows[:14]: print("  %-60s n %5s avg %10.1f us"%(r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
