#!/bin/bash
# FETCH_SIZE calibration (scratch/fetchcal.hip): prints requested bytes per kernel and the counter beside them
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/fetchcal
mkdir -p $O
$R/scratch/fetchcal > $O/requested.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O -o fetch -- $R/scratch/fetchcal > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O -o write -- $R/scratch/fetchcal > $O/write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats -- $R/scratch/fetchcal > $O/stats.log 2>&1
python3 - <<PY
import csv, glob
for f in sorted(glob.glob("$O/**/*counter_collection.csv", recursive=True)):
    print(f)
    for r in csv.DictReader(open(f)):
        print("  %-40s %-12s %s" % (r.get("Kernel_Name", "")[:40], r.get("Counter_Name"), r.get("Counter_Value")))
for f in sorted(glob.glob("$O/**/*kernel_stats.csv", recursive=True)):
    print(open(f).read())
PY
