#!/bin/bash
# rocprofv3 kernel stats of one leg: scratch/r6/prof_leg.sh <leg> <tag>   (LEG_CFG=serial=1 for every kernel alone)
leg=$1; tag=$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r6_prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o leg -- python $R/scratch/r6/leg.py $leg 6 1 > $out/leg.json 2> $out/leg.err
f=$(find $out -name "leg_kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:26]:
    print("%-64s calls %5s avg %9.1f us total %9.1f ms" % (r['Name'].replace('void mcrx::','').replace('mcrx::','')[:64], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
find $out -name "leg_kernel_trace.csv" -size +20M -delete
