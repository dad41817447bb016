#!/bin/bash
# a rank of the 8-GPU job as a continuous stream (scratch/mg8_stream.py 16): kernel durations, and against the build before the slot windows
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4au; mkdir -p $O
for v in ${VARIANTS:-default}; do
  if [ $v = default ]; then unset MCRX_LIB; else export MCRX_LIB=$R/scratch/libs/libmcrx_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o $v -- python $R/scratch/mg8_stream.py 16 > $O/$v.log 2>&1
  echo "== $v"; tail -1 $O/$v.log | cut -c1-400
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/${v}_kernel_stats.csv")))
for r in rows[:9]: print("  %-60s n %5s avg %10.1f us"%(r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
done
