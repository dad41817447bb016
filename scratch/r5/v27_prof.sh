#!/bin/bash
# per-kernel durations of configs[1] with the K = 7 code (scratch/r5/v27_probe.py) -> gpurun_out/prof_v27_<tag>/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_v27_${1:-a}; rm -rf $O; mkdir -p $O
(cd $R; python scratch/r5/v27_probe.py > $O/plain.log 2>&1; tail -2 $O/plain.log)
(cd $R; rocprofv3 --kernel-trace --stats --output-format csv -d $O -o v -- python scratch/r5/v27_probe.py > $O/v.log 2> $O/v.err)
tail -2 $O/v.log
python3 - $O/v_kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1000:8.1f} us  {r['Percentage']}%")
PY
