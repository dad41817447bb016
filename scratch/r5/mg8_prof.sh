#!/bin/bash
# one rank of the 8-GPU job as a continuous stream (scratch/mg8_stream.py): per-kernel durations
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/mg8_prof; rm -rf $O; mkdir -p $O
(cd $R; rocprofv3 --kernel-trace --stats --output-format csv -d $O -o m -- python scratch/mg8_stream.py 16 > $O/m.log 2> $O/m.err)
tail -1 $O/m.log | cut -c1-300
python3 - $O/m_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    print(f"  {r['Name'][:64]:64s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1000:8.1f} us  {r['Percentage']}%")
PY
