#!/bin/bash
# bench.py --pipeline with the serial receiver (every kernel alone), this round's library against round 4's: which kernel is slower by itself?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for t in new old; do
  d=$R; [ $t = old ] && d=$R/scratch/r5/oldtree
  O=$R/gpurun_out/pipe_serial_$t; rm -rf $O; mkdir -p $O
  (cd $d; rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python bench.py --pipeline --serial --no-cpu --steps 8 --warmup 3 --reps 1 > $O/s.json 2> $O/s.err)
  echo "== $t $(python -c "import json;print(json.load(open('$O/s.json'))['value'])" 2>/dev/null || tail -2 $O/s.err)"
  python3 - $O/s_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1000:8.1f} us tot {float(r['TotalDurationNs'])/1e6:8.1f} ms")
PY
done
