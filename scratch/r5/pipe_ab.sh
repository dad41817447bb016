#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for t in new old; do
  d=$R; [ $t = old ] && d=$R/scratch/r5/oldtree
  O=$R/gpurun_out/pipe_ab_$t; rm -rf $O; mkdir -p $O
  (cd $d; rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python bench.py --pipeline --no-cpu --steps 20 --warmup 5 > $O/s.json 2> $O/s.err)
  echo "== $t $(python -c "import json;print(json.load(open('$O/s.json'))['value'])")"
  head -16 $O/s_kernel_stats.csv | awk -F, '{printf "%-60s calls %6s avg %9.1f us tot %8.1f ms\n", substr($1,1,60), $2, $4/1000, $3/1e6}'
done
