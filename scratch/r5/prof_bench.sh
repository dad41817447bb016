#!/bin/bash
# rocprofv3 kernel stats of a short bench run, pipelined and serial: scratch/r5/prof_bench.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o pipe -- python $R/bench.py --steps 10 --warmup 3 --reps 2 --no-cpu --no-configs --no-harvest --no-aperiodic "$@" > $out/pipe.json 2> $out/pipe.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o serial -- python $R/bench.py --serial --steps 6 --warmup 2 --reps 1 --no-cpu --no-configs --no-harvest --no-aperiodic "$@" > $out/serial.json 2> $out/serial.err
for n in pipe serial; do
  f=$(find $out -name "${n}_kernel_stats.csv" | head -1)
  echo "== $n: $f"; head -22 "$f" | cut -d, -f1-7 | cut -c1-200
done
