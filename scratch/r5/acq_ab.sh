#!/bin/bash
# serial-receiver kernel durations of several builds in one call: scratch/r5/acq_ab.sh <tag> <tag> ...   (default build = "default")
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for tag in "$@"; do
  if [ "$tag" = default ]; then unset MCRX_LIB; else export MCRX_LIB=$R/liquid-usrp_amd/lib/libmcrx_hip_$tag.so; fi
  O=$R/gpurun_out/acq_ab_$tag; rm -rf $O; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o s -- python $R/bench.py --serial --steps 4 --warmup 2 --reps 1 --no-cpu --no-configs --no-harvest --no-aperiodic --no-variants > $O/s.json 2> $O/s.err
  echo "== $tag serial: $(python -c "import json;d=json.load(open('$O/s.json'));print(d['roofline']['kernels_ms'])")"
  grep -E "acq_lean|sync_walk|sync_spec" $O/s_kernel_stats.csv | cut -d, -f1-7 | cut -c1-150
  python $R/bench.py --steps 10 --warmup 4 --reps 3 --no-cpu --no-configs --no-harvest --no-aperiodic --no-variants 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.readline());print('   pipelined value', d['value'], d['value_min'], d['value_max'])"
done
