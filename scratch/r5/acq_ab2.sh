#!/bin/bash
# pipelined headline + ragged leg of several builds, alternating, in one call: scratch/r5/acq_ab2.sh <tag[:scout_build]> ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
for spec in "$@"; do
  tag=${spec%%:*}; sb=${spec##*:}; [ "$sb" = "$spec" ] && sb=0
  if [ "$tag" = default ]; then unset MCRX_LIB; else export MCRX_LIB=$R/liquid-usrp_amd/lib/libmcrx_hip_$tag.so; fi
  python $R/bench.py --steps 12 --warmup 4 --reps 3 --no-cpu --no-configs --no-harvest --no-variants --scout-build $sb 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.readline());print('$spec', 'value', d['value'], d['value_min'], d['value_max'], 'aperiodic', d.get('value_aperiodic'), 'acq alone', d['roofline']['kernels_ms']['sync_kernel'])"
done; done
