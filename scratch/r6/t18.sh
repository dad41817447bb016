#!/bin/bash
# configs[1] timelines (8ch, 8ch_v27): kernel stats + steady-state timeline
cd $GRAFT_REPO_ROOT
for leg in 8ch 8ch_v27; do
  bash scratch/r6/prof_leg.sh $leg c1_$leg
  f=$(find gpurun_out/r6_prof_c1_$leg -name "leg_kernel_trace.csv" | head -1)
  python scratch/r6/timeline.py $f channelizer 2
  cat gpurun_out/r6_prof_c1_$leg/leg.json | tail -1 | cut -c1-400
done
