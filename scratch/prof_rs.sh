#!/bin/bash
# usage: prof_rs.sh <tag>  -- per workload: rocprofv3 kernel stats, then FETCH_SIZE and WRITE_SIZE in their own passes
tag=${1:-rs1}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
mkdir -p $O
for w in ${WORKLOADS:-rs0.5 rs0.8 rs0.37 rs2.0 pfb1024 pfb128 pfb16}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ${w}_stats -- python $R/scratch/rs_prof.py $w > $O/${w}_stats.log 2>&1
  grep -E "msresamp|firpfbch2 M" $O/${w}_stats.log
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O -o ${w}_pmc_$c -- python $R/scratch/rs_prof.py $w > $O/${w}_pmc_$c.log 2>&1
  done
done
rm -f $O/*agent_info.csv $O/*domain_stats.csv
ls $O | wc -l
