#!/bin/bash
# usage: prof.sh <tag>   -- rocprofv3 kernel stats + separate PMC passes of the default bench
tag=${1:-final}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
mkdir -p $O
B="python $R/bench.py --no-cpu --steps 5 --warmup 2"
# kernel durations: the default run (200 steps, clocks settled), so that the averages are comparable with bench.py's own
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats -- python $R/bench.py --no-cpu > $O/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O -o pmc_$n -- $B > $O/pmc_$n.log 2>&1
done
python $R/bench.py 2>/dev/null | tail -1 > $O/bench.json
ls $O
