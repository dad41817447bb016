#!/bin/bash
# usage: prof_r2.sh <tag>   -- rocprofv3 kernel stats + separate PMC passes of the bench with the SERIAL receiver (every kernel alone),
# plus a kernel trace of the pipelined receiver (the timeline the overlap is read from)
tag=${1:-v1}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
mkdir -p $O
B="python $R/bench.py --serial --no-cpu --no-harvest --no-aperiodic --steps 6 --warmup 2 --serial-steps 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats -- python $R/bench.py --serial --no-cpu --no-harvest --no-aperiodic --steps 40 --warmup 5 --serial-steps 1 > $O/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O -o pmc_$n -- $B > $O/pmc_$n.log 2>&1
done
rocprofv3 --kernel-trace --output-format csv -d $O -o pipelined -- python $R/bench.py --no-cpu --no-harvest --no-aperiodic --steps 10 --warmup 3 --serial-steps 1 > $O/pipelined.log 2>&1
python $R/bench.py 2>/dev/null | tail -1 > $O/bench.json
ls $O
