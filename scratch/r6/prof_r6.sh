#!/bin/bash
# usage: prof_r6.sh <tag>   -- rocprofv3 kernel stats of the driver's command + separate PMC passes of a short run of it (round 6)
tag=${1:-final}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$tag
mkdir -p $O
B="python $R/bench.py --no-cpu --no-harvest --no-aperiodic --no-configs --steps 5 --warmup 2 --reps 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats -- python $R/bench.py --no-cpu --no-harvest --no-aperiodic --no-configs --steps 20 --warmup 5 > $O/stats.log 2>&1
# ... and of the serial receiver (--serial: every kernel alone on one stream): the averages that bench.py's roofline.kernels_ms must agree with
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o serial -- python $R/bench.py --serial --no-cpu --no-harvest --no-aperiodic --no-configs --steps 6 --warmup 2 --reps 1 > $O/serial.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O -o pmc_$n -- $B > $O/pmc_$n.log 2>&1
done
cd $R
python bench.py --steps 20 --warmup 5 2>$O/bench.err | tail -1 > $O/bench.json
python bench.py --pipeline --no-cpu --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_pipeline.json
python bench_duplex.py 2>/dev/null | tail -1 > $O/duplex.json
echo "== harvest leg host phases (MCRX_DEBUG=8)"
MCRX_DEBUG=8 python bench.py --no-cpu --no-aperiodic --no-configs --steps 5 --warmup 2 --reps 1 2>&1 | grep -E "mcrx bulk path|value_with_harvest" | cut -c1-400 | tail -4
ls $O | head -30
