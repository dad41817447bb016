#!/bin/bash
# the folded oversampled front end (cfg.front_end = 1) at 512 channels: rocprofv3 kernel stats of the pipelined and the serial receiver,
# FETCH_SIZE / WRITE_SIZE / instruction counters in passes of their own -> gpurun_out/prof_r6pfb2/, condensed by scratch/r6/summarize_pfb2.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r6pfb2
mkdir -p $O
L="python $R/scratch/r6/leg.py 512ch_pfb2_front_end"
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o pipe -- $L 6 2 > $O/pipe.json 2> $O/pipe.err
LEG_CFG=serial=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o serial -- $L 4 1 > $O/serial.json 2> $O/serial.err
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  n=$(echo $c | cut -d' ' -f1)
  LEG_CFG=serial=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O -o pmc_$n -- $L 2 1 > $O/pmc_$n.log 2>&1
done
find $O -name "*kernel_trace.csv" -size +8M -delete
rm -f $O/*agent_info.csv $O/*domain_stats.csv
ls $O | wc -l
