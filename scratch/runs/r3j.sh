#!/bin/bash
O=gpurun_out/r3j; mkdir -p $O; rm -f $O/probe.log
for cus in 32 64 96; do
echo "contiguous walk_cus $cus" | tee -a $O/probe.log
MCRX_WALK_CONTIG=1 MCRX_WALK_CUS=$cus MCRX_ACQ_MODE=2 python scratch/aper_probe.py 0 2>&1 | grep -v "amdgpu.ids\|policy" | tee -a $O/probe.log
done
MCRX_WALK_CUS=32 MCRX_ACQ_MODE=2 python scratch/aper_probe.py 0 2>&1 | grep -v "amdgpu.ids\|policy" | tee -a $O/probe.log
