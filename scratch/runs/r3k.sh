#!/bin/bash
O=gpurun_out/r3k; mkdir -p $O
MCRX_LIB=$PWD/scratch/libs/libmcrx_prof.so MCRX_DEBUG=2 MCRX_ACQ_MODE=2 MCRX_CHAIN_WALK=0 python scratch/aper_probe.py 1 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/prof.log
