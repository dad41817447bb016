#!/bin/bash
O=gpurun_out/r3d; mkdir -p $O
L=$PWD/scratch/libs
for lib in $PWD/scratch/libmcrx_r2ref.so $L/libmcrx_D.so $L/libmcrx_D1.so; do
  for cfg in "72 300 65536" "32 300 65536"; do
    MCRX_LIB=$lib python scratch/probe/coresident2.py $cfg 2>&1 | grep -v amdgpu.ids | tee -a $O/probe2.log
  done
done
