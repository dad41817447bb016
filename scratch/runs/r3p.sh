#!/bin/bash
O=gpurun_out/r3p; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_soak.py -q -x -k "conv or soak" 2>&1 | tail -4 | tee $O/tests.log
python scratch/configs_r2.py C2_conv_v27 2>&1 | grep -v amdgpu | tail -3 | tee $O/c2conv.log
python scratch/configs_r2.py C2 C3 2>&1 | grep -v amdgpu | tail -4 | tee -a $O/c2conv.log
