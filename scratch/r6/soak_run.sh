#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r6_soak
for seed in 81 82 83 84; do timeout 900 python scratch/soak.py $seed 10 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r6_soak/soak.jsonl; done
for seed in 91 92; do SOAK_SNR_LO=1 SOAK_SNR_HI=9 timeout 900 python scratch/soak.py $seed 10 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r6_soak/soak_low_snr.jsonl; done
cat gpurun_out/r6_soak/*.jsonl
