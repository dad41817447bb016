#!/bin/bash
# randomised soak on the round's last library: new seeds, 20 iterations each (scratch/soak.py)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r6_soak2
for seed in 101 102 103 104 105 106; do timeout 1200 python scratch/soak.py $seed 20 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r6_soak2/soak.jsonl; done
for seed in 111 112 113; do SOAK_SNR_LO=1 SOAK_SNR_HI=9 timeout 1200 python scratch/soak.py $seed 20 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r6_soak2/soak_low_snr.jsonl; done
cat gpurun_out/r6_soak2/*.jsonl
