#!/bin/bash
# randomised soak on the round's last library: new seeds, 20 iterations each (scratch/soak.py)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r6_soak3
for seed in 121 122 123 124 125 126; do timeout 1200 python scratch/soak.py $seed 20 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r6_soak3/soak.jsonl; done
for seed in 131 132 133; do SOAK_SNR_LO=1 SOAK_SNR_HI=9 timeout 1200 python scratch/soak.py $seed 20 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r6_soak3/soak_low_snr.jsonl; done
cat gpurun_out/r6_soak3/*.jsonl
