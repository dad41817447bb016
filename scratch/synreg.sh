#!/bin/bash
# register / spill / instruction-mix report of the fused synthesis kernel: scratch/synreg.sh K R IN [extra hipcc flags]
K=${1:-1024}; R=${2:-8}; IN=${3:-1}; shift 3
mkdir -p /tmp/isa; cd /root/repo/liquid-usrp_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Rpass-analysis=kernel-resource-usage -save-temps=obj "$@" -c txgen.hip -o /tmp/isa/txgen.o 2>&1 | grep -E "error|synth_kernelILi${K}ELi${R}ELi${IN}E" -A10 | egrep "error|SGPRs:|VGPRs|Scratch" | sed 's/.*remark: *//; s/\[-Rpass.*//' | tr '\n' ' '; echo
cd /tmp/isa && awk "/^_ZN4mcrx3syn12synth_kernelILi${K}ELi${R}ELi${IN}EEEvNS_11TxSynthArgsEj:/,/s_endpgm/" txgen-hip-amdgcn-amd-amdhsa-gfx950.s > syn.s
echo "lines $(wc -l < syn.s) pk_fma $(grep -c v_pk_fma syn.s) v_mov $(grep -cE '^\s+v_mov_b32' syn.s) pk_mov $(grep -c v_pk_mov syn.s) scratch $(grep -c scratch_ syn.s) ds $(grep -cE '^\s+ds_' syn.s) valu $(grep -cE '^\s+v_' syn.s) salu $(grep -cE '^\s+s_' syn.s) waitcnt $(grep -c s_waitcnt syn.s)"
