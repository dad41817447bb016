#!/bin/bash
# register / spill report of the K = 1024 channelizer build: scratch/creg.sh [extra hipcc flags]
cd /root/repo/liquid-usrp_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-division-by-zero -Rpass-analysis=kernel-resource-usage "$@" -c channelizer.hip -o /tmp/cz.o 2>&1 | grep -A9 "channelizer_kernelILi1024ELi2ELi512ELi" | egrep "SGPRs:|VGPRs|Scratch|Occup" | sed 's/.*remark: *//; s/\[-Rpass.*//' | tr '\n' ' '; echo
