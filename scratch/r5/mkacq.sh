#!/bin/bash
# build a variant of the library that differs in the lean segment-wave kernel only: scratch/r5/mkacq.sh <tag> <extra hipcc flags...>
tag=$1; shift
cd /root/repo/liquid-usrp_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Rpass-analysis=kernel-resource-usage -DSY_PART=4 "$@" -c ofdmsync.hip -o /tmp/p4_$tag.o 2>&1 | grep -E "remark: .*(VGPRs:|ScratchSize|LDS Size)" | sed 's/.*remark: *//; s/\[-Rpass.*//' | paste - - - | head -2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../lib/libmcrx_hip_$tag.so channelizer.o mcrx_hip.o msresamp.o txgen.o pfb2.o pipeline.o ofdmsync_p0.o ofdmsync_p1.o ofdmsync_p2.o ofdmsync_p3.o /tmp/p4_$tag.o -ldl
ls -la ../lib/libmcrx_hip_$tag.so
