#!/bin/bash
# build libmcrx_hip.so variants that differ in how the acquisition kernels (ofdmsync.hip part 3) are compiled: scratch/libs/libmcrx_<name>.so
cd /root/repo/liquid-usrp_amd/csrc
make -s -j8 2>&1 | grep -E "rror"
mkdir -p ../../scratch/libs
mk() { name=$1; shift
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DSY_PART=3 "$@" -c ofdmsync.hip -o /tmp/p3_$name.o 2>/dev/null &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../scratch/libs/libmcrx_$name.so /tmp/p3_$name.o channelizer.o mcrx_hip.o msresamp.o txgen.o pfb2.o pipeline.o ofdmsync_p0.o ofdmsync_p1.o ofdmsync_p2.o -ldl && echo built $name; }
rm -f ../../scratch/libs/*.so
B="-mllvm -vgpr-regalloc=basic"
mk b168 $B -DSY_SEG_BURST=1 &
mk n168 $B -DSY_SEG_BURST=0 &
mk b256 -DSY_SEG_BURST=1 -DSY_ACQ_WAVES_OVERRIDE=2 &
mk n256 -DSY_SEG_BURST=0 -DSY_ACQ_WAVES_OVERRIDE=2 &
mk b512 -DSY_SEG_BURST=1 -DSY_ACQ_WAVES_OVERRIDE=1 &
wait
ls -la ../../scratch/libs
