#!/bin/bash
# build libmcrx_hip.so variants that differ in how channelizer.hip is compiled: scratch/libs/libmcrx_<name>.so
cd /root/repo/liquid-usrp_amd/csrc
make -s -j8 2>&1 | grep -E "error|Error" 
IT="-Xarch_device -mllvm=-misched=gcn-iterative-max-occupancy-experimental"
MR="-Xarch_device -mllvm=-misched=gcn-iterative-minreg"
mk() { name=$1; shift
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-division-by-zero "$@" -c channelizer.hip -o /tmp/chan_$name.o 2>/dev/null &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../scratch/libs/libmcrx_$name.so /tmp/chan_$name.o mcrx_hip.o msresamp.o txgen.o pfb2.o ofdmsync_p0.o ofdmsync_p1.o ofdmsync_p2.o ofdmsync_p3.o && echo built $name; }
rm -f ../../scratch/libs/*.so
mk R8 -DCH_ROUND_1024=8 -DCH_RECOMPUTE_IDX=0 $IT &
mk R4a -DCH_RECOMPUTE_IDX=1 $MR &
mk R4b -DCH_RECOMPUTE_IDX=1 $IT &
mk R4c -DCH_RECOMPUTE_IDX=0 $MR &
mk R4d -DCH_RECOMPUTE_IDX=1 -DCH_WAVES_1024=3 $IT &
wait
ls ../../scratch/libs
