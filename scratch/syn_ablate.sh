#!/bin/bash
# development builds of the library with phases of the fused synthesis kernel removed: liquid-usrp_amd/lib/ab/libmcrx_ab<bits>.so
cd /root/repo/liquid-usrp_amd/csrc; mkdir -p ../lib/ab
OTHER="channelizer.o mcrx_hip.o msresamp.o pfb2.o pipeline.o ofdmsync_p0.o ofdmsync_p1.o ofdmsync_p2.o ofdmsync_p3.o"
for b in "$@"; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DSYN_ABLATE=$b -c txgen.hip -o /tmp/txgen_ab$b.o 2>&1 | grep -E "error" ;
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../lib/ab/libmcrx_ab$b.so /tmp/txgen_ab$b.o $OTHER -ldl ) &
done
wait; ls -la ../lib/ab
