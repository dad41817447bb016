#!/bin/bash
# variant builds of the channelizer only, linked with the product's other objects: scratch/r6/exp/libs/libexp_<tag>.so
cd /root/repo/scratch/r6/exp; mkdir -p libs obj
CS=/root/repo/liquid-usrp_amd/csrc
build() { tag=$1; shift
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I$CS "$@" -c ${SRC:-channelizer_exp.hip} -o obj/ch_$tag.o 2>obj/ch_$tag.log || { echo FAIL $tag; tail -5 obj/ch_$tag.log; return; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o libs/libexp_$tag.so obj/ch_$tag.o $CS/mcrx_hip.o $CS/msresamp.o $CS/txgen.o $CS/pfb2.o $CS/pipeline.o $CS/ofdmsync_p0.o $CS/ofdmsync_p1.o $CS/ofdmsync_p2.o $CS/ofdmsync_p3.o $CS/ofdmsync_p4.o -ldl && echo built $tag; }
for a in "$@"; do tag=${a%%:*}; flags=${a#*:}; build $tag $flags & done; wait
