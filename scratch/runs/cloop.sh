#!/bin/bash
# dump the interior (EDGE = false) round loop of the K = 1024 channelizer to /tmp/loop.s and summarise it
cd /root/repo/liquid-usrp_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-division-by-zero "$@" -S --cuda-device-only -c channelizer.hip -o /tmp/cz.s 2>/dev/null
awk '/^_ZN4mcrx18channelizer_kernelILi1024ELi2ELi512ELi[0-9]EEEvNS_8ChanArgsE:/{p=1} p{print} /\.end_amdhsa_kernel/{if(p){exit}}' /tmp/cz.s > /tmp/cz1024.s
# the first inner loop that contains s_barrier
python3 - <<'PY'
import re
L=open('/tmp/cz1024.s').read().split('\n')
labels={m.group(1):i for i,l in enumerate(L) for m in [re.match(r'^(\.LBB\d+_\d+):',l)] if m}
best=None
for i,l in enumerate(L):
    m=re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)',l)
    if m and m.group(1) in labels and labels[m.group(1)]<i:
        body=L[labels[m.group(1)]:i+1]
        if sum('s_barrier' in x for x in body)>=3:
            best=body; break
open('/tmp/loop.s','w').write('\n'.join(best))
def cnt(p): return sum(1 for x in best if re.search(p,x))
print("loop lines",len(best),"VALU",cnt(r'^\s*v_'),"v_mov",cnt(r'^\s*v_mov'),"pk",cnt(r'^\s*v_pk_'),"ds",cnt(r'^\s*ds_'),"global",cnt(r'^\s*global_'),"scratch",cnt(r'^\s*scratch_'),"salu",cnt(r'^\s*s_(?!waitcnt|barrier|nop)'))
PY
grep -n "global_load\|global_store\|scratch_" /tmp/loop.s | cut -c1-100
