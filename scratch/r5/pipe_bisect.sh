#!/bin/bash
# bench.py --pipeline with features of this round's library switched off one at a time (devel build: MCRX_NO_SEEKST, MCRX_NO_VITSCRATCH,
# MCRX_NO_VITALLOC), against the release library and round 4's tree, on one box
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
run() { python bench.py --pipeline --no-cpu --steps 20 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"; }
for i in 1 2; do
  echo "release            $(run)"
  echo "devel              $(MCRX_LIB=$R/liquid-usrp_amd/lib/libmcrx_hip_devel.so run)"
  echo "devel no seekst    $(MCRX_LIB=$R/liquid-usrp_amd/lib/libmcrx_hip_devel.so MCRX_NO_SEEKST=1 run)"
  echo "devel no vit alloc $(MCRX_LIB=$R/liquid-usrp_amd/lib/libmcrx_hip_devel.so MCRX_NO_VITALLOC=1 run)"
  echo "devel acq mode 3   $(MCRX_LIB=$R/liquid-usrp_amd/lib/libmcrx_hip_devel.so MCRX_ACQ_MODE=3 run)"
  echo "devel lean build 0 $(MCRX_LIB=$R/liquid-usrp_amd/lib/libmcrx_hip_devel.so MCRX_LEAN_BUILD=0 run)"
  echo "round 4 tree       $(cd scratch/r5/oldtree && python bench.py --pipeline --no-cpu --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])")"
done
