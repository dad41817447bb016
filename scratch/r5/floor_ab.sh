#!/bin/bash
# grid floors of the list-driven launches that are normally empty (payload_lean_rest_kernel 256 -> 16, decode_general_kernel 128 -> 8):
# release library against a build with the small floors, alternating on one box
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
for i in 1 2 3; do for t in release floors; do
  lib=$R/liquid-usrp_amd/lib/libmcrx_hip.so; [ $t = floors ] && lib=$R/liquid-usrp_amd/lib/libmcrx_hip_floor.so
  MCRX_LIB=$lib python bench.py --no-cpu --no-variants --no-harvest 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$t:', d['value'], 'aperiodic', d['value_aperiodic'], {k:v.get('value') for k,v in d['configs'].items()})"
done; done
for t in release floors; do
  lib=$R/liquid-usrp_amd/lib/libmcrx_hip.so; [ $t = floors ] && lib=$R/liquid-usrp_amd/lib/libmcrx_hip_floor.so
  MCRX_LIB=$lib python bench.py --pipeline --no-cpu --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipeline $t:', d['value'])"
done
