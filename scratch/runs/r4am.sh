#!/bin/bash
# harvest with and without the payload arena's device-to-host copy (scratch build): is it the copy that slows the GPU side?
cd $GRAFT_REPO_ROOT
for v in default nod2h default nod2h; do
  if [ $v = default ]; then unset MCRX_LIB; else export MCRX_LIB=$GRAFT_REPO_ROOT/scratch/libs/libmcrx_$v.so; fi
  echo "== $v"; MCRX_DEBUG=8 python bench.py --no-cpu --no-aperiodic --no-configs --serial-steps 2 2>gpurun_out/h.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ', d['value'], d['value_with_harvest'], round(d['value_with_harvest']/d['value'],3))"; grep "bulk path" gpurun_out/h.err | sed -n 2p
done
