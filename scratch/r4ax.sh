#!/bin/bash
# harvest with the generation's counters in one 32-byte block (default build) against libmcrx_eo.so; stream / discard tests first
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "stream or soak or refapp or overflow or pool or poll or harvest" 2>&1 | tail -3
for v in eo default eo default; do
  if [ $v = default ]; then unset MCRX_LIB; else export MCRX_LIB=$GRAFT_REPO_ROOT/scratch/libs/libmcrx_$v.so; fi
  echo "== $v"; MCRX_DEBUG=8 python bench.py --no-cpu --no-aperiodic --no-configs --serial-steps 2 2>gpurun_out/h.err | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ', d['value'], d['value_with_harvest'], round(d['value_with_harvest']/d['value'],3))"; grep "bulk path" gpurun_out/h.err | sed -n 2p | cut -c1-330
done
