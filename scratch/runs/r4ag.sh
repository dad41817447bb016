#!/bin/bash
# windowed slot headers + tight hop loop (default build) against libmcrx_r8tx.so: GPU suite, the 8-channel leg over push lengths, the headline
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for f in 100 400 800; do echo "== frames/ch/slab $f"; FRAMES=$f python scratch/cfg_probe.py 8ch 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ', d['value'], d['ms_per_step'], d['kernels_ms_overlapped'], d['verified']['ok'], d['frames_acquired'])"; done
run() { python bench.py --no-cpu --no-harvest --no-aperiodic --no-configs --steps 20 --warmup 5 --reps 3 --serial-steps 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  value', d['value'], d['value_min'], d['value_max'], d['roofline']['kernels_ms'], d['verified']['ok'])"; }
for v in r8tx default r8tx default; do
  if [ $v = default ]; then unset MCRX_LIB; else export MCRX_LIB=$GRAFT_REPO_ROOT/scratch/libs/libmcrx_$v.so; fi
  echo "== $v"; run
done
unset MCRX_LIB
