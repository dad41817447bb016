#!/bin/bash
# txsym_kernel with its loads hoisted / requested ahead (default build) against libmcrx_r8tx.so
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4ac; mkdir -p $O
cd $R; timeout 900 python -m pytest tests -m gpu -x -q -k "tx or refapp or ofdm" 2>&1 | tail -3; cd /tmp
for v in r8tx default; do
  if [ $v = default ]; then unset MCRX_LIB; else export MCRX_LIB=$R/scratch/libs/libmcrx_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o $v -- python $R/bench.py --no-cpu --no-harvest --no-aperiodic --no-configs --steps 2 --warmup 1 --reps 1 --serial-steps 1 > $O/$v.log 2>&1
  echo "== $v"; grep -E "txsym|synth_kernel" $O/${v}_kernel_stats.csv
done
for v in r8tx default r8tx default; do
  if [ $v = default ]; then unset MCRX_LIB; else export MCRX_LIB=$R/scratch/libs/libmcrx_$v.so; fi
  echo "== $v"; python $R/bench_duplex.py --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  duplex', d['value'], d['ms_per_step'])"
done
