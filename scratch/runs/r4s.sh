#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4s
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r4s/tests.txt 2>&1; tail -3 gpurun_out/r4s/tests.txt
MCRX_DEBUG=8 python bench.py --no-cpu --no-aperiodic --no-configs --steps 20 --warmup 5 --reps 3 2>gpurun_out/r4s/err.txt | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  value', d['value'], 'harvest', d['value_with_harvest'], round(d['value_with_harvest']/d['value'],3), 'full', d['value_with_full_harvest'], d['value_with_harvest_detail'])"
grep "mcrx bulk path" gpurun_out/r4s/err.txt | cut -c1-300
