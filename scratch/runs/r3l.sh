#!/bin/bash
O=$PWD/gpurun_out/r3l; mkdir -p $O
R=$PWD
MCTX_R8=1 python -m pytest tests/test_gpu_tx.py tests/test_gpu_txshard.py -q -x 2>&1 | tail -3 | tee $O/tests.log
python -m pytest tests/test_gpu_stream.py -q -x -k policy 2>&1 | grep -B30 "Error" | head -60 > $O/policy.log
cd /tmp && export TMPDIR=/tmp
for n in 512 256; do
MCTX_R8=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$n -o tx -- python $R/scratch/tx_time.py $n > $O/tx$n.log 2>&1
grep -E "synth_kernel" $O/prof$n/tx_kernel_stats.csv | cut -c1-150
done
cd $R; for r8 in 0 1; do MCTX_R8=$r8 python bench_duplex.py --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c90-200; done
MCTX_SYNTH=0 python bench_duplex.py --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c90-200
