# synthesis-kernel iteration: TX parity tests, rocprofv3 stats of scratch/tx_time.py (N = 512 and 256), duplex bench
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s5/tx_${1:-a}; mkdir -p $O
cd $R
(timeout 900 python -m pytest tests/test_gpu_tx.py tests/test_gpu_txshard.py tests/test_gpu_baseline_shapes.py tests/test_gpu_pipeline.py -m gpu -x -q 2>&1 | tail -8) > $O/tests.txt
cd /tmp
for n in 512 256; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o tx$n -- python $R/scratch/tx_time.py $n > $O/tx$n.log 2>&1
done
cd $R
python bench_duplex.py --sub-blocks 262144 > $O/duplex.json 2> $O/duplex.err
python bench_duplex.py --sub-blocks 524288 > $O/duplex2.json 2> $O/duplex2.err
cat $O/tests.txt; for n in 512 256; do grep -h "synth_kernel\|txsym" $O/*tx${n}_kernel_stats.csv | cut -c1-160; done; cut -c1-300 $O/duplex.json;  cut -c1-300 $O/duplex2.json
