# kernel breakdown of the full-duplex bench (rocprofv3 stats), sub-slab size from $1
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s5/dup_${2:-a}; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o dup -- python $R/bench_duplex.py --sub-blocks ${1:-262144} --steps 10 --warmup 3 > $O/dup.log 2>&1
sort -t, -k3 -n -r $O/*dup_kernel_stats.csv | head -14 | cut -c1-200
tail -1 $O/dup.log | cut -c1-200
