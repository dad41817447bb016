# whole GPU suite + transmitter timing
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s5/all_${1:-a}; mkdir -p $O
cd $R
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12) > $O/tests.txt
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o tx512 -- python $R/scratch/tx_time.py 512 > $O/tx512.log 2>&1
cat $O/tests.txt; grep -h "synth_kernel\|txsym" $O/*tx512_kernel_stats.csv | cut -c1-160
