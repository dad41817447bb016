# synthesis-kernel phase ablation: rocprofv3 average of synth_kernel for every development build under liquid-usrp_amd/lib/ab/
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s5/ab; mkdir -p $O
for so in $R/liquid-usrp_amd/lib/libmcrx_hip.so $R/liquid-usrp_amd/lib/ab/*.so; do
  t=$(basename $so .so)
  MCRX_LIB=$so rocprofv3 --kernel-trace --stats --output-format csv -d $O -o $t -- python $R/scratch/tx_time.py 512 > $O/$t.log 2>&1
  echo "$t $(grep -h synth_kernel $O/*${t}_kernel_stats.csv | cut -d, -f4)"
done | tee $O/summary.txt
