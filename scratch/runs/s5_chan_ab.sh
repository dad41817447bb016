# A/B of the channelizer (and the synthesis kernel) after the rotation-in-the-add change: parity first, then kernel times
cd $GRAFT_REPO_ROOT; O=gpurun_out/s5/chanab; mkdir -p $O; rm -f $O/*
(timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tx.py tests/test_gpu_baseline_shapes.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -4) | tee $O/tests.txt
for lib in liquid-usrp_amd/lib/ab/libmcrx_base.so liquid-usrp_amd/lib/libmcrx_hip.so liquid-usrp_amd/lib/ab/libmcrx_base.so liquid-usrp_amd/lib/libmcrx_hip.so; do
  echo "$lib: $(MCRX_LIB=$PWD/$lib timeout 120 python scratch/chan_time.py 0 2>/dev/null | tail -1)"
done | tee $O/chan.txt
cd /tmp; export TMPDIR=/tmp
for lib in ab/libmcrx_base.so libmcrx_hip.so; do
  MCRX_LIB=$GRAFT_REPO_ROOT/liquid-usrp_amd/lib/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O -o tx_$(basename $lib .so) -- python $GRAFT_REPO_ROOT/scratch/tx_time.py 512 > /dev/null 2>&1
  echo "$lib $(grep -h synth_kernel $GRAFT_REPO_ROOT/$O/*tx_$(basename $lib .so)_kernel_stats.csv | cut -d, -f4-6)"
done
