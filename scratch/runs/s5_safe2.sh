# guarded check of an acquisition change: a small parity test first, then the 8-channel stream, each under its own timeout
cd $GRAFT_REPO_ROOT; O=gpurun_out/s5/safe2; mkdir -p $O; rm -f $O/*
(timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5) > $O/t1.txt; echo "rc $? parity" >> $O/progress.txt
tail -3 $O/t1.txt
if grep -q passed $O/t1.txt && ! grep -q failed $O/t1.txt; then
  MCRX_LEAN_BUILD=1 timeout 200 python scratch/configs_r2.py C2 2> $O/cfg.err | tail -1 | cut -c1-400; echo "rc $? C2" >> $O/progress.txt
fi
cat $O/progress.txt; tail -3 $O/cfg.err 2>/dev/null
