# guarded runs: every command under its own timeout, progress written as it goes
cd $GRAFT_REPO_ROOT; O=gpurun_out/s5/safe; mkdir -p $O; rm -f $O/*
echo "tests" >> $O/progress.txt
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6) > $O/tests.txt; echo "rc $? tests" >> $O/progress.txt
for lb in 1 0; do
  echo "start lb=$lb" >> $O/progress.txt
  MCRX_LEAN_BUILD=$lb timeout 300 python scratch/configs_r2.py C2 C2_conv_v27 C3 2> $O/cfg_$lb.err | tail -1 > $O/cfg_$lb.json; echo "rc $? lb=$lb cfg" >> $O/progress.txt
done
cat $O/tests.txt $O/progress.txt; python - <<PY
import json
for lb in (1, 0):
    d = json.loads(open("$O/cfg_%d.json" % lb).read())
    for k, v in d.items(): print("lean_build", lb, k, v["Msamples_per_s"], v["ms_per_step"], v["verified"])
PY
