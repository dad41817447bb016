#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_trace; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_refapp.py > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/$O -o trace -- python $R/bench.py --no-cpu --no-harvest --steps 6 --warmup 3 --serial-steps 2 > $R/$O/trace.log 2>&1
ls -la $R/$O | head
python - <<PY
import csv, glob
f = glob.glob("$R/$O/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last ~40 dispatches of the timed region
t0 = int(rows[-120]["Start_Timestamp"])
for r in rows[-120:-60]:
    n = r["Kernel_Name"].split("(")[0].replace("void mcrx::","").replace("mcrx::","")[:34]
    print("%-36s start %9.1f us  dur %7.1f us  q %s" % (n, (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Queue_Id", "")))
PY
