#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2_trace2; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o trace -- python $R/bench.py --no-cpu --no-harvest --steps 10 --warmup 3 --serial-steps 1 > $O/trace.log 2>&1
python - <<PY
import csv, glob
rows = [r for r in csv.DictReader(open("$O/trace_kernel_trace.csv")) if "mcrx::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ch = [i for i, r in enumerate(rows) if "channelizer_kernel" in r["Kernel_Name"]]
i0, i1 = ch[20], ch[24]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i1]:
    n = r["Kernel_Name"].split("(")[0].replace("void mcrx::","").replace("mcrx::","")[:30]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print("%-32s %9.1f %9.1f %8.1f" % (n, s / 1e3, e / 1e3, (e - s) / 1e3))
print("per push", (int(rows[i1]["Start_Timestamp"]) - t0) / 4e3)
PY
