#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4r; mkdir -p $O
B="--no-cpu --no-harvest --no-aperiodic --no-configs --steps 6 --warmup 4 --reps 1 --serial-steps 1"
rocprofv3 --kernel-trace --output-format csv -d $O -o direct -- python $R/bench.py $B > $O/direct.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O -o pipe -- python $R/bench.py --pipeline $B > $O/pipe.log 2>&1
python3 - <<PY
import csv
for tag in ("direct","pipe"):
    rows=[r for r in csv.DictReader(open("$O/%s_kernel_trace.csv"%tag))]
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    t_end=int(rows[-1]["End_Timestamp"])
    # last 3 ms of the run
    sel=[r for r in rows if int(r["Start_Timestamp"])>t_end-4000000]
    t0=int(sel[0]["Start_Timestamp"])
    print("==",tag, len(sel),"kernels in the last 4 ms")
    for r in sel[:70]:
        n=r["Kernel_Name"].split("(")[0].replace("void mcrx::","").replace("mcrx::","")[:28]
        print("  %-28s q%-3s start %8.1f us  dur %7.1f us" % (n, r.get("Queue_Id","?"), (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
PY
