#!/bin/bash
# pipelined timeline of four pushes in the steady state + how long which kernels ran together
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_trace; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -o trace -- python $R/bench.py --no-cpu --no-harvest --no-aperiodic --steps 10 --warmup 3 --serial-steps 1 > $O/trace.log 2>&1
python - <<PY
import csv, glob, collections
rows = [r for r in csv.DictReader(open("$O/trace_kernel_trace.csv")) if "mcrx::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ch = [i for i, r in enumerate(rows) if "channelizer_kernel" in r["Kernel_Name"]]
i0, i1 = ch[20], ch[24]
t0 = int(rows[i0]["Start_Timestamp"]); t1 = int(rows[i1]["Start_Timestamp"])
def nm(r): return r["Kernel_Name"].split("(")[0].replace("void mcrx::","").replace("mcrx::","").split("<")[0][:24]
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e <= t0 or s >= t1: continue
    n = nm(r)
    print("%-26s %9.1f %9.1f %8.1f" % (n, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3))
    ev.append((max(s, t0), 1, n)); ev.append((min(e, t1), -1, n))
print("per push", (t1 - t0) / 4e3)
ev.sort()
act = collections.Counter(); tot = collections.Counter(); last = t0
for t, d, n in ev:
    if t > last:
        key = "+".join(sorted(k for k, v in act.items() if v > 0)) or "(idle)"
        tot[key] += t - last
    last = t; act[n] += d
for k, v in tot.most_common(): print("%6.1f %%  %s" % (100.0 * v / (t1 - t0), k))
PY
