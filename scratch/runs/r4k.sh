#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4k; mkdir -p $O
for w in 8ch; do
  echo "== $w"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o $w -- python $R/scratch/cfg_probe.py $w > $O/$w.log 2>&1
  tail -1 $O/$w.log | cut -c1-300
  python3 - <<PY
import csv
for r in csv.DictReader(open("$O/${w}_kernel_stats.csv")):
    n=r["Name"]
    if "mcrx::" in n and not any(k in n for k in ("txsym","synth","ilmap","txfir")): print("   %-40s calls %5s avg %9.1f us  min %9.1f max %9.1f  total %8.1f ms" % (n.split("(")[0].replace("void mcrx::","")[:40], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
done
