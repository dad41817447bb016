#!/bin/bash
# round 4: acquisition-kernel builds A/B (seek bursts on/off x register budget), same box, same call
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4f; mkdir -p $O
for v in b168 n168 b256 n256 b512; do
  export MCRX_LIB=$GRAFT_REPO_ROOT/scratch/libs/libmcrx_$v.so
  echo "== $v"
  python bench.py --no-cpu --no-harvest --no-aperiodic --steps 30 --warmup 8 --serial-steps 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  periodic value', d['value'], 'sync alone', d['roofline']['kernels_ms'].get('sync_kernel'), 'overlapped', d['roofline']['kernels_ms_overlapped'].get('sync_kernel'), d['verified']['ok'])"
  python scratch/aper_probe.py 0 2>/dev/null | tail -1 | cut -c1-200 | sed 's/^/  /'
  python scratch/aper_probe.py 1 2>/dev/null | tail -1 | cut -c1-200 | sed 's/^/  /'
done
cd /tmp && export TMPDIR=/tmp
for v in b168 n256; do
  export MCRX_LIB=$GRAFT_REPO_ROOT/scratch/libs/libmcrx_$v.so
  echo "== kernel split, ragged serial, $v"
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O -o rag_$v -- python $GRAFT_REPO_ROOT/scratch/aper_probe.py 1 > $GRAFT_REPO_ROOT/$O/rag_$v.log 2>&1
  python3 - <<PY
import csv
for r in csv.DictReader(open("$GRAFT_REPO_ROOT/$O/rag_${v}_kernel_stats.csv")):
    n=r["Name"]
    if "sync_" in n: print("   %-30s calls %5s avg %9.1f us  min %9.1f max %9.1f" % (n.split("(")[0].replace("void mcrx::","")[:30], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
done
