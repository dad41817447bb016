#!/bin/bash
# round 4: where does the segment-parallel acquisition spend its time?  per-kernel averages (rocprofv3 --stats) by segments per channel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4d; mkdir -p $O
summ() { python3 - "$1" <<'PY'
import csv,sys,glob
for f in glob.glob(sys.argv[1]+"*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        n=r["Name"]
        if any(k in n for k in ("sync_","place_jobs","payload_lean_kernel","decode_kernel","channelizer_kernel")):
            print("   %-34s calls %5s avg %9.1f us  min %9.1f max %9.1f" % (n.split("(")[0].replace("void mcrx::","")[:34], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
}
for ns in auto 1 2 4 8 16; do
  if [ $ns = auto ]; then unset MCRX_NSEG; else export MCRX_NSEG=$ns; fi
  echo "== periodic (bench --serial), NSEG=$ns"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o per_$ns -- python $R/bench.py --serial --no-cpu --no-harvest --no-aperiodic --steps 6 --warmup 2 --serial-steps 1 > $O/per_$ns.log 2>&1
  python3 -c "import json;d=json.loads(open('$O/per_$ns.log').read().strip().split('\n')[-1]);print('   value',d['value'],d['frames_acquired'])"
  summ $O/per_${ns}_
  echo "== ragged (aper_probe serial), NSEG=$ns"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -o rag_$ns -- python $R/scratch/aper_probe.py 1 > $O/rag_$ns.log 2>&1
  tail -1 $O/rag_$ns.log
  summ $O/rag_${ns}_
done
