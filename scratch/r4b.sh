#!/bin/bash
# round 4: MCRX_TILE 16 -- GPU tests, bench line, FETCH/WRITE passes of the serial bench
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4b; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.txt 2>&1; echo "pytest rc $?" >> $O/tests.txt
tail -15 $O/tests.txt
timeout 600 python bench.py --no-cpu 2>$O/bench.err | tail -1 > $O/bench.json
python - <<PY
import json
d = json.loads(open("$O/bench.json").read())
print("value", d["value"], "harvest", d.get("value_with_harvest"), "aper", d.get("value_aperiodic"), d["roofline"]["kernels_ms"], d["roofline"]["kernels_ms_overlapped"], d["verified"])
PY
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --serial --no-cpu --no-harvest --no-aperiodic --steps 6 --warmup 2 --serial-steps 1"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O -o pmc_$c -- $B > $R/$O/pmc_$c.log 2>&1
done
python - <<PY
import csv,glob,collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/$O/**/*counter_collection.csv", recursive=True):
    per=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r["Kernel_Name"].split("(")[0][-44:], r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (k,c,_),v in per.items(): d[k][c].append(v)
for k,v in sorted(d.items()):
    if not any(s in k for s in ("payload","decode_kernel","channelizer")): continue
    print(k, {c: round(sum(x)/len(x)*1024/1e6,1) for c,x in sorted(v.items())}, "MB (x2 for FETCH)")
PY
