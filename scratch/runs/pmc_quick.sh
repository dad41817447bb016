#!/bin/bash
# usage: pmc_quick.sh <tag> [ENV=VAL ...] -- two PMC passes of the serial bench, per-kernel means printed
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcq_$tag
mkdir -p $O
B="python $R/bench.py --serial --no-cpu --no-harvest --no-aperiodic --steps 6 --warmup 2 --serial-steps 1"
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
  n=$(echo $c | cut -d' ' -f1)
  env "$@" rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O -o pmc_$n -- $B > $O/pmc_$n.log 2>&1
done
python - <<PY
import csv,glob,collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(d.items()):
    if not any(s in k for s in ("payload","decode_kernel","channelizer","sync_spec","sync_lean")): continue
    print(k, {c: round(sum(x)/len(x)/1e6,2) for c,x in sorted(v.items())})
PY
