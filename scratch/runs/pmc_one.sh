#!/bin/bash
# usage: pmc_one.sh <tag> <workload args of rs_prof.py...>  -- a few PMC passes of one workload
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$tag
mkdir -p $O
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O -o p$i -- python $R/scratch/rs_prof.py "$@" > $O/p$i.log 2>&1
  i=$((i+1))
done
python - <<PY
import csv, collections, glob
acc = collections.defaultdict(list)
for fn in sorted(glob.glob("$O/p*_counter_collection.csv")):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(fn)):
        if "at::" in r["Kernel_Name"] or "distribution" in r["Kernel_Name"]: continue
        per[(r["Kernel_Name"].split("(")[0].replace("void ",""), r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (k, c, _), v in per.items(): acc[(k, c)].append(v)
for (k, c), v in sorted(acc.items()): print("%-40s %-24s %14.0f" % (k, c, sum(v) / len(v)))
PY
