#!/bin/bash
# usage: pmc_bench.sh <tag> [kernel-substring]  -- SQ counter passes of the serial bench, summary for kernels matching the substring
tag=$1; pat=${2:-payload}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcb_$tag
mkdir -p $O
B="python $R/bench.py --serial --no-cpu --no-harvest --steps 4 --warmup 2 --serial-steps 1"
i=0
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES"; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O -o p$i -- $B > $O/p$i.log 2>&1
  i=$((i+1))
done
python - <<PY
import csv, collections, glob
acc = collections.defaultdict(list)
for fn in sorted(glob.glob("$O/p*_counter_collection.csv")):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(fn)):
        if "$pat" not in r["Kernel_Name"]: continue
        per[(r["Kernel_Name"].split("(")[0].replace("void ",""), r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (k, c, _), v in per.items(): acc[(k, c)].append(v)
for (k, c), v in sorted(acc.items()): print("%-40s %-26s %14.0f" % (k, c, sum(v) / len(v)))
PY
