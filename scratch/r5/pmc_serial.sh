#!/bin/bash
# PMC passes of the serial receiver (every kernel alone): scratch/r5/pmc_serial.sh <tag>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_$tag
mkdir -p $O
B="python $R/bench.py --serial --no-cpu --no-harvest --no-aperiodic --no-configs --steps 3 --warmup 1 --reps 1"
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
  n=$(echo $c | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O -o pmc_$n -- $B "$@" > $O/pmc_$n.log 2>&1
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$O/pmc_*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"].startswith("SQ_WAVES") or r["Counter_Name"] == "SQ_INSTS_LDS": cnt[k] += 1
    for k in agg:
        if any(s in k for s in ("acq_lean", "sync_walk", "payload_lean_kernel", "decode_kernel", "sync_spec")):
            n = max(cnt[k], 1)
            print(k, "launches", n, {c: round(v / n) for c, v in agg[k].items()})
PY
