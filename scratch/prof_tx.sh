#!/bin/bash
# usage: prof_tx.sh <tag>  -- GPU transmitter (scratch/tx_time.py, N = 512: txsym + fused synthesis kernel over one bench slab):
# rocprofv3 --kernel-trace --stats, then separate PMC passes (HBM bytes: FETCH_SIZE / WRITE_SIZE each alone), condensed into
# gpurun_out/prof_tx_<tag>/{kernel_stats.csv,pmc.csv,traffic.json}
tag=${1:-a}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_tx_$tag
mkdir -p $O
W="python $R/scratch/tx_time.py 512"
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o stats -- $W > $O/stats.log 2>&1
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O -o p$i -- $W > $O/p$i.log 2>&1
  i=$((i+1))
done
python - <<PY
import csv, collections, glob, json
O = "$O"
acc = collections.defaultdict(lambda: [0.0, 0])
for fn in sorted(glob.glob(O + "/*_counter_collection.csv")):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(fn)):
        if "mcrx::" not in r["Kernel_Name"]: continue
        per[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (k, c, _), v in per.items():
        a = acc[(k, c)]; a[0] += v; a[1] += 1
with open(O + "/pmc.csv", "w") as f:
    f.write("kernel,counter,dispatches,mean_per_dispatch\n")
    for (k, c), (s, n) in sorted(acc.items()): f.write('"%s",%s,%d,%.1f\n' % (k, c, n, s / n))
tr = {}
for (k, c), (s, n) in acc.items():
    if c in ("FETCH_SIZE", "WRITE_SIZE"): tr.setdefault(k, {})[c + "_KiB"] = s / n
for k, t in tr.items(): t["hbm_bytes_per_launch"] = (2.0 * t.get("FETCH_SIZE_KiB", 0.0) + t.get("WRITE_SIZE_KiB", 0.0)) * 1024.0
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of scratch/tx_time.py 512",
           "correction": "hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts wide coalesced reads at half size)", "kernels": tr},
          open(O + "/traffic.json", "w"), indent=1)
rows = [r for r in csv.DictReader(open(glob.glob(O + "/*stats_kernel_stats.csv")[0])) if "mcrx::" in r["Name"]]
with open(O + "/kernel_stats.csv", "w") as f:
    w = csv.DictWriter(f, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows)
for r in rows: print(r["Name"][:60], r["Calls"], r["AverageNs"])
for k, t in tr.items(): print(k, {a: round(b / 1e6, 1) for a, b in t.items()})
PY
