# kernel table of configs[1] (8 channels) as a continuous stream
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s5/c2; mkdir -p $O
CFG_STEPS=10 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o c2 -- python $R/scratch/configs_r2.py ${1:-C2} > $O/c2.log 2>&1
python - <<PY
import csv,glob
rows=list(csv.DictReader(open(glob.glob("$O/*c2_kernel_stats.csv")[0])))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:16]: print(f"{r['Name'][:64]:64s} {r['Calls']:>5s} {float(r['TotalDurationNs'])/1e6:9.3f} ms avg {float(r['AverageNs'])/1e3:8.1f} us max {float(r['MaxNs'])/1e3:8.1f}")
PY
tail -2 $O/c2.log | cut -c1-600
