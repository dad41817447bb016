# final lines of the session: python bench.py (default), the driver's form, the C-ABI pipeline on one GPU, bench_duplex.py
cd $GRAFT_REPO_ROOT; O=gpurun_out/s5/final; mkdir -p $O
timeout 900 python bench.py 2>/dev/null | tail -1 > $O/bench.json
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_steps20.json
timeout 600 python bench.py --pipeline --steps 20 --warmup 5 --no-cpu 2>/dev/null | tail -1 > $O/bench_pipeline.json
timeout 600 python bench_duplex.py 2>/dev/null | tail -1 > $O/duplex.json
python - <<PY
import json
for f in ("bench", "bench_steps20", "bench_pipeline", "duplex"):
    d = json.loads(open("$O/%s.json" % f).read())
    print(f, d["value"], d["ms_per_step"], d.get("verified", {}).get("ok"), d.get("value_with_harvest"), d.get("value_aperiodic"), (d.get("roofline") or {}).get("frac"), (d.get("cpu_baseline") or {}).get("value"))
PY
