# end-of-session verification on one box: smoke(), the whole GPU suite, the default bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/s5/verify; mkdir -p $O; rm -f $O/*
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -4) | tee $O/tests.txt
timeout 600 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench20.json
python - <<PY
import json
d=json.loads(open("$O/bench20.json").read()); r=d["roofline"]
print(d["value"], d["ms_per_step"], d["verified"]["ok"], r["kernel"], r["frac"], r["traffic"], r["traffic_source"][:30], d["cpu_baseline"]["value"], d.get("value_aperiodic"), d.get("value_with_harvest"))
PY
