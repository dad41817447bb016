#!/bin/bash
# round 3, first GPU call: tests, baseline bench, slab-size and packed-worker experiments
O=gpurun_out/r3a; mkdir -p $O
python -m pytest tests -m gpu -q -x -s 2>&1 | grep -v "amdgpu.ids" | tail -40 > $O/tests.log
tail -3 $O/tests.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err
B="python bench.py --no-cpu --no-harvest --steps 30 --warmup 8 --serial-steps 3"
for sb in 0 200 128 96; do $B --slab-blocks $sb 2>/dev/null | tail -1 > $O/bench_slab$sb.json; done
for fr in 2 4 0; do MCRX_PAYLOAD_FR=$fr $B 2>/dev/null | tail -1 > $O/bench_fr$fr.json; done
python bench.py --pipeline --no-cpu --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_pipeline.json
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3a/bench_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("/")[-1], "value", d["value"], "ms/step", d["ms_per_step"], r["kernel"], r["frac"], "alone", r["kernels_ms"], "ovl", r["kernels_ms_overlapped"], d["verified"]["ok"], d.get("spec_hit_rate"))
    except Exception as e:
        print(f, "ERR", e)
PY
