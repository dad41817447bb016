#!/bin/bash
# round 3 final: GPU tests, soak seeds, default bench (with cpu_baseline, harvest and aperiodic legs), driver-form bench, pipeline bench, configs, duplex
O=gpurun_out/r3_final; mkdir -p $O
python -m pytest tests -m gpu -q -x -s 2>&1 | grep -v amdgpu.ids | grep -E "rel err|passed|failed|worst|GPU multichanneltx" | tee $O/tests.log
for seed in 301 302 303 304 305 306 307 308 309 310; do python scratch/soak.py $seed 12 2>&1 | tail -1; done | tee $O/soak.jsonl
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2>/dev/null
python bench.py --pipeline --no-cpu --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_pipeline.json
python bench_duplex.py 2>/dev/null | tail -1 > $O/duplex.json
python scratch/configs_r2.py C2 C2_conv_v27 C3 2>/dev/null | tail -1 > $O/configs.json
python - <<'PY'
import json
for n in ("bench", "bench_steps20", "bench_pipeline", "duplex"):
    try:
        d = json.loads(open("gpurun_out/r3_final/%s.json" % n).read().strip().splitlines()[-1]); print(n, d["value"], d.get("value_aperiodic"), d.get("value_with_harvest"), d.get("verified"))
    except Exception as e: print(n, "ERR", e)
PY
