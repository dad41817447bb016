#!/bin/bash
# the driver's command, timed: python bench.py (defaults) -> gpurun_out/r6_bench_<tag>.json
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tag=${1:-a}
( time python bench.py > gpurun_out/r6_bench_$tag.json 2> gpurun_out/r6_bench_$tag.err ) 2>&1 | grep real
python - <<PY
import json
d=json.load(open("gpurun_out/r6_bench_$tag.json"))
print("value", d["value"], d["value_min"], d["value_max"], "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["ms_per_launch"], "pipe frac", d["roofline"]["pipeline_frac_of_16B_roofline"])
for k in ("value_with_harvest","value_aperiodic","value_awgn30","value_without_framesyms","value_default_hw_queues"):
    print(k, d.get(k))
for k,v in d.get("configs",{}).items():
    print(k, v.get("value"), v.get("frac_of_roofline"), v.get("verified",{}).get("ok"), v.get("error"), (v.get("roofline") or {}).get("frac"))
print("cpu", d.get("cpu_baseline",{}).get("value"), d.get("verified"))
PY
