#!/bin/bash
O=gpurun_out/r3f; mkdir -p $O
B="python bench.py --no-cpu --no-harvest --steps 30 --warmup 8 --serial-steps 3"
run() { name=$1; shift; env "$@" $B 2>$O/err_$name.log | tail -1 > $O/bench_$name.json; }
run base X=1
run lock MCRX_PHASE_LOCK=1
run lock_sp0 MCRX_PHASE_LOCK=1 MCRX_SPACER=0
run lock_sp8k MCRX_PHASE_LOCK=1 MCRX_SPACER=8000
run lock_cp MCRX_PHASE_LOCK=1 MCRX_CHAN_PRIO=1
run lock_wpb8 MCRX_PHASE_LOCK=1 MCRX_PAYLOAD_WPB=8
run lock_fr4 MCRX_PHASE_LOCK=1 MCRX_PAYLOAD_FR=4
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3f/bench_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("/")[-1], "value", d["value"], "alone", r["kernels_ms"], "ovl", r["kernels_ms_overlapped"], d["verified"]["ok"])
    except Exception as e:
        print(f, "ERR", e)
PY
