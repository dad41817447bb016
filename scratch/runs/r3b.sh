#!/bin/bash
O=gpurun_out/r3b; mkdir -p $O
MCRX_LIB=$PWD/scratch/libmcrx_r2ref.so python scratch/chan_ab.py $O/ref.pt write 2>&1 | grep -v amdgpu.ids | tee $O/ab.log
for v in B C D E F G; do MCRX_LIB=$PWD/scratch/libs/libmcrx_$v.so python scratch/chan_ab.py $O/ref.pt 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.log; done
B="python bench.py --no-cpu --no-harvest --steps 30 --warmup 8 --serial-steps 3"
MCRX_LIB=$PWD/scratch/libmcrx_r2ref.so $B 2>/dev/null | tail -1 > $O/bench_ref.json
for v in B C D E F G; do MCRX_LIB=$PWD/scratch/libs/libmcrx_$v.so $B 2>/dev/null | tail -1 > $O/bench_$v.json; done
rm -f $O/ref.pt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3b/bench_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("/")[-1], "value", d["value"], "alone", r["kernels_ms"], "ovl", r["kernels_ms_overlapped"], d["verified"]["ok"])
    except Exception as e:
        print(f, "ERR", e)
PY
