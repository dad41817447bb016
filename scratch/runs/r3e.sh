#!/bin/bash
O=gpurun_out/r3e; mkdir -p $O
L=$PWD/scratch/libs
MCRX_LIB=$L/libmcrx_R8.so python scratch/chan_ab.py $O/ref.pt write 2>&1 | grep -v amdgpu.ids | tee $O/ab.log
for v in R4a R4b R4c R4d; do MCRX_LIB=$L/libmcrx_$v.so python scratch/chan_ab.py $O/ref.pt 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.log; done
MCRX_LIB=$PWD/scratch/libmcrx_r2ref.so python scratch/chan_ab.py $O/ref.pt 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.log
rm -f $O/ref.pt
MCRX_LIB=$L/libmcrx_R4a.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "channelizer or any_channel or full_chain" 2>&1 | tail -5 | tee $O/tests.log
B="python bench.py --no-cpu --no-harvest --steps 30 --warmup 8 --serial-steps 3"
for v in R8 R4a R4b R4c R4d; do MCRX_LIB=$L/libmcrx_$v.so $B 2>/dev/null | tail -1 > $O/bench_$v.json; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3e/bench_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("/")[-1], "value", d["value"], "alone", r["kernels_ms"], "ovl", r["kernels_ms_overlapped"], d["verified"]["ok"])
    except Exception as e:
        print(f, "ERR", e)
PY
