#!/bin/bash
O=gpurun_out/r3c; mkdir -p $O
L=$PWD/scratch/libs
MCRX_LIB=$PWD/scratch/libmcrx_r2ref.so python scratch/chan_ab.py $O/ref.pt write 2>&1 | grep -v amdgpu.ids | tee $O/ab.log
for v in D D1 H1 H2 H3; do MCRX_LIB=$L/libmcrx_$v.so python scratch/chan_ab.py $O/ref.pt 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.log; done
rm -f $O/ref.pt
B="python bench.py --no-cpu --no-harvest --steps 30 --warmup 8 --serial-steps 3"
run() { name=$1; shift; env "$@" $B 2>/dev/null | tail -1 > $O/bench_$name.json; }
run D MCRX_LIB=$L/libmcrx_D.so
run D_wpb8 MCRX_LIB=$L/libmcrx_D.so MCRX_PAYLOAD_WPB=8
run D_wpb8_cp MCRX_LIB=$L/libmcrx_D.so MCRX_PAYLOAD_WPB=8 MCRX_CHAN_PRIO=1
run D1_wpb8_cp MCRX_LIB=$L/libmcrx_D1.so MCRX_PAYLOAD_WPB=8 MCRX_CHAN_PRIO=1
run D1_wpb4_cp MCRX_LIB=$L/libmcrx_D1.so MCRX_PAYLOAD_WPB=4 MCRX_CHAN_PRIO=1
run D1_cp MCRX_LIB=$L/libmcrx_D1.so MCRX_CHAN_PRIO=1
run D1_fr4_cp MCRX_LIB=$L/libmcrx_D1.so MCRX_PAYLOAD_FR=4 MCRX_CHAN_PRIO=1
run H1 MCRX_LIB=$L/libmcrx_H1.so
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r3c/bench_*.json")):
    try:
        d = json.load(open(f)); r = d["roofline"]
        print(f.split("/")[-1], "value", d["value"], "alone", r["kernels_ms"], "ovl", r["kernels_ms_overlapped"], d["verified"]["ok"])
    except Exception as e:
        print(f, "ERR", e)
PY
