#!/bin/bash
# viterbi_frames_kernel: duration against the number of frames per launch (latency of a wave or throughput of the chip?)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for fr in 50 100 200 400; do
  O=$R/gpurun_out/prof_v27_f$fr; rm -rf $O; mkdir -p $O
  (cd $R; V27_SERIAL=${V27_SERIAL:-0} V27_FRAMES=$fr rocprofv3 --kernel-trace --stats --output-format csv -d $O -o v -- python scratch/r5/v27_probe.py > $O/v.log 2> $O/v.err)
  echo "== frames/ch $fr: $(tail -1 $O/v.log)"; grep Gsample $O/v.log | cut -c1-60
  grep -E "viterbi|decode_general|decode_kernel" $O/v_kernel_stats.csv | awk -F, '{printf "   %-50s calls %4s avg %8.1f us\n", substr($1,1,50), $2, $4/1000}'
done
python3 - <<'PY'
import csv, os
R = os.environ.get("GRAFT_REPO_ROOT", ".")
for fr in (50, 100, 200, 400):
    rows = [r for r in csv.DictReader(open(f"{R}/gpurun_out/prof_v27_f{fr}/v_kernel_trace.csv")) if "viterbi" in r["Kernel_Name"]]
    print(fr, [round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000) for r in rows], "grid", rows[-1]["Grid_Size_X"], "lds", rows[-1]["LDS_Block_Size"])
PY
