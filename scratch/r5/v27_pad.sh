#!/bin/bash
# viterbi_frames_kernel: workgroups per CU (unused dynamic LDS) against launch duration; needs a -DVF_PROF build
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for fr in 100 200 400; do for pad in 0 4096 24576 65536; do
  O=$R/gpurun_out/prof_v27_pad; rm -rf $O; mkdir -p $O
  (cd $R; VF_PAD=$pad V27_FRAMES=$fr rocprofv3 --kernel-trace --output-format csv -d $O -o v -- python scratch/r5/v27_probe.py > $O/v.log 2> $O/v.err)
  python3 - $O/v_kernel_trace.csv $fr $pad <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "viterbi" in r["Kernel_Name"]]
d = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000 for r in rows)
print("frames/ch", sys.argv[2], "pad", sys.argv[3], "median %.0f us" % d[len(d) // 2], "lds", rows[-1]["LDS_Block_Size"])
PY
done; done
