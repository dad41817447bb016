# Viterbi block-size A/B: the convolutional-code tests first (byte equality with the oracle), then the 8-channel stream with the K = 7 code
cd $GRAFT_REPO_ROOT; O=gpurun_out/s5/vit; mkdir -p $O
(timeout 400 python -m pytest tests -m gpu -x -q -k "conv or viterbi or soak or v27" 2>&1 | tail -3) | tee $O/tests.txt
timeout 200 python scratch/configs_r2.py C2_conv_v27 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read())['C2_conv_v27']; print(d['Msamples_per_s'], d['ms_per_step'], d['verified'], d['kernels_ms_overlapped'])"
