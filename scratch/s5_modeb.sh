# mode B by the number of copy threads (MCRX_H2D_THREADS), after the host-path parity tests
cd $GRAFT_REPO_ROOT; O=gpurun_out/s5/modeb; mkdir -p $O
(timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stream.py tests/test_gpu_refapp.py -m gpu -x -q 2>&1 | tail -3) | tee $O/tests.txt
for t in 1 2 4 6 8; do for a in "512 8 5" "8 100 5"; do echo "threads $t: $(MCRX_H2D_THREADS=$t timeout 120 liquid-usrp_amd/lib/modeb $a 2>/dev/null | tail -1 | cut -c75-330)"; done; done | tee $O/modeb.txt
