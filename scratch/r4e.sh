#!/bin/bash
# round 4: pipelined throughput by segments per channel (periodic bench stream / ragged stream), seek bursts on and off
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4e; mkdir -p $O
one() {
  python bench.py --no-cpu --no-harvest --no-aperiodic --steps 30 --warmup 8 --serial-steps 2 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  periodic value', d['value'], d['frames_acquired'], d['roofline']['kernels_ms'].get('sync_kernel'), d['roofline']['kernels_ms_overlapped'].get('sync_kernel'), d['verified']['ok'])"
  python scratch/aper_probe.py 0 2>/dev/null | tail -1 | sed 's/^/  /'
  python scratch/aper_probe.py 1 2>/dev/null | tail -1 | sed 's/^/  /'
}
for ns in auto 1 2 3 4 6; do
  if [ $ns = auto ]; then unset MCRX_NSEG; else export MCRX_NSEG=$ns; fi
  echo "== NSEG=$ns"; one
done
unset MCRX_NSEG
echo "== auto, MCRX_SEEK_BURST=0"; MCRX_SEEK_BURST=0 one
echo "== GPU tests"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
