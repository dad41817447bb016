#!/bin/bash
# scouts ranking the window's chain (default build) against serial hops (libmcrx_win2.so ~ the build before): GPU suite, 8-channel legs, headline, ragged
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for f in 100 400 800; do echo "== 8 channels, frames/ch/push $f"; FRAMES=$f python scratch/cfg_probe.py 8ch 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ', d['value'], d['ms_per_step'], d['kernels_ms_overlapped'], d['verified']['ok'], d['frames_acquired'])"; done
python bench.py --no-cpu --no-harvest --no-configs --steps 20 --warmup 5 --reps 3 --serial-steps 4 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  value', d['value'], d['roofline']['kernels_ms'], d['verified']['ok'], 'aperiodic', d.get('value_aperiodic'), d['value_aperiodic_detail']['verified']['ok'], d['value_aperiodic_detail']['frames_acquired'])"
