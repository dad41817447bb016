#!/bin/bash
# configs[1] (8 channels) against the slab length: frames per channel and push
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q -k "stream or soak or spec or ragged or launch or baseline" 2>&1 | tail -3
for f in ${FR:-100 200 400 800}; do echo "== frames/ch/slab $f"; FRAMES=$f python scratch/cfg_probe.py 8ch 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  ', d['value'], d['ms_per_step'], d['kernels_ms_overlapped'], d['verified']['ok'], d['frames_acquired'])"; done
