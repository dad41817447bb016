#!/bin/bash
cd $GRAFT_REPO_ROOT
j() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  value', d['value'], d['value_min'], d['value_max'], d['config'].get('multi_gpu_path'), d['verified']['ok'], d['roofline']['kernels_ms_overlapped'])"; }
echo "== direct"; python bench.py --no-cpu --no-harvest --no-aperiodic --no-configs --steps 20 --warmup 5 --reps 3 2>/dev/null | tail -1 | j
echo "== --pipeline (C-ABI)"; python bench.py --pipeline --no-cpu --steps 20 --warmup 5 --reps 3 2>/dev/null | tail -1 | j
echo "== --pipeline --exchange torch"; python bench.py --pipeline --exchange torch --no-cpu --steps 20 --warmup 5 --reps 3 2>/dev/null | tail -1 | j
echo "== direct"; python bench.py --no-cpu --no-harvest --no-aperiodic --no-configs --steps 20 --warmup 5 --reps 3 2>/dev/null | tail -1 | j
exit 0
python - <<'PY'
import sys, os, json
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch, bench
from __graft_entry__ import load_product
prod = load_product()
r = bench.config_leg(prod, torch, torch.device("cuda", 0), 8, 48, 6, 100, 1200, 40, 6, False, steps=6, reps=3, what="8ch M=48 cp=6")
print("  ", r["value"], r["frac_of_roofline"], r["kernels_ms_overlapped"], r["verified"], r["frames_acquired"])
PY
