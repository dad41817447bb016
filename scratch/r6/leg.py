"""One leg of bench.py's `configs` block by name (round 6 iteration tool): python scratch/r6/leg.py 512ch_pfb2_front_end [steps reps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
import torch
from __graft_entry__ import load_product
prod = load_product()
dev = torch.device("cuda:0")
LEGS = {
    "8ch": (8, 64, 8, 100, 1200, 40, 6, False, 0),
    "8ch_v27": (8, 64, 8, 100, 1200, 40, 11, False, 0),
    "64ch_m256_qam16_resamp": (64, 256, 32, 32, 1200, 27, 7, True, 0),
    "64ch_m256_qpsk": (64, 256, 32, 32, 1200, 40, 7, False, 0),
    "64ch_m256_qam16": (64, 256, 32, 32, 1200, 27, 7, False, 0),
    "64ch_m256_qam64": (64, 256, 32, 32, 1200, 29, 7, False, 0),
    "8ch_long_pushes": (8, 64, 8, 400, 1200, 40, 6, False, 0),
    "512ch_m48": (512, 48, 6, 16, 1200, 40, 6, False, 0),
    "512ch": (512, 64, 8, 16, 1200, 40, 6, False, 0),
    "512ch_pfb2_front_end": (512, 64, 8, 16, 1200, 40, 6, False, 1),
    "512ch_pfb2_chain": (512, 64, 8, 16, 1200, 40, 6, False, 2),
    "64ch_pfb2_front_end": (64, 64, 8, 32, 1200, 40, 6, False, 1),
}
name = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
N, M, cp, fr, pl, mod, fec1, rsmp, fe = LEGS[name]
for kv in os.environ.get('LEG_CFG', '').split(','):
    if '=' in kv:
        k, v = kv.split('='); bench.LEG_CFG[k] = int(v)
out = bench.config_leg(prod, torch, dev, N, M, cp, fr, pl, mod, fec1, rsmp, steps=steps, reps=reps, what=name, front_end=fe)
print(json.dumps({name: out}))
