"""one leg of bench.py's configs block by itself (for rocprofv3): python scratch/cfg_probe.py 8ch|8ch_v27|c3|m48"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
from __graft_entry__ import load_product
prod = load_product()
dev = torch.device("cuda", 0)
which = sys.argv[1] if len(sys.argv) > 1 else "8ch"
legs = {"8ch": (8, 64, 8, 100, 1200, 40, 6, False), "8ch_v27": (8, 64, 8, 100, 1200, 40, 11, False), "c3": (64, 256, 32, 32, 1200, 27, 7, True),
        "m48": (512, 48, 6, 4, 1200, 40, 6, False)}          # the applications' default symbol (not a power of two): the general path
N, M, cp, fr, pl, mod, fec1, rs = legs[which]
fr = int(os.environ.get("FRAMES", fr))
r = bench.config_leg(prod, torch, dev, N, M, cp, fr, pl, mod, fec1, rs, steps=int(os.environ.get("STEPS", "6")), reps=3, what=which)
print(json.dumps(r))
