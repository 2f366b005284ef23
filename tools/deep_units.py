#!/usr/bin/env python3
"""units per phase of the persistent deep-level program (tuning tool): python tools/deep_units.py [--batch 8] [--length 1500] [--cfg]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jen-1-pytorch_amd")):
    sys.path.insert(0, p)
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--length", type=int, default=1500)
ap.add_argument("--cfg", action="store_true")
ap.add_argument("--dtype", default="bf16")
args = ap.parse_args()
import torch
from jen1_amd.config import full_model_config
from jen1_amd.model import UNetCFG1d
model = UNetCFG1d(**full_model_config(), init_seed=1234, compute_dtype=args.dtype, device="cuda")
plan = model.engine().plan(args.batch, args.length, 2 if args.cfg else 1, False, n_t=100)
print("deep_level", plan.deep_level, plan.deep_errors)
prog = plan.deep
h = prog.hdr.cpu().view(torch.int32).view(-1, 4)
tot = 0
for i in range(len(prog)):
    nu = int(h[i, 0])
    tot += (nu + prog.nwg - 1) // prog.nwg
    print(f"{i:3d} units={nu:5d} rounds={(nu + prog.nwg - 1) // prog.nwg} {prog.labels[i]}")
print("total rounds", tot, "phases", len(prog))
