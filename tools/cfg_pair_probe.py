#!/usr/bin/env python3
"""Tuning tool: configs[2] (CFG pair, 2B = 16) -- step time, the persistent launch alone, the long-level launches alone, units per phase."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "jen-1-pytorch_amd"))
import torch
import bench as Bn
from jen1_amd.config import full_model_config
from jen1_amd.model import UNetCFG1d
model = UNetCFG1d(**full_model_config(), init_seed=1234, compute_dtype="bf16", device="cuda")
st = Bn.build_stepper(model, 8, 1500, "cuda", cfg_pair=True, use_graph=True)
dts = Bn.timed_steps(st, 100, 10, lambda: None, repeats=3)
print("configs[2] ms/step", [round(d / 100 * 1e3, 4) for d in dts])
d = Bn.deep_roofline(st, "bf16")
print({k: d[k] for k in ("avg_launch_us", "phases", "us_per_phase", "alg_bytes_per_launch", "frac")})
c = Bn.conv_roofline(st)
print({k: c[k] for k in ("launches_per_step", "avg_launch_us", "conv_ms_per_step")})
prog = st.plan.deep
h = prog.hdr.cpu().view(torch.int32).view(-1, 4)
import collections
cnt = collections.Counter(int(h[i, 0]) for i in range(len(prog)))
print("units per phase histogram", sorted(cnt.items()))
