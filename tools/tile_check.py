#!/usr/bin/env python3
"""debugging aid: the plan with tile phases against the launch-per-layer plan, activation by activation (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jen-1-pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
from jen1_amd import synth
from jen1_amd.config import full_model_config
from jen1_amd.model import UNetCFG1d

dt = sys.argv[1] if len(sys.argv) > 1 else "f32"
B, T, nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 8, int(sys.argv[3]) if len(sys.argv) > 3 else 1500, int(sys.argv[4]) if len(sys.argv) > 4 else 1
model = UNetCFG1d(**full_model_config(), init_seed=1234, compute_dtype=dt, device="cuda")
eng = model.engine()
pd = eng.plan(B, T, nrep, False, deep=True)
pl = eng.plan(B, T, nrep, False, deep=False)
print("deep_level", pd.deep_level, "tile_lens", sorted(pd.tile_lens), "programs", [(len(p), p.kinds.count("tile"), p.lds) for p in pd.progs], "launches", pd.n_launch, "vs", pl.n_launch)
print("tile errors:", pd.tile_errors[:6])
for p in pd.progs:
    for i, (k, l) in enumerate(zip(p.kinds, p.labels)):
        if k in ("tile", "stats"):
            print("   ", i, l)
dev = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()
x, cond = synth.latents(B, T), synth.conditioning(B, T, "text_guided")
t = np.array([(131 * i + 7) % 1000 for i in range(B)], dtype=np.int64)
s = torch.cuda.current_stream().cuda_stream
for plan in (pl, pd):
    model._prepare(plan, dev(x), dev(t), dev(cond["cross_attn_cond"]), dev(cond["cross_attn_masks"]), [dev(cond["input_concat_cond"])], None)
    plan.run(s)
    torch.cuda.synchronize()
print("error word", pd.take_error())
pairs = list(zip(pd.acts, pl.acts)) if len(pd.acts) == len(pl.acts) else [(pd.taps[k], pl.taps[k]) for k in pd.taps]
print("acts", len(pd.acts), len(pl.acts))
worst = 0.0
for i, (a, b) in enumerate(pairs):
    ra, rb = a.t[:, :, : a.C].float(), b.t[:, :, : b.C].float()
    if not torch.isfinite(rb).all():
        continue
    den = float(rb.abs().max())
    if den == 0:
        continue
    fin = bool(torch.isfinite(ra).all())
    e = float((ra - rb).abs().max()) / den if fin else float("nan")
    worst = max(worst, e) if fin else float("inf")
    if not fin or e > (1e-4 if dt == "f32" else 6e-2):
        d = (ra - rb).abs()
        d = torch.where(torch.isfinite(d), d, torch.full_like(d, 1e30))
        idx = np.unravel_index(int(d.argmax()), d.shape)
        print(f"act {i} shape {tuple(a.t.shape)} C={a.C}: err {e:.3e} finite={fin} worst at {idx}: {float(ra[idx]):.5f} vs {float(rb[idx]):.5f}; nonfinite count {int((~torch.isfinite(ra)).sum())}")
        if i > 40:
            break
print("worst", worst)
