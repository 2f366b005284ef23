#!/usr/bin/env python3
"""Where the torch glue launches of a training pass come from (tuning tool): one eager forward + backward of the bench shape under
torch.profiler, device kernels grouped by name and by the aten op / autograd node that launched them.

    python tools/train_glue.py [--batch 8] > gpurun_out/train_glue.txt
"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jen-1-pytorch_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from jen1_amd import synth  # noqa: E402
from jen1_amd.config import full_model_config  # noqa: E402
from jen1_amd.diffusion import GaussianDiffusion, get_beta_schedule  # noqa: E402
from jen1_amd.model import UNetCFG1d  # noqa: E402
from jen1_amd.optim import FusedAdamW  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--length", type=int, default=1500)
a = ap.parse_args()
dev = lambda v: None if v is None else torch.from_numpy(np.ascontiguousarray(v)).cuda()   # noqa: E731
model = UNetCFG1d(**full_model_config(), init_seed=1234, compute_dtype="bf16", device="cuda")
model.train()
opt = FusedAdamW(model.parameters())
graph = model.train_graph("bf16")
graph.attach_optimizer(opt)
betas, _ = get_beta_schedule("linear", 1000)
gd = GaussianDiffusion(steps=1000, betas=betas, objective="noise", loss_type="l2", device="cuda", cfg_dropout_proba=0.2,
                       embedding_scale=0.8, batch_cfg=True, scale_cfg=True)
B, T = a.batch, a.length
x0 = dev(synth.latents(B, T, key="clip"))
cond = {k: dev(v) for k, v in synth.conditioning(B, T, "music_inpaint").items()}
t = torch.randint(0, 1000, (B,), device="cuda")
for _ in range(2):
    opt.zero_grad()
    gd.training_loosses(graph, x0, t, cond, causal=False).backward()
torch.cuda.synchronize()
opt.zero_grad()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    gd.training_loosses(graph, x0, t, cond, causal=False).backward()
    torch.cuda.synchronize()
ev = prof.events()
# device kernels by name
kern = collections.Counter()
ktime = collections.Counter()
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CUDA:
        kern[e.name[:70]] += 1
        ktime[e.name[:70]] += e.device_time if hasattr(e, "device_time") else e.cuda_time
print("# device kernels: count, total us, name")
for k, n in kern.most_common(25):
    print(f"{n:6d} {ktime[k]:10.0f}  {k}")
# aten ops that launched kernels, grouped by (op, first python frame in jen1_amd)
grp = collections.Counter()
gt = collections.Counter()
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith("aten::"):
        continue
    nk = len(e.kernels)
    if nk == 0:
        continue
    frame = next((f for f in (e.stack or []) if "jen1_amd" in f), None)
    if frame is None:
        frame = next((f for f in (e.stack or []) if "autograd" in f or "Backward" in f), "(autograd engine)")
    key = (e.name, frame.split("jen-1-pytorch_amd/")[-1][:90])
    grp[key] += nk
    gt[key] += sum(k.duration for k in e.kernels)
print("# aten ops with device kernels: kernels, total us, op, python frame")
for k, n in sorted(grp.items(), key=lambda kv: -gt[kv[0]])[:60]:
    print(f"{n:6d} {gt[k]:10.0f}  {k[0]:28s} {k[1]}")

# ---- the GEMM launches of the pass by shape: an event pair around every jen1_train_gemm call (eager, ~3 us of event overhead each)
rt = graph.rt
orig = rt.gemm
recs = []


def timed(a_, b_, c_ptr, M, N, K, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(a_, b_, c_ptr, M, N, K, **kw)
    e1.record()
    kind = "wgrad" if kw.get("taps_in_z") else ("attn" if kw.get("batches", 1) > 1 else "fprop/dgrad")
    recs.append(((kind, M, N, K, kw.get("taps", 1), kw.get("batches", 1), kw.get("splitk", 1)), e0, e1))


rt.gemm = timed
opt.zero_grad()
gd.training_loosses(graph, x0, t, cond, causal=False).backward()
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for key, e0, e1 in recs:
    agg[key][0] += 1
    agg[key][1] += e0.elapsed_time(e1) * 1e3
tot = sum(v[1] for v in agg.values())
print(f"# jen1_train_gemm by shape: {len(recs)} launches, {tot / 1e3:.2f} ms with event pairs; kind M N K taps batches splitk: count, total us, avg us, GFLOP/launch")
for key, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    kind, M, N, K, taps, z, sk = key
    print(f"{kind:12s} M={M:6d} N={N:5d} K={K:6d} taps={taps} z={z:3d} sk={sk:3d}: {n:4d} {us:9.0f} {us / n:8.1f}  {2e-9 * M * N * K * taps * z:8.3f}")
bykind = collections.Counter()
for key, (n, us) in agg.items():
    bykind[key[0]] += us
print("# by kind (us):", dict(bykind))
