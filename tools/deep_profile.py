#!/usr/bin/env python3
"""Per-phase timeline of the persistent deep-level launch (tuning tool, not part of the product).

Builds a second copy of the library with -DJEN1_DEEP_PROFILE (every workgroup stamps the 100 MHz counter at the stages of
each unit it runs), replays one denoiser step of the bench workload and prints, per phase: units, first start -> last
arrival, and the mean duration of the stages over the workgroups that had a unit.

    python tools/deep_profile.py [--batch 8] [--length 1500] [--cfg] [--dtype bf16] > gpurun_out/deep_profile.txt
"""
import argparse
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jen-1-pytorch_amd")):
    sys.path.insert(0, p)

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--length", type=int, default=1500)
ap.add_argument("--cfg", action="store_true")
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--general", action="store_true", help="one timestep per batch element (model.forward) instead of the sampler's table mode")
ap.add_argument("--defs", default="", help="extra -D flags for the tuning build, space separated")
ap.add_argument("--prog", type=int, default=-1, help="which persistent program of the plan (default: the one with the deep levels)")
args = ap.parse_args()

out_lib = os.path.join(ROOT, "gpurun_out", "libjen1_hip_prof.so")
os.makedirs(os.path.dirname(out_lib), exist_ok=True)
csrc = os.path.join(ROOT, "jen-1-pytorch_amd", "csrc")
os.environ["JEN1_LIB"] = out_lib
from jen1_amd import lib as L  # noqa: E402
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DJEN1_DEEP_PROFILE", *args.defs.split(),
       f"-I{os.path.join(ROOT, 'include')}", f"-I{csrc}", *[os.path.join(csrc, s) for s in L.SOURCES], "-o", out_lib]
subprocess.run(cmd, check=True)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from jen1_amd import synth  # noqa: E402
from jen1_amd.config import full_model_config  # noqa: E402
from jen1_amd.model import UNetCFG1d  # noqa: E402

lib = L.load()
lib.jen1_deep_debug_buffer.restype = C.c_int
lib.jen1_deep_debug_buffer.argtypes = [C.c_void_p]
dev = "cuda"
model = UNetCFG1d(**full_model_config(), init_seed=1234, compute_dtype=args.dtype, device=dev)
B, T = args.batch, args.length
nrep = 2 if args.cfg else 1
plan = model.engine().plan(B, T, nrep, False, n_t=None if args.general else 100)
assert plan.deep_level is not None, plan.deep_errors
x, cond = synth.latents(B, T), synth.conditioning(B, T)
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
t = np.array([(131 * i + 7) % 1000 for i in range(plan.n_t)], dtype=np.int64)
if plan.table_mode:
    plan.t_in.copy_(tt(t))
model._prepare(plan, tt(x), tt(t), tt(cond["cross_attn_cond"]), tt(cond["cross_attn_masks"]), [tt(cond["input_concat_cond"])], None)
prog = plan.deep if args.prog < 0 else plan.progs[args.prog]
print(f"# programs of the plan: {[(len(p_), p_.kinds.count('tile')) for p_ in plan.progs]}; profiling {'the deep one' if args.prog < 0 else args.prog}")
n, nwg = len(prog), prog.nwg
dbg = torch.zeros((n, nwg, 16), dtype=torch.int64, device=dev)
s = torch.cuda.current_stream().cuda_stream
if plan.table_mode:
    plan.run_time(s)
for rep in range(args.reps):
    plan.run(s)
torch.cuda.synchronize()
# the stamps are indexed by (phase, workgroup) of ONE program: run the step, then the chosen program alone on what the step left
# in its input buffers
for rep in range(3):
    prog.poison(s)
    prog.launch(s)
torch.cuda.synchronize()
assert lib.jen1_deep_debug_buffer(dbg.data_ptr()) == 0
dbg.zero_()
prog.poison(s)
torch.cuda.synchronize()
prog.launch(s)
torch.cuda.synchronize()
assert prog.error() == 0
d = dbg.cpu().numpy().astype(np.float64) * 0.01       # microseconds
t0 = d[d[:, :, 0] > 0][:, 0].min()
print(f"# B={B} T={T} nrep={nrep} dtype={args.dtype}: {n} phases on {nwg} workgroups, {prog.lds} B LDS; times in us")
print(f"# stages: issue = unit start -> ring + parameter requests issued; wait = dependency wait; stage = loads + norm -> LDS; "
      f"kloop; epi = reduce + epilogue + drain; arr = arrive")
print(f"{'ph':>3} {'units':>5} {'start':>8} {'end':>8} {'span':>6} | {'issue':>6} {'wait':>6} {'stage':>6} {'kloop':>6} {'epi':>6} {'arr':>5} | {'idle->start':>10}  label")
tot = 0.0
prev_end = t0
for p in range(n):
    m = d[p, :, 0] > 0
    if not m.any():
        continue
    st = d[p, m]
    start, end = st[:, 0].min() - t0, st[:, 6].max() - t0
    seg = [np.mean(st[:, i + 1] - st[:, i]) for i in range(6)]
    print(f"{p:3d} {int(m.sum()):5d} {start:8.2f} {end:8.2f} {end - (prev_end - t0):6.2f} | " + " ".join(f"{v:6.2f}" for v in seg[:5]) + f" {seg[5]:5.2f} | "
          f"{np.mean(st[:, 2] - st[:, 0]):10.2f}  {prog.labels[p]}")
    prev_end = st[:, 6].max()
print(f"# whole launch: {prev_end - t0:.1f} us")
# critical path of the data-is-the-flag protocol: stamp 2 = the unit's polled loads came back complete, stamp 15 = its epilogue stores
# are issued.  exchange = last store of phase p-1 -> loads complete in phase p (first / mean / last unit); compute = 2 -> 15 per unit
print("# chain: phase | exchange first / mean / last consumer | compute 2->15 mean / max | 2->3 stage 3->4 kloop 4->14 reduce-barrier 14->15 epilogue | 2->11 raw part staged 11->13 normalised part stored 13->3 barrier")
ex_l, cp_l = [], []
for p in range(1, n):
    m, mp = d[p, :, 2] > 0, d[p - 1, :, 15] > 0
    if not m.any() or not mp.any():
        continue
    last_store = d[p - 1, mp, 15].max()
    v = d[p, m, 2] - last_store
    c = d[p, m, 15] - d[p, m, 2] if (d[p, m, 15] > 0).all() else d[p, m, 5] - d[p, m, 2]
    ex_l.append(v.max()); cp_l.append(c.max())
    seg = [np.mean(d[p, m, b] - d[p, m, a]) if (d[p, m, b] > 0).all() and (d[p, m, a] > 0).all() else float("nan") for a, b in ((2, 3), (3, 4), (4, 14), (14, 15), (2, 11), (11, 13), (13, 3))]
    print(f"{p:3d} | {v.min():6.2f} {v.mean():6.2f} {v.max():6.2f} | {c.mean():6.2f} {c.max():6.2f} | " + " ".join(f"{x:5.2f}" for x in seg) + f"  {prog.labels[p][:60]}")
print(f"# chain totals: sum of (exchange to the last consumer) {sum(ex_l):.1f} us, sum of (slowest unit's compute) {sum(cp_l):.1f} us over {len(ex_l)} phases")
if "JEN1_DEEP_PROFILE_NORM" in args.defs:
    # slots 7..9 restamped inside the normalised part: 11->7 own sums, 7->8 lane-set sums + statistics, 8->9 normalise + SiLU + LDS stores
    print("# norm part: phase | 11->7 sums 7->8 lane-set reduction 8->9 normalise+store 9->13 further trips")
    for p in range(1, n):
        m = (d[p, :, 11] > 0) & (d[p, :, 7] > d[p, :, 11]) & (d[p, :, 9] > 0)
        if not m.any():
            continue
        seg = [np.mean(d[p, m, b] - d[p, m, a]) for a, b in ((11, 7), (7, 8), (8, 9), (9, 13))]
        print(f"{p:3d} | " + " ".join(f"{x:5.2f}" for x in seg) + f"  {prog.labels[p][:60]}")
raw = dbg.cpu().numpy().astype(np.float64)
mm = (raw[:, :, 7] > 0) & (raw[:, :, 8] > 0) & (raw[:, :, 6] > raw[:, :, 0])
fr = (raw[:, :, 8] - raw[:, :, 7])[mm] / ((raw[:, :, 6] - raw[:, :, 0])[mm] * 0.01)
print(f"# shader clock during the GEMM units (s_memtime ticks per us): median {np.median(fr):.0f} MHz, 10th / 90th percentile {np.percentile(fr, 10):.0f} / {np.percentile(fr, 90):.0f}")
# pre-wait setup of the GEMM units (mean over the workgroups of a phase): 0->7 wave / geometry scalars, 7->8 normalised-part
# addresses + parameter requests, 8->9 raw-part addresses, 9->10 halo rows zeroed, 10->12 epilogue operands, 12->1 K-loop offsets
order = [0, 7, 8, 9, 10, 12, 1]
print("# setup: " + " ".join(f"{a}>{b}" for a, b in zip(order[:-1], order[1:])))
for p in range(n):
    m = (d[p, :, 0] > 0) & (d[p, :, 7] > 0)
    if not m.any():
        continue
    st = d[p, m]
    vals = [np.mean(st[:, b] - st[:, a]) for a, b in zip(order[:-1], order[1:])]
    print(f"{p:3d} " + " ".join(f"{v:5.2f}" for v in vals) + "  " + prog.labels[p][:70])

# attention units: 2->3 operands -> LDS (+ LayerNorm statistics), 3->10 scores (+ barrier), 10->11 V^T into LDS, 11->12 softmax (+ barrier),
# 12->4 P V (+ barrier), 4->5 store + drain
order = [2, 3, 10, 11, 12, 4, 5]
print("# attention: " + " ".join(f"{a}>{b}" for a, b in zip(order[:-1], order[1:])))
for p in range(n):
    if not prog.labels[p].startswith("attention"):
        continue
    m = (d[p, :, 0] > 0) & (d[p, :, 10] > 0)
    if not m.any():
        continue
    st = d[p, m]
    vals = [np.mean(st[:, b] - st[:, a]) for a, b in zip(order[:-1], order[1:])]
    print(f"{p:3d} " + " ".join(f"{v:5.2f}" for v in vals) + "  " + prog.labels[p][:70])

# tile units: 0 start, 1 addresses ready (first polled round goes out), 7 weight ring / parameters requested, 2 statistics partials + first batch complete, 8 affine
# tables, 11 tile staged, 3 barrier, 4 MFMA loop, 15 epilogue stores + output partials, 5 end
order = [0, 1, 7, 2, 8, 11, 3, 4, 9, 10, 12, 13, 15, 5]
print("# (4>9 epilogue addresses + residual, 9>10 bias / stores, 10>12 output partials -> LDS, 12>13 barrier, 13>15 partial store)")
print("# tile: " + " ".join(f"{a}>{b}" for a, b in zip(order[:-1], order[1:])) + " | unit total, units")
for p in range(n):
    if not prog.labels[p].startswith("tile"):
        continue
    m = (d[p, :, 0] > 0) & (d[p, :, 5] > 0)
    if not m.any():
        continue
    st = d[p, m]
    vals = [np.mean(st[:, b] - st[:, a]) for a, b in zip(order[:-1], order[1:])]
    print(f"{p:3d} " + " ".join(f"{v:5.2f}" for v in vals) + f" | {np.mean(st[:, 5] - st[:, 0]):5.2f} {int(m.sum()):4d}  " + prog.labels[p][:90])
