#!/usr/bin/env python3
"""Per-phase timeline of the sample-resident long-level launches (tuning tool, not part of the product).

Builds a second copy of the library with -DJEN1_LONG_PROFILE (thread 0 of every workgroup stamps the 100 MHz counter at the stages
of each unit), replays one denoiser step of the bench workload, then each long-level program alone, and prints per phase the mean
duration of the stages over the workgroups that had a unit:
  0 unit start | 1 polled round issued | 2 polled round complete (wave 0) | 3 statistics reduced + barrier (= every wave's poll complete) |
  4 affine tables (+ barrier) | 5 tile staged (+ barrier) | 6 MFMA loop done | 7 epilogue stores issued

    python tools/long_profile.py [--batch 8] [--length 1500] [--cfg] [--dtype bf16] > gpurun_out/long_profile.txt
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
ap.add_argument("--defs", default="", help="extra -D flags for the tuning build, space separated")
ap.add_argument("--build-only", action="store_true")
args = ap.parse_args()

# the tuning library is built where hipcc and the product objects are (the build container: ``--build-only``), in-tree so that it
# travels to the GPU box with the snapshot (jen1_amd/libjen1_hip_longprof.so; *.so is git-ignored)
out_lib = os.path.join(ROOT, "jen-1-pytorch_amd", "jen1_amd", "libjen1_hip_longprof.so")
csrc = os.path.join(ROOT, "jen-1-pytorch_amd", "csrc")
os.environ["JEN1_LIB"] = out_lib
from jen1_amd import lib as L  # noqa: E402
src = os.path.join(csrc, "long_kernel.hip")
if not os.path.exists(out_lib) or os.path.getmtime(out_lib) < os.path.getmtime(src) or args.defs or args.build_only:
    objdir = os.path.join(ROOT, "jen-1-pytorch_amd", "build", "obj")
    objs = [os.path.join(objdir, s + ".o") for s in L.SOURCES if s != "long_kernel.hip"]
    if not all(os.path.exists(o) for o in objs):       # (the product objects of every other source are reused when they are there)
        objs = [os.path.join(csrc, s) for s in L.SOURCES if s != "long_kernel.hip"]
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DJEN1_LONG_PROFILE", *args.defs.split(), f"-I{os.path.join(ROOT, 'include')}", f"-I{csrc}"]
    if objs[0].endswith(".o"):
        pobj = os.path.join(objdir, "long_kernel.prof.o")
        subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-c", src, "-o", pobj], check=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", pobj, *objs, "-o", out_lib], check=True)
    else:
        subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-shared", src, *objs, "-o", out_lib], check=True)
if args.build_only:
    sys.exit(0)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from jen1_amd import synth  # noqa: E402
from jen1_amd.config import full_model_config  # noqa: E402
from jen1_amd.model import UNetCFG1d  # noqa: E402

lib = L.load()
dev = "cuda"
model = UNetCFG1d(**full_model_config(), init_seed=1234, compute_dtype=args.dtype, device=dev)
B, T = args.batch, args.length
nrep = 2 if args.cfg else 1
plan = model.engine().plan(B, T, nrep, False, n_t=100)
assert plan.long_levels >= 1, plan.long_errors
x, cond = synth.latents(B, T), synth.conditioning(B, T)
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
t = np.array([(131 * i + 7) % 1000 for i in range(plan.n_t)], dtype=np.int64)
plan.t_in.copy_(tt(t))
model._prepare(plan, tt(x), tt(t), tt(cond["cross_attn_cond"]), tt(cond["cross_attn_masks"]), [tt(cond["input_concat_cond"])], None)
s = torch.cuda.current_stream().cuda_stream
plan.run_time(s)
for rep in range(args.reps):
    plan.run(s)
torch.cuda.synchronize()
assert plan.take_error() == 0
names = ["setup", "poll", "bar1", "tables", "stage", "mfma", "epilogue"]
for pi, prog in enumerate(plan.progs):
    if prog.kinds[0] != "long":
        continue
    n, nwg = len(prog), prog.nwg
    dbg = torch.zeros((n, nwg, 8), dtype=torch.int64, device=dev)
    for rep in range(3):
        prog.poison(s)
        prog.launch(s)
    torch.cuda.synchronize()
    assert lib.jen1_long_debug_buffer(dbg.data_ptr()) == 0
    dbg.zero_()
    prog.poison(s)
    torch.cuda.synchronize()
    prog.launch(s)
    torch.cuda.synchronize()
    assert lib.jen1_long_debug_buffer(None) == 0
    assert prog.error() == 0
    d = dbg.cpu().numpy().astype(np.float64) * 0.01       # microseconds
    t0 = d[d[:, :, 0] > 0][:, 0].min()
    print(f"# program {pi}: B={B} T={T} nrep={nrep} dtype={args.dtype}: {n} phases, {prog.Bs} samples x {prog.G} workgroups, {prog.lds} B LDS; times in us")
    print(f"{'ph':>3} {'units':>5} {'start':>8} {'end':>8} {'span':>6} | " + " ".join(f"{k:>8}" for k in names) + " | exchange (last store of p-1 -> poll complete: min / mean / max)  label")
    prev_end = t0
    for p in range(n):
        m = d[p, :, 0] > 0
        if not m.any():
            continue
        st = d[p, m]
        start, end = st[:, 0].min() - t0, st[:, 7].max() - t0
        seg = [np.mean(st[:, i + 1] - st[:, i]) for i in range(7)]
        ex = ""
        if p > 0 and (d[p - 1, :, 7] > 0).any():
            v = st[:, 2] - d[p - 1, d[p - 1, :, 7] > 0, 7].max()
            ex = f"{v.min():6.2f} {v.mean():6.2f} {v.max():6.2f}"
        print(f"{p:3d} {int(m.sum()):5d} {start:8.2f} {end:8.2f} {end - (prev_end - t0):6.2f} | " + " ".join(f"{v:8.2f}" for v in seg) + f" | {ex:>20}  {prog.labels[p][:110]}")
        prev_end = st[:, 7].max()
    print(f"# whole launch: {prev_end - t0:.1f} us")
