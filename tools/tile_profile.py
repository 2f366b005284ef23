#!/usr/bin/env python3
"""Stage timeline of the long-level conv launches (tuning tool, not part of the product): builds a copy of the library with
-DJEN1_TILE_PROFILE (every workgroup of tile_gemm_kernel stamps the 100 MHz counter), runs each conv launch of the bench plan
alone and prints per launch: workgroups, first start -> last end, and the mean duration of the stages.

    python tools/tile_profile.py [--batch 8] [--length 1500] [--cfg] > gpurun_out/tile_profile.txt
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
ap.add_argument("--defs", default="")
args = ap.parse_args()
out_lib = os.path.join(ROOT, "gpurun_out", "libjen1_hip_tprof.so")
os.makedirs(os.path.dirname(out_lib), exist_ok=True)
csrc = os.path.join(ROOT, "jen-1-pytorch_amd", "csrc")
os.environ["JEN1_LIB"] = out_lib
from jen1_amd import lib as L  # noqa: E402
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DJEN1_TILE_PROFILE", *args.defs.split(),
                f"-I{os.path.join(ROOT, 'include')}", f"-I{csrc}", *[os.path.join(csrc, s) for s in L.SOURCES], "-o", out_lib], check=True)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from jen1_amd import synth  # noqa: E402
from jen1_amd.config import full_model_config  # noqa: E402
from jen1_amd.model import UNetCFG1d  # noqa: E402

lib = L.load()
lib.jen1_tile_debug_buffer.restype = C.c_int
lib.jen1_tile_debug_buffer.argtypes = [C.c_void_p]
dev = "cuda"
model = UNetCFG1d(**full_model_config(), init_seed=1234, compute_dtype=args.dtype, device=dev)
B, T = args.batch, args.length
plan = model.engine().plan(B, T, 2 if args.cfg else 1, False, n_t=100)
x, cond = synth.latents(B, T), synth.conditioning(B, T)
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
plan.t_in.copy_(tt(np.array([(131 * i + 7) % 1000 for i in range(plan.n_t)], dtype=np.int64)))
model._prepare(plan, tt(x), None, tt(cond["cross_attn_cond"]), tt(cond["cross_attn_masks"]), [tt(cond["input_concat_cond"])], None)
s = torch.cuda.current_stream().cuda_stream
plan.run_time(s)
for _ in range(3):
    plan.run(s)
torch.cuda.synchronize()
dbg = torch.zeros((8192, 8), dtype=torch.int64, device=dev)
assert lib.jen1_tile_debug_buffer(dbg.data_ptr()) == 0
names = ["kernarg", "issue", "table", "stage", "mfma", "epi", "tail"]
print(f"# stages (us, mean over workgroups): kernarg = start -> hot block loaded; issue = ring + first batch requested; table = GroupNorm "
      f"tables; stage = barrier + normalise -> LDS + barrier; mfma = k loop; epi = bias / residual / stores; tail = statistics atomics + drain")
print(f"{'WGs':>5} {'span':>6} {'spread':>6} | " + " ".join(f"{n:>7}" for n in names) + " | label")
tot = 0.0
for op in plan.ops:
    if getattr(op, "kind", "") != "conv_gemm":
        continue
    for _ in range(2):
        op(s)
    torch.cuda.synchronize()
    dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    op(s)
    e1.record()
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().astype(np.float64) * 0.01
    m = d[:, 0] > 0
    if not m.any():
        print(f"{'-':>5} {e0.elapsed_time(e1) * 1e3:6.1f} (not a tile_gemm launch) | {op.label[:100]}")
        continue
    st = d[m]
    span = st[:, 7].max() - st[:, 0].min()
    spread = st[:, 0].max() - st[:, 0].min()
    seg = [np.mean(st[:, i + 1] - st[:, i]) for i in range(7)]
    tot += span
    print(f"{int(m.sum()):5d} {span:6.2f} {spread:6.2f} | " + " ".join(f"{v:7.2f}" for v in seg) + f" | {op.label[:110]}")
print(f"# sum of device spans: {tot:.1f} us")
