#!/usr/bin/env python3
"""Per-launch time of a deep-level forward product with WARM weights (the same weight every launch: L2 / infinity-cache hits) against
COLD ones (a different copy every launch, 64 copies > the 256 MB infinity cache) -- what the replayed pass sees (tuning tool).

    python tools/cold_weights.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jen-1-pytorch_amd"))
import torch  # noqa: E402
from jen1_amd.graphs import capture as capture_graph  # noqa: E402

from jen1_amd import train as T  # noqa: E402

rt = T.TrainRuntime("bf16", "cuda")
N = 64


def timed(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with capture_graph(g):
            fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * N)


print(f"{'rows':>6} {'ci':>5} {'co':>5} k  warm_us  cold_us")
for rows, Lx, ci, co, k in [(16, 1, 1024, 1024, 1), (16, 1, 1024, 1024, 3), (32, 2, 1024, 1024, 3), (192, 12, 512, 512, 3), (96, 6, 512, 512, 1)]:
    B = rows // Lx
    kind = "linear" if k == 1 else "conv"
    x = torch.randn(B, Lx, ci, device="cuda").to(torch.bfloat16)
    g = T.ConvGeom(kind, k, 1, (k - 1) // 2, Lx, Lx, ci, co) if k > 1 else T.ConvGeom("linear", 1, 1, 0, rows, rows, ci, co)
    if k == 1:
        x = x.view(1, rows, ci)
    wps = [torch.randn(k, co, ci, device="cuda").to(torch.bfloat16) for _ in range(N)]
    bias = torch.zeros(co, device="cuda")
    warm = timed(lambda: [T._conv_forward(rt, x, wps[0], bias, g) for _ in range(N)])
    cold = timed(lambda: [T._conv_forward(rt, x, wps[i], bias, g) for i in range(N)])
    print(f"{rows:6d} {ci:5d} {co:5d} {k}  {warm:7.1f}  {cold:7.1f}")
