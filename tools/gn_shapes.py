#!/usr/bin/env python3
"""Per-launch time of the GroupNorm (+ SiLU) forward / backward of single layers inside a replayed chain (tuning tool).

    python tools/gn_shapes.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jen-1-pytorch_amd"))
import torch  # noqa: E402
from jen1_amd.graphs import capture as capture_graph  # noqa: E402

from jen1_amd import train as T  # noqa: E402

rt = T.TrainRuntime("bf16", "cuda")
N = 50


def timed(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with capture_graph(g):
            for _ in range(N):
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


print(f"{'B':>3} {'L':>5} {'C':>5} {'G':>3}  fwd_us  bwd_us   MB")
for B, Lx, C, G in [(16, 1500, 128, 8), (16, 1500, 256, 8), (16, 375, 256, 8), (16, 375, 512, 8), (16, 94, 512, 8), (16, 24, 512, 8), (16, 1, 1024, 8)]:
    x = torch.randn(B, Lx, C, device="cuda").to(torch.bfloat16)
    dy = torch.randn(B, Lx, C, device="cuda").to(torch.bfloat16)
    gamma = torch.nn.Parameter(torch.ones(C, device="cuda"))
    beta = torch.nn.Parameter(torch.zeros(C, device="cuda"))
    gamma.grad = torch.zeros_like(gamma)
    beta.grad = torch.zeros_like(beta)
    sums = torch.empty(B, G, 2, device="cuda")
    y = torch.empty_like(x)
    dx = torch.empty_like(x)
    P = torch.empty(B, C, 4, device="cuda")
    Gm = torch.empty(B, G, 2, device="cuda")
    L = T.L
    fwd = lambda: L.check(rt.lib.jen1_gn_forward(x.data_ptr(), sums.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, 0, y.data_ptr(),
                                                 B, Lx, C, C, G, 1e-5, 1, rt.dt, rt.stream()), "fwd")
    bwd = lambda: L.check(rt.lib.jen1_gn_backward(dy.data_ptr(), x.data_ptr(), sums.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, 0,
                                                  dx.data_ptr(), gamma.grad.data_ptr(), beta.grad.data_ptr(), None, P.data_ptr(),
                                                  Gm.data_ptr(), B, Lx, C, C, G, 1e-5, 1, rt.dt, rt.stream()), "bwd")
    print(f"{B:3d} {Lx:5d} {C:5d} {G:3d}  {timed(fwd):6.1f}  {timed(bwd):6.1f}  {x.numel() * 2 / 1e6:5.1f}")
