#!/usr/bin/env python3
"""Per-launch time of the weight-gradient / data-gradient / forward products of single layers (tuning tool): each shape is recorded 50
times into a HIP graph and replayed, so the figure is the steady-state cost of one launch in a replayed chain.

    python tools/wgrad_shapes.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jen-1-pytorch_amd"))
import torch  # noqa: E402
from jen1_amd.graphs import capture as capture_graph  # noqa: E402

from jen1_amd import train as T  # noqa: E402

rt = T.TrainRuntime("bf16", "cuda")
rt.wgrad_group = 0
SHAPES = [  # (rows = B * L, L, ci, co, taps)
    (16, 1, 1024, 1024, 1), (16, 1, 1024, 1024, 3), (32, 2, 1024, 1024, 3), (48, 3, 512, 512, 1), (192, 12, 512, 512, 1),
    (192, 12, 512, 512, 3), (384, 24, 256, 256, 3), (1504, 94, 256, 256, 3), (6000, 375, 128, 128, 3), (24000, 1500, 128, 128, 3),
]
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


print(f"{'rows':>6} {'ci':>5} {'co':>5} k   fwd_us  dgrad_us  wgrad_us  pair_us")
for rows, Lx, ci, co, k in SHAPES:
    B = rows // Lx
    kind = "linear" if k == 1 else "conv"
    x = torch.randn(B, Lx, ci, device="cuda").to(torch.bfloat16)
    dy = torch.randn(B, Lx, co, device="cuda").to(torch.bfloat16)
    w = torch.randn(*((co, ci) if k == 1 else (co, ci, k)), device="cuda")
    gw = torch.zeros_like(w)
    gb = torch.zeros(co, device="cuda")
    bias = torch.zeros(co, device="cuda")
    g = T.ConvGeom(kind, k, 1, (k - 1) // 2, Lx, Lx, ci, co) if k > 1 else T.ConvGeom("linear", 1, 1, 0, rows, rows, ci, co)
    if k == 1:
        x, dy = x.view(1, rows, ci), dy.view(1, rows, co)
    wp = rt.packed(w, kind, torch.bfloat16)
    wd = rt.packed(w, kind + "D", torch.bfloat16)
    t_f = timed(lambda: T._conv_forward(rt, x, wp, bias, g))
    t_d = timed(lambda: T._conv_dgrad(rt, dy, wp, g, wd))
    t_w = timed(lambda: T._conv_wgrad(rt, x, dy, gw, g, gb))

    def pair():
        _, blk = T._conv_wgrad(rt, x, dy, gw, g, gb, defer=True)
        T._conv_dgrad(rt, dy, wp, g, wd, pair_with=blk)
    t_p = timed(pair)
    print(f"{rows:6d} {ci:5d} {co:5d} {k}  {t_f:7.1f}  {t_d:8.1f}  {t_w:8.1f}  {t_p:7.1f}")
