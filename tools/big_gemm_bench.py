#!/usr/bin/env python3
"""Tuning tool: jen1_big_gemm on the shapes of the hot path, replayed in a HIP graph (launch-to-launch time), against torch.matmul
(hipBLASLt) on the same operands.   python tools/big_gemm_bench.py > gpurun_out/big_gemm_bench.txt"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jen-1-pytorch_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from jen1_amd.graphs import capture as capture_graph  # noqa: E402
from jen1_amd import lib as L  # noqa: E402

lib = L.load()


def graph_us(fn, R=50):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with capture_graph(g):
            for _ in range(10):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(R):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (R * 10) * 1e3


for (M, N, K, label) in [(2064, 2048, 1024, "to_kv fwd C=1024 (2B*129 rows)"), (2064, 1024, 1024, "to_kv fwd C=512"), (2064, 512, 1024, "to_kv fwd C=256"),
                         (2064, 1024, 2048, "to_kv dgrad C=1024"), (1024, 17408, 1024, "set_context: 13 layers stacked, B=8"),
                         (2048, 17408, 1024, "set_context B=16"), (4096, 4096, 4096, "4096^3"), (8192, 8192, 8192, "8192^3")]:
    a = (torch.randn((M, K), device="cuda") * 0.5).to(torch.bfloat16)
    b = (torch.randn((N, K), device="cuda") * 0.5).to(torch.bfloat16)
    c = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    tab = L.bgemm_group_table([(c.data_ptr(), None, 0, N, N)], "cuda")
    g = L.BGemmArgs()
    g.a, g.b, g.groups, g.M, g.Ntot, g.K, g.lda, g.ldb, g.n_groups, g.dtype, g.alpha = a.data_ptr(), b.data_ptr(), tab.data_ptr(), M, N, K, K, K, 1, L.BF16, 1.0

    def own():
        L.check(lib.jen1_big_gemm(C.byref(g), torch.cuda.current_stream().cuda_stream), "big_gemm")

    def blas():
        torch.matmul(a, b.t(), out=c)
    t_own, t_blas = graph_us(own), graph_us(blas)
    forms = {}
    os.environ["JEN1_BGEMM_S4"] = "0"
    for form in ("0", "1"):                 # 128 x 128 / 256 x 128 only; 256 x 256 tiles forced (auto: the dispatcher's choice, above)
        os.environ["JEN1_BGEMM_T256"] = form
        forms[form] = graph_us(own)
    del os.environ["JEN1_BGEMM_T256"]
    forms["272"] = float("nan")
    if N % 272 == 0:                        # the four-stage 256 x 272 form
        os.environ["JEN1_BGEMM_S4"] = "272"
        forms["272"] = graph_us(own)
    del os.environ["JEN1_BGEMM_S4"]
    fl = 2.0 * M * N * K
    print(f"{label:40s} M={M:5d} N={N:5d} K={K:5d}  own {t_own:8.1f} us = {fl / t_own / 1e6:7.1f} TF/s ({fl / t_own / 1e6 / 2500 * 100:4.1f} % of peak)   "
          f"[small tiles only {forms['0']:7.1f} us, 256x256 only {forms['1']:7.1f}, 4-stage 256x272 {forms['272']:7.1f}]   hipBLASLt {t_blas:8.1f} us = {fl / t_blas / 1e6:7.1f} TF/s", flush=True)

for (M, N, K, label) in [(2064, 2048, 1024, "to_kv wgrad C=1024"), (2064, 1024, 1024, "to_kv wgrad C=512"), (2064, 512, 1024, "to_kv wgrad C=256")]:
    a = (torch.randn((M, N), device="cuda") * 0.5).to(torch.bfloat16)
    b = (torch.randn((M, K), device="cuda") * 0.5).to(torch.bfloat16)
    c = torch.zeros((N, K), device="cuda")

    def own():
        L.check(lib.jen1_big_gemm_tn(a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, N, K, K, 1.0, torch.cuda.current_stream().cuda_stream), "tn")

    def blas():
        c.add_(torch.matmul(a.t(), b))
    t_own, t_blas = graph_us(own), graph_us(blas)
    fl = 2.0 * M * N * K
    print(f"{label:40s} M={M:5d} N={N:5d} K={K:5d}  own {t_own:8.1f} us = {fl / t_own / 1e6:7.1f} TF/s ({fl / t_own / 1e6 / 2500 * 100:4.1f} % of peak)   "
          f"hipBLASLt + add {t_blas:8.1f} us = {fl / t_blas / 1e6:7.1f} TF/s", flush=True)
