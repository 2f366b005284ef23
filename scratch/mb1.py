import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, 'jen-1-pytorch_amd')
import torch
from jen1_amd import lib as L
from jen1_amd.engine import OpBuilder, KernelCtx, Act
from jen1_amd.packing import pack_gemm_weight
kc = KernelCtx("bf16")
dev = "cuda"
def bench(M, K, rows, taps, nw, reps=400, res=True, stats=True):
    ob = OpBuilder(kc)
    ws = [pack_gemm_weight(torch.randn(taps, M, K, device=dev) * 0.02, torch.bfloat16) for _ in range(nw)]
    x = Act(torch.randn(rows, 1, K, device=dev).to(torch.bfloat16), rows, 1, K, K)
    bias = torch.randn(M, device=dev)
    outs = []
    for i in range(reps):
        out = Act(torch.zeros(rows, 1, M, device=dev, dtype=torch.bfloat16), rows, 1, M, M,
                  torch.zeros(rows * 64, device=dev) if stats else None)
        r = Act(torch.zeros(rows, 1, M, device=dev, dtype=torch.bfloat16), rows, 1, M, M) if res else None
        ob.conv(ob.ops, src0=x, w=ws[i % nw], bias=bias, out=out, taps=taps, pad_left=(taps - 1) // 2, residual=r)
    ob.finalize_workspace()
    ob.run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ob.run(s.cuda_stream)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        ob.run(torch.cuda.current_stream().cuda_stream)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5 / reps * 1e6
    a = ob._keep[0][0]
    print(f"M={M} K={K} rows={rows} taps={taps} nw={nw} res={res} stats={stats} cfg={a.cfg} direct={a.direct}: {dt:.2f} us/launch  ({taps*M*K*2/1e6:.2f} MB w -> {taps*M*K*2/dt/1e3:.0f} GB/s)")
for nw in (1, 64):
    bench(1024, 1024, 8, 1, nw)
    bench(1024, 1024, 8, 3, nw)
    bench(1024, 2048, 16, 3, nw)
    bench(512, 512, 48, 1, nw)
bench(1024, 1024, 8, 1, 64, res=False, stats=False)
bench(1024, 1024, 8, 3, 64, res=False, stats=False)
bench(256, 256, 8, 1, 64, res=False, stats=False)
