import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, 'jen-1-pytorch_amd')
import torch
from jen1_amd import lib as L
from jen1_amd.engine import OpBuilder, KernelCtx, Act
from jen1_amd.packing import pack_gemm_weight, conv_weight_to_gemm
kc = KernelCtx("bf16")
dev = "cuda"
def bench(B, Ln, Ci, Co, taps, pro, stats, res, reps=200, cfg=None, label=""):
    ob = OpBuilder(kc)
    w = pack_gemm_weight(conv_weight_to_gemm(torch.randn(Co, Ci, taps, device=dev) * 0.05), torch.bfloat16)
    gam, bet = torch.ones(Ci, device=dev), torch.zeros(Ci, device=dev)
    bias = torch.randn(Co, device=dev)
    xs = [Act(torch.randn(B, Ln, Ci, device=dev).to(torch.bfloat16), B, Ln, Ci, Ci, torch.rand(B * 64, device=dev) * 100 + 50) for _ in range(4)]
    for a in xs:   # plausible stats: sum, sumsq
        a.gn.view(B, 32, 2)[:, :, 0] = 0.0; a.gn.view(B, 32, 2)[:, :, 1] = float(Ln * Ci // 32)
    for i in range(reps):
        out = Act(torch.zeros(B, Ln, Co, device=dev, dtype=torch.bfloat16), B, Ln, Co, Co, torch.zeros(B * 64, device=dev) if stats else None)
        r = Act(torch.zeros(B, Ln, Co, device=dev, dtype=torch.bfloat16), B, Ln, Co, Co) if res else None
        ob.conv(ob.ops, src0=xs[i % 4], w=w, bias=bias, out=out, taps=taps, pad_left=(taps - 1) // 2, residual=r, pro=pro,
                gn=(8, Ci, gam, bet, 1e-5) if pro in (L.PRO_GN, L.PRO_GN_SILU) else None, force=cfg)
    ob.finalize_workspace()
    ob.run(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s): ob.run(s.cuda_stream)
    torch.cuda.synchronize()
    with torch.cuda.graph(g): ob.run(torch.cuda.current_stream().cuda_stream)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5 / reps * 1e6
    a = [k[0] for k in ob._keep if isinstance(k, tuple) and hasattr(k[0], "cfg")][0]
    fl = 2 * taps * Co * Ci * B * Ln
    print(f"{label:28s} B={B} L={Ln} {Ci}->{Co} k={taps} pro={pro} stats={stats} res={res} cfg={a.cfg} tb={a.tb} direct={a.direct} kst={a.kc_stage}: {dt:7.2f} us  {fl/dt/1e6:6.1f} TF/s")
bench(8, 1500, 128, 128, 3, L.PRO_NONE, False, False, label="plain conv")
bench(8, 1500, 128, 128, 3, L.PRO_NONE, True, False, label="+stats")
bench(8, 1500, 128, 128, 3, L.PRO_NONE, True, True, label="+stats+res")
bench(8, 1500, 128, 128, 3, L.PRO_GN, True, True, label="+GN")
bench(8, 1500, 128, 128, 3, L.PRO_GN_SILU, True, True, label="+GN+SiLU (resblock conv)")
bench(8, 1500, 128, 128, 3, L.PRO_GN_SILU, True, True, cfg={"cfg": 1}, label="same, W128x64")
bench(8, 1500, 128, 128, 1, L.PRO_NONE, False, False, label="1x1 plain")
bench(8, 1500, 288, 128, 3, L.PRO_GN_SILU, True, False, label="to_in conv1 (288)")
bench(8, 375, 128, 128, 3, L.PRO_GN_SILU, True, True, label="level1 conv")
bench(8, 375, 128, 128, 3, L.PRO_GN_SILU, True, True, cfg={"cfg": 0}, label="level1 conv forced wide")
