import sys, os
os.environ["JEN1_LIB"] = os.path.abspath("jen-1-pytorch_amd/jen1_amd/libjen1_prof.so")
sys.path.insert(0, 'tests'); sys.path.insert(0, 'jen-1-pytorch_amd')
import torch
from jen1_amd import lib as L
from jen1_amd.engine import OpBuilder, KernelCtx, Act
from jen1_amd.packing import pack_gemm_weight, conv_weight_to_gemm
kc = KernelCtx("bf16")
dev = "cuda"
def prof(B, Ln, Ci, Co, taps, pro, stats, res, cfg=None, label=""):
    ob = OpBuilder(kc)
    w = pack_gemm_weight(conv_weight_to_gemm(torch.randn(Co, Ci, taps, device=dev) * 0.05), torch.bfloat16)
    gam, bet = torch.ones(Ci, device=dev), torch.zeros(Ci, device=dev)
    bias = torch.randn(Co, device=dev)
    x = Act(torch.randn(B, Ln, Ci, device=dev).to(torch.bfloat16), B, Ln, Ci, Ci, torch.zeros(B * 64, device=dev))
    x.gn.view(B, 32, 2)[:, :, 1] = float(Ln * Ci // 32)
    out = Act(torch.zeros(B, Ln, Co, device=dev, dtype=torch.bfloat16), B, Ln, Co, Co, torch.zeros(B * 64, device=dev) if stats else None)
    r = Act(torch.zeros(B, Ln, Co, device=dev, dtype=torch.bfloat16), B, Ln, Co, Co) if res else None
    ob.conv(ob.ops, src0=x, w=w, bias=bias, out=out, taps=taps, pad_left=(taps - 1) // 2, residual=r, pro=pro,
            gn=(8, Ci, gam, bet, 1e-5) if pro in (L.PRO_GN, L.PRO_GN_SILU) else None, force=cfg)
    a = [k[0] for k in ob._keep if isinstance(k, tuple) and hasattr(k[0], "cfg")][0]
    dbg = torch.zeros(16, dtype=torch.int64, device=dev)
    a.slab = dbg.data_ptr()
    res_ = []
    for it in range(5):
        ob.run(); torch.cuda.synchronize()
        st = dbg.cpu().tolist()
        res_.append([((st[i] - st[0]) / 100.0 if st[i] else float("nan")) for i in range(9)])   # 100 MHz -> us
    med = [sorted(r[i] for r in res_)[2] for i in range(9)]
    names = (["entry", "ring filled", "epilogue operands requested", "first mfma done", "main loop done", "reduce/splitk done", "epilogue stores done", "end", "-"] if a.direct else ["entry", "setup done", "phase0 loads issued", "tables/barrier1", "tile staged+barrier2", "MFMA loop done", "reduce/splitk done", "epilogue done", "end"])
    print(f"{label}: cfg={a.cfg} direct={a.direct} tb={a.tb} nb={a.nb}")
    for n, v, pv in zip(names, med, [0] + med[:-1]):
        print(f"    {n:44s} t={v:7.2f} us   (+{v - pv:5.2f})")
prof(8, 1, 1024, 1024, 1, L.PRO_NONE, True, True, cfg={"splitk":1}, label="L=1 1x1 1024->1024 (8 rows)")
prof(8, 6, 512, 512, 3, L.PRO_NONE, True, True, cfg={"splitk":1}, label="L=6 k3 512->512 (48 rows, 12-row tiles)")
prof(8, 6, 512, 1536, 1, L.PRO_NONE, False, False, cfg={"splitk":1}, label="L=6 qkv 512->1536")
prof(8, 24, 256, 256, 3, L.PRO_NONE, True, True, cfg={"splitk":1}, label="L=24 k3 256->256")
prof(8, 2, 2048, 1024, 3, L.PRO_NONE, True, False, cfg={"splitk":1}, label="L=2 k3 2048->1024")
