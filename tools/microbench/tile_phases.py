import sys, os
os.environ["JEN1_LIB"] = os.path.abspath("jen-1-pytorch_amd/jen1_amd/libjen1_prof.so")
sys.path.insert(0, 'tests'); sys.path.insert(0, 'jen-1-pytorch_amd')
import torch
from jen1_amd import lib as L
from jen1_amd.engine import OpBuilder, KernelCtx, Act
from jen1_amd.packing import pack_gemm_weight, conv_weight_to_gemm
kc = KernelCtx("bf16"); dev = "cuda"
names = ["entry", "hot args", "weight ring issued", "first staging batch issued", "GN tables built", "epi args + barrier", "tile staged + barrier", "MFMA loop done", "epilogue + stats done"]
def prof(B, Ln, Ci, Co, taps, stride, pro, two, label, cfg=None):
    ob = OpBuilder(kc)
    C1 = Ci if two else 0
    w = pack_gemm_weight(conv_weight_to_gemm(torch.randn(Co, Ci + C1, taps, device=dev) * 0.05), torch.bfloat16)
    gam, bet = torch.ones(Ci + C1, device=dev), torch.zeros(Ci + C1, device=dev)
    bias = torch.randn(Co, device=dev)
    mk = lambda C: Act(torch.randn(B, Ln, C, device=dev).to(torch.bfloat16), B, Ln, C, C, torch.rand(B * 64, device=dev) + 1)
    x0, x1 = mk(Ci), (mk(C1) if two else None)
    Lo = -(-Ln // stride)
    out = Act(torch.zeros(B, Lo, Co, device=dev, dtype=torch.bfloat16), B, Lo, Co, Co, torch.zeros(B * 64, device=dev))
    film = torch.randn(4, 2 * (Ci + C1), device=dev); step = torch.zeros(1, dtype=torch.int32, device=dev)
    res = Act(torch.zeros(B, Lo, Co, device=dev, dtype=torch.bfloat16), B, Lo, Co, Co) if stride == 1 else None
    ob.conv(ob.ops, src0=x0, src1=x1, src1_scale=0.7 if two else 1.0, w=w, bias=bias, out=out, taps=taps, stride=stride, pad_left=(taps - 1) // 2, L_out=Lo,
            residual=res, pro=pro, gn=(8, Ci + C1, gam, bet, 1e-5) if pro else None, film=(film, None, 0, Ci + C1, step) if pro else None,
            force={"cfg": cfg} if cfg else None)
    a = [k[0] for k in ob._keep if isinstance(k, tuple) and hasattr(k[0], "cfg")][0]
    dbg = torch.zeros(16, dtype=torch.int64, device=dev); a.slab = dbg.data_ptr()
    rs = []
    for it in range(5):
        ob.run(); torch.cuda.synchronize(); st = dbg.cpu().tolist(); rs.append([(st[i] - st[0]) / 100.0 for i in range(9)])
    med = [sorted(r[i] for r in rs)[2] for i in range(9)]
    print(f"{label}: cfg={a.cfg} tb={a.tb}")
    pv = 0
    for n, v in zip(names, med):
        print(f"    {n:30s} t={v:6.2f} (+{v - pv:5.2f})"); pv = v
prof(8, 1500, 128, 128, 3, 1, L.PRO_GN_SILU, False, "T=1500 resblock conv 128->128 k3 GN+FiLM+SiLU +res")
prof(8, 1500, 128, 128, 3, 1, L.PRO_GN_SILU, True, "T=1500 up-path conv1 [128|128]->128 k3")
prof(8, 1500, 128, 128, 9, 4, 0, False, "down conv k9 s4 1500->375")
prof(8, 375, 128, 128, 3, 1, L.PRO_GN_SILU, False, "T=375 resblock conv")
prof(8, 94, 256, 256, 3, 1, L.PRO_GN_SILU, False, "T=94 resblock conv 256")
