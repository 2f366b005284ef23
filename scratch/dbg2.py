import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, 'jen-1-pytorch_amd')
import torch, torch.nn.functional as F
from test_gpu_kernels import to_cl, new_out, from_cl, pack_conv, run
from helpers import rel_err
from jen1_amd import lib as L
from jen1_amd.engine import OpBuilder, KernelCtx
kc = KernelCtx("f32")
torch.manual_seed(11)
B, C0, C1, Co, Ln, G = 3, 64, 64, 128, 29, 8
sc = 2 ** -0.5
x0 = torch.randn(B, C0, Ln, device="cuda") * 1.5 + 0.3
x1 = torch.randn(B, C1, Ln, device="cuda") * 0.7 - 0.2
Ct = C0 + C1
gam, bet = torch.rand(Ct, device="cuda") + 0.5, torch.randn(Ct, device="cuda") * 0.1
w = torch.randn(Co, Ct, 3, device="cuda") / (Ct * 3) ** 0.5
bias = torch.randn(Co, device="cuda") * 0.1
resid = torch.randn(B, Co, Ln, device="cuda")
xin = torch.cat([x0, x1 * sc], 1)
h = F.silu(F.group_norm(xin, G, gam, bet, 1e-5))
for name, causal, use_res, gn_out, rs_out, force in [
        ("causal", True, False, False, False, None), ("res", False, True, False, False, None),
        ("gnout", False, False, True, False, None), ("rsout", False, False, False, True, None),
        ("all", True, True, True, True, None), ("all sk1", True, True, True, True, {"splitk": 1}),
        ("res sk1", False, True, False, False, {"splitk": 1})]:
    ref = F.conv1d(F.pad(h, (2, 0) if causal else (1, 1)), w, bias) + (resid if use_res else 0)
    ob = OpBuilder(kc)
    out = new_out(kc, B, Ln, Co, gn=gn_out, rs=rs_out)
    ob.conv(ob.ops, src0=to_cl(x0, kc), src1=to_cl(x1, kc), src1_scale=sc, w=pack_conv(w, kc), bias=bias, out=out, taps=3,
            pad_left=2 if causal else 1, pro=L.PRO_GN_SILU, gn=(G, Ct, gam, bet, 1e-5), force=force,
            residual=to_cl(resid, kc) if use_res else None)
    a = ob._keep[-1][0]
    run(ob)
    y = from_cl(out)
    print(name, "splitk", a.splitk, "cfg", a.cfg, "tb", a.tb, "nb", a.nb, "err", rel_err(y.cpu().numpy(), ref.cpu().numpy()))
