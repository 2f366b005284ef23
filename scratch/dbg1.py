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
xin = torch.cat([x0, x1 * sc], 1)
for name, pro, force in [("raw sk1", L.PRO_NONE, {"splitk": 1}), ("raw sk2", L.PRO_NONE, {"splitk": 2}),
                         ("gn sk1", L.PRO_GN_SILU, {"splitk": 1}), ("gn sk2", L.PRO_GN_SILU, {"splitk": 2}),
                         ("gn nosilu sk1", L.PRO_GN, {"splitk": 1})]:
    if pro == L.PRO_NONE:
        ref = F.conv1d(F.pad(xin, (1, 1)), w, bias)
    else:
        h = F.group_norm(xin, G, gam, bet, 1e-5)
        if pro == L.PRO_GN_SILU: h = F.silu(h)
        ref = F.conv1d(F.pad(h, (1, 1)), w, bias)
    ob = OpBuilder(kc)
    out = new_out(kc, B, Ln, Co)
    ob.conv(ob.ops, src0=to_cl(x0, kc), src1=to_cl(x1, kc), src1_scale=sc, w=pack_conv(w, kc), bias=bias, out=out, taps=3,
            pad_left=1, pro=pro, gn=(G, Ct, gam, bet, 1e-5) if pro != L.PRO_NONE else None, force=force)
    run(ob)
    y = from_cl(out)
    e_all = rel_err(y.cpu().numpy(), ref.cpu().numpy())
    # which half is wrong? zero the contribution of each source in the reference
    print(name, "err", e_all)
