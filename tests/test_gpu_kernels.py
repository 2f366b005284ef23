"""-m gpu: kernel-level parity of libjen1_hip.so, called through the C ABI.

Each HIP kernel is compared with (a) a plain PyTorch float32 reference of the same op
evaluated on the GPU and (b) the golden unit fixtures produced by the reference.
float32 mode must agree to 1e-4 relative (exact-fp32 MFMA, only the summation order
differs); bf16 mode to 3e-2 (bf16 storage of activations and weights).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import SEED, golden, rel_err
from jen1_amd.init_fill import fill, fill_normal

pytestmark = pytest.mark.gpu

TOL = {"f32": 1e-4, "bf16": 3e-2}


def _skip_no_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


@pytest.fixture(scope="module", params=["f32", "bf16"])
def ctx(request):
    _skip_no_gpu()
    from jen1_amd.engine import KernelCtx
    return KernelCtx(request.param, "cuda", target_wgs=256), request.param


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def to_cl(x_bct: torch.Tensor, kc, ld=None):
    """[B,C,L] float32 -> Act (channel-last, padded to a multiple of 32 channels) with its statistics."""
    from jen1_amd.engine import Act
    B, Cc, Ln = x_bct.shape
    ld = ld or (Cc + 31) // 32 * 32
    t = torch.zeros((B, Ln, ld), dtype=kc.tdtype, device="cuda")
    t[:, :, :Cc] = x_bct.permute(0, 2, 1).to(kc.tdtype)
    tf = t.float()
    cpf = ld // 32
    g = tf.reshape(B, Ln, 32, cpf)
    gn = torch.stack([g.sum(dim=(1, 3)), (g * g).sum(dim=(1, 3))], dim=-1).reshape(B * 64).contiguous()
    rs = torch.stack([tf.sum(-1), (tf * tf).sum(-1)], dim=-1).reshape(-1).contiguous()
    return Act(t, B, Ln, Cc, ld, gn, rs)


def new_out(kc, B, Ln, Cc, gn=False, rs=False, f32=False):
    from jen1_amd.engine import Act
    ld = (Cc + 31) // 32 * 32
    t = torch.zeros((B, Ln, ld), dtype=torch.float32 if f32 else kc.tdtype, device="cuda")
    return Act(t, B, Ln, Cc, ld, torch.zeros(B * 64, device="cuda") if gn else None,
               torch.zeros(B * Ln * 2, device="cuda") if rs else None)


def from_cl(a):
    return a.t[:, :, : a.C].float().permute(0, 2, 1).contiguous()


def pack_conv(w, kc):
    from jen1_amd.packing import conv_weight_to_gemm, pack_gemm_weight
    return pack_gemm_weight(conv_weight_to_gemm(w), kc.tdtype)


def run(builder):
    builder.finalize_workspace()
    builder.run()
    torch.cuda.synchronize()


# ------------------------------------------------------------------ plain convs (a1)
@pytest.mark.parametrize("k,s", [(1, 1), (3, 1), (5, 2), (9, 4)])
@pytest.mark.parametrize("causal", [False, True])
def test_conv1d_matches_golden_and_torch(ctx, k, s, causal):
    from jen1_amd.engine import OpBuilder
    kc, mode = ctx
    # golden case: C_in=8 -> 12 is not 16-aligned: embed it in a 32->16 channel problem with zero padding
    w = torch.zeros(16, 8, k)
    w[:12] = torch.from_numpy(fill(f"u.conv.k{k}s{s}.conv.weight", (12, 8, k), SEED))
    b = torch.zeros(16)
    b[:12] = torch.from_numpy(fill(f"u.conv.k{k}s{s}.conv.bias", (12,), SEED))
    x = dev(fill_normal("u.conv.x.8", (2, 8, 37)))
    Lo = -(-37 // s)
    ob = OpBuilder(kc)
    src = to_cl(x, kc)
    out = new_out(kc, 2, Lo, 16)
    ob.conv(ob.ops, src0=src, w=pack_conv(w.cuda(), kc), bias=b.cuda(), out=out, taps=k, stride=s,
            pad_left=(k - 1) if causal else (k - 1) // 2, L_out=Lo)
    run(ob)
    y = from_cl(out)[:, :12].cpu().numpy()
    assert rel_err(y, golden("units")[f"conv.k{k}s{s}.c{int(causal)}"]) < TOL[mode]


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 12, 13, 14])
@pytest.mark.parametrize("splitk", [1, 3])
def test_conv_all_tile_configs_and_splitk(ctx, cfg, splitk):
    """every tile configuration (10+k = streaming cfg k with LDS staging instead of the direct
    register-ring path) and the split-K reduction give the same convolution."""
    from jen1_amd.engine import OpBuilder
    kc, mode = ctx
    torch.manual_seed(cfg * 7 + splitk)
    B, Ci, Co, Ln, k = 3, 96, 160, 45, 3
    x = torch.randn(B, Ci, Ln, device="cuda")
    w = torch.randn(Co, Ci, k, device="cuda") / (Ci * k) ** 0.5
    b = torch.randn(Co, device="cuda") * 0.1
    ref = F.conv1d(F.pad(x, (1, 1)), w, b)
    ob = OpBuilder(kc)
    out = new_out(kc, B, Ln, Co)
    ob.conv(ob.ops, src0=to_cl(x, kc), w=pack_conv(w, kc), bias=b, out=out, taps=k, pad_left=1,
            force={"cfg": cfg % 10, "splitk": splitk, "direct": cfg < 10})
    run(ob)
    if splitk > 1:
        assert int(ob.counters.abs().sum().item()) == 0, "split-K counters must be left at zero"
    assert rel_err(from_cl(out).cpu().numpy(), ref.cpu().numpy()) < TOL[mode]


@pytest.mark.parametrize("f", [2, 4])
@pytest.mark.parametrize("crop", [0, 1, 3])
def test_conv_transpose_subpixel_with_crop(ctx, f, crop):
    """ConvTranspose1d(k=2f, s=f) as a 2-tap sub-pixel GEMM, centre-crop folded into the store
    (reference blocks.py:88-95 + utils/module.py:186-204)."""
    from jen1_amd.engine import OpBuilder
    from jen1_amd.packing import convT_weight_to_gemm, pack_gemm_weight
    kc, mode = ctx
    torch.manual_seed(f)
    B, Ci, Co, Ln = 2, 64, 64, 11
    x = torch.randn(B, Ci, Ln, device="cuda")
    w = torch.randn(Ci, Co, 2 * f, device="cuda") / (Co * 2 * f) ** 0.5
    b = torch.randn(Co, device="cuda") * 0.1
    p = f // 2 + f % 2
    full = F.conv_transpose1d(x, w, b, stride=f, padding=p, output_padding=f % 2)
    assert full.shape[-1] == f * Ln
    L_need = f * Ln - crop
    st = crop // 2
    ref = full[:, :, st: st + L_need]
    ob = OpBuilder(kc)
    out = new_out(kc, B, L_need, Co, gn=True)
    ob.conv(ob.ops, src0=to_cl(x, kc), w=pack_gemm_weight(convT_weight_to_gemm(w, f), kc.tdtype), bias=b, out=out,
            taps=2, pad_left=1, L_out=Ln + 1, ps_f=f, ps_off=p + st, out_C=Co)
    run(ob)
    y = from_cl(out)
    assert rel_err(y.cpu().numpy(), ref.cpu().numpy()) < TOL[mode]
    # the GroupNorm sums of the cropped window were accumulated by the epilogue
    exp = to_cl(y, kc).gn
    assert rel_err(out.gn.cpu().numpy(), exp.cpu().numpy()) < 2e-3


def test_upsample_golden(ctx):
    from jen1_amd.engine import OpBuilder
    from jen1_amd.packing import convT_weight_to_gemm, pack_gemm_weight
    kc, mode = ctx
    g = golden("units")
    x = dev(fill_normal("u.up.x", (2, 8, 11)))
    for f in (2, 4):
        w = torch.zeros(8, 16, 2 * f)
        w[:, :12] = torch.from_numpy(fill(f"u.up.f{f}.upsample.weight", (8, 12, 2 * f), SEED))
        b = torch.zeros(16)
        b[:12] = torch.from_numpy(fill(f"u.up.f{f}.upsample.bias", (12,), SEED))
        ob = OpBuilder(kc)
        out = new_out(kc, 2, 11 * f, 16)
        ob.conv(ob.ops, src0=to_cl(x, kc), w=pack_gemm_weight(convT_weight_to_gemm(w.cuda(), f), kc.tdtype), bias=b.cuda(),
                out=out, taps=2, pad_left=1, L_out=12, ps_f=f, ps_off=f // 2 + f % 2, out_C=16)
        run(ob)
        assert rel_err(from_cl(out)[:, :12].cpu().numpy(), g[f"upsample.f{f}"]) < TOL[mode]


# ------------------------------------------------------------------ the lean tile kernel of the long levels (T* cfgs)
TILE_CFGS = [5, 6, 7, 8, 9, 10]


@pytest.mark.parametrize("cfg", TILE_CFGS)
@pytest.mark.parametrize("k,s", [(3, 1), (9, 4), (5, 2), (1, 1)])
def test_tile_kernel_plain_and_strided_conv(ctx, cfg, k, s):
    """jen1_conv_gemm with the T* tile configurations: plain / strided convolution (blocks.py:34-53), partial tiles in
    both directions, output statistics."""
    from jen1_amd.engine import OpBuilder
    kc, mode = ctx
    torch.manual_seed(cfg * 11 + k)
    B, Ci, Co, Ln = 3, 96, 192, 117
    x = torch.randn(B, Ci, Ln, device="cuda")
    w = torch.randn(Co, Ci, k, device="cuda") / (Ci * k) ** 0.5
    b = torch.randn(Co, device="cuda") * 0.1
    pl = (k - 1) // 2
    Lo = -(-Ln // s)
    ref = F.conv1d(F.pad(x, (pl, k - 1 - pl)), w, b, stride=s)
    assert ref.shape[-1] == Lo
    ob = OpBuilder(kc)
    out = new_out(kc, B, Lo, Co, gn=True)
    ob.conv(ob.ops, src0=to_cl(x, kc), w=pack_conv(w, kc), bias=b, out=out, taps=k, stride=s, pad_left=pl, L_out=Lo, force={"cfg": cfg})
    run(ob)
    y = from_cl(out)
    assert rel_err(y.cpu().numpy(), ref.cpu().numpy()) < TOL[mode]
    assert rel_err(out.gn.cpu().numpy(), to_cl(y, kc).gn.cpu().numpy()) < 2e-3


@pytest.mark.parametrize("cfg", TILE_CFGS)
@pytest.mark.parametrize("two_src,film,groups", [(False, False, 8), (True, True, 8), (False, True, 1), (True, False, 4)])
def test_tile_kernel_groupnorm_film_silu(ctx, cfg, two_src, film, groups):
    """ConvBlock1d on the tile kernel: GroupNorm -> FiLM -> SiLU -> conv k=3 over [x, skip * 2^-1/2] + residual + statistics
    (blocks.py:137-145, :219-231, :732-734); groups = 1 is the Patcher / Unpatcher case (blocks.py:251, :279)."""
    from jen1_amd import lib as L
    from jen1_amd.engine import OpBuilder
    kc, mode = ctx
    if groups == 1 and two_src:
        pytest.skip("single group over two sources is not a JEN-1 shape")
    torch.manual_seed(cfg + 100)
    B, C0, C1, Co, Ln = 2, 64, 64 if two_src else 0, 128, 150
    sc = 2 ** -0.5
    x0 = torch.randn(B, C0, Ln, device="cuda") * 1.5 + 0.3
    x1 = torch.randn(B, C1, Ln, device="cuda") * 0.7 - 0.2 if two_src else None
    Ct = C0 + C1
    gam, bet = torch.rand(Ct, device="cuda") + 0.5, torch.randn(Ct, device="cuda") * 0.1
    w = torch.randn(Co, Ct, 3, device="cuda") / (Ct * 3) ** 0.5
    bias = torch.randn(Co, device="cuda") * 0.1
    resid = torch.randn(B, Co, Ln, device="cuda")
    ftab = torch.randn(5, 2 * Ct + 7, device="cuda") * 0.3
    frow = torch.tensor([4, 0], dtype=torch.int32, device="cuda")
    xin = x0 if not two_src else torch.cat([x0, x1 * sc], 1)
    h = F.group_norm(xin, groups, gam, bet, 1e-5)
    if film:
        fs = ftab[frow.long()][:, 7: 7 + Ct, None]
        fh = ftab[frow.long()][:, 7 + Ct: 7 + 2 * Ct, None]
        h = h * (fs + 1) + fh
    ref = F.conv1d(F.pad(F.silu(h), (1, 1)), w, bias) + resid
    ob = OpBuilder(kc)
    out = new_out(kc, B, Ln, Co, gn=True)
    ob.conv(ob.ops, src0=to_cl(x0, kc), src1=to_cl(x1, kc) if two_src else None, src1_scale=sc if two_src else 1.0,
            w=pack_conv(w, kc), bias=bias, out=out, taps=3, pad_left=1, pro=L.PRO_GN_SILU,
            gn=(groups, Ct, gam, bet, 1e-5), film=(ftab, frow, 7, Ct) if film else None, residual=to_cl(resid, kc), force={"cfg": cfg})
    run(ob)
    y = from_cl(out)
    assert rel_err(y.cpu().numpy(), ref.cpu().numpy()) < TOL[mode]
    assert rel_err(out.gn.cpu().numpy(), to_cl(y, kc).gn.cpu().numpy()) < 2e-3


@pytest.mark.parametrize("cfg", [5, 6, 7, 10])
@pytest.mark.parametrize("f,crop", [(4, 0), (4, 3), (2, 1)])
def test_tile_kernel_conv_transpose_subpixel(ctx, cfg, f, crop):
    """Upsample1d on the tile kernel: ConvTranspose1d(k=2f, s=f) as a 2-tap sub-pixel GEMM with the centre crop folded
    into the store and the final residual (blocks.py:88-95, utils/module.py:186-204, model.py:261)."""
    from jen1_amd import lib as L
    from jen1_amd.engine import OpBuilder
    from jen1_amd.packing import convT_weight_to_gemm, pack_gemm_weight
    kc, mode = ctx
    BM = L.load().jen1_cfg_bm(cfg)
    Co = 128 if BM <= 128 else 256
    torch.manual_seed(f + cfg)
    B, Ci, Ln = 2, 64, 70
    x = torch.randn(B, Ci, Ln, device="cuda")
    w = torch.randn(Ci, Co, 2 * f, device="cuda") / (Co * 2 * f) ** 0.5
    b = torch.randn(Co, device="cuda") * 0.1
    p = f // 2 + f % 2
    full = F.conv_transpose1d(x, w, b, stride=f, padding=p, output_padding=f % 2)
    L_need = f * Ln - crop
    st = crop // 2
    resid = torch.randn(B, Co, L_need, device="cuda")
    ref = full[:, :, st: st + L_need] + resid
    ob = OpBuilder(kc)
    out = new_out(kc, B, L_need, Co, gn=True)
    ob.conv(ob.ops, src0=to_cl(x, kc), w=pack_gemm_weight(convT_weight_to_gemm(w, f), kc.tdtype), bias=b, out=out,
            taps=2, pad_left=1, L_out=Ln + 1, ps_f=f, ps_off=p + st, out_C=Co, residual=to_cl(resid, kc), force={"cfg": cfg})
    run(ob)
    y = from_cl(out)
    assert rel_err(y.cpu().numpy(), ref.cpu().numpy()) < TOL[mode]
    assert rel_err(out.gn.cpu().numpy(), to_cl(y, kc).gn.cpu().numpy()) < 2e-3


# ------------------------------------------------------------------ fused prologues / epilogues
@pytest.mark.parametrize("two_src", [False, True])
@pytest.mark.parametrize("film", [False, True])
def test_groupnorm_film_silu_prologue(ctx, two_src, film):
    """ConvBlock1d: GroupNorm -> x*(scale+1)+shift -> SiLU -> conv (blocks.py:137-145), optionally over
    the channel concat [x, skip * 2^-1/2] (blocks.py:732-734), + residual + output statistics."""
    from jen1_amd import lib as L
    from jen1_amd.engine import OpBuilder
    kc, mode = ctx
    torch.manual_seed(11)
    B, C0, C1, Co, Ln, G = 3, 64, 64 if two_src else 0, 128, 29, 8
    sc = 2 ** -0.5
    x0 = torch.randn(B, C0, Ln, device="cuda") * 1.5 + 0.3
    x1 = torch.randn(B, C1, Ln, device="cuda") * 0.7 - 0.2 if two_src else None
    Ct = C0 + C1
    gam, bet = torch.rand(Ct, device="cuda") + 0.5, torch.randn(Ct, device="cuda") * 0.1
    w = torch.randn(Co, Ct, 3, device="cuda") / (Ct * 3) ** 0.5
    bias = torch.randn(Co, device="cuda") * 0.1
    resid = torch.randn(B, Co, Ln, device="cuda")
    ftab = torch.randn(5, 2 * Ct + 7, device="cuda") * 0.3
    frow = torch.tensor([4, 0, 2], dtype=torch.int32, device="cuda")
    xin = x0 if not two_src else torch.cat([x0, x1 * sc], 1)
    h = F.group_norm(xin, G, gam, bet, 1e-5)
    if film:
        fs = ftab[frow.long()][:, 7: 7 + Ct, None]
        fh = ftab[frow.long()][:, 7 + Ct: 7 + 2 * Ct, None]
        h = h * (fs + 1) + fh
    ref = F.conv1d(F.pad(F.silu(h), (2, 0)), w, bias) + resid          # causal padding
    for force in (None, {"cfg": 0}, {"cfg": 2, "direct": False}, {"cfg": 3}):
        ob = OpBuilder(kc)
        out = new_out(kc, B, Ln, Co, gn=True, rs=True)
        ob.conv(ob.ops, src0=to_cl(x0, kc), src1=to_cl(x1, kc) if two_src else None, src1_scale=sc if two_src else 1.0,
                w=pack_conv(w, kc), bias=bias, out=out, taps=3, pad_left=2, pro=L.PRO_GN_SILU,
                gn=(G, Ct, gam, bet, 1e-5), film=(ftab, frow, 7, Ct) if film else None, residual=to_cl(resid, kc),
                force=force)
        run(ob)
        y = from_cl(out)
        assert rel_err(y.cpu().numpy(), ref.cpu().numpy()) < TOL[mode], force
        exp = to_cl(y, kc)
        assert rel_err(out.gn.cpu().numpy(), exp.gn.cpu().numpy()) < 2e-3, force
        assert rel_err(out.rs.cpu().numpy(), exp.rs.cpu().numpy()) < 2e-3, force


@pytest.mark.parametrize("splitk", [1, 2])
@pytest.mark.parametrize("Ln", [1, 2, 13])
def test_extra_k_segments_fused_shortcut(ctx, splitk, Ln):
    """ResnetBlock1d second half on a streaming level: conv k=3 over GroupNorm+SiLU(h) plus the 1x1 shortcut over
    the raw [x, skip] as two extra K segments of the same launch (blocks.py:219-231, :732-734); Ln = 1, 2 exercise
    the dead-segment skip (taps that only see zero padding)."""
    from jen1_amd import lib as L
    from jen1_amd.engine import OpBuilder
    from jen1_amd.packing import conv_weight_to_gemm, pack_gemm_weight
    kc, mode = ctx
    torch.manual_seed(5 + Ln)
    B, Ch, C0, C1, Co, G = 4, 64, 64, 32, 128, 8
    h = torch.randn(B, Ch, Ln, device="cuda") * 1.3 + 0.2
    x0 = torch.randn(B, C0, Ln, device="cuda")
    x1 = torch.randn(B, C1, Ln, device="cuda")
    gam, bet = torch.rand(Ch, device="cuda") + 0.5, torch.randn(Ch, device="cuda") * 0.1
    w2 = torch.randn(Co, Ch, 3, device="cuda") / (Ch * 3) ** 0.5
    ws = torch.randn(Co, C0 + C1, 1, device="cuda") / (C0 + C1) ** 0.5
    b2, bs = torch.randn(Co, device="cuda") * 0.1, torch.randn(Co, device="cuda") * 0.1
    ref = F.conv1d(F.pad(F.silu(F.group_norm(h, G, gam, bet, 1e-5)), (1, 1)), w2, b2) + F.conv1d(torch.cat([x0, x1], 1), ws, bs)
    wf = torch.cat([pack_gemm_weight(conv_weight_to_gemm(w2), kc.tdtype).flatten(0, 1),
                    pack_gemm_weight(conv_weight_to_gemm(ws), kc.tdtype).flatten(0, 1)], 0).contiguous()
    ob = OpBuilder(kc)
    out = new_out(kc, B, Ln, Co, gn=True, rs=True)
    ob.conv(ob.ops, src0=to_cl(h, kc), w=wf, bias=b2 + bs, out=out, taps=3, pad_left=1, pro=L.PRO_GN_SILU,
            gn=(G, Ch, gam, bet, 1e-5), extra_segs=[(to_cl(x0, kc), 0), (to_cl(x1, kc), 0)],
            force={"cfg": L.CFG_S16x16 if B * Ln <= 16 else L.CFG_S16x64, "splitk": splitk})
    run(ob)
    y = from_cl(out)
    assert rel_err(y.cpu().numpy(), ref.cpu().numpy()) < TOL[mode]
    # the kernel sums the float32 values it is about to round; with Ln = 1 a fine group holds 4 numbers, so in
    # bf16 mode the sums of the ROUNDED outputs differ by up to one bf16 ulp of an element
    exp = to_cl(y, kc)
    stol = 2e-3 if mode == "f32" else 1e-2
    assert rel_err(out.gn.cpu().numpy(), exp.gn.cpu().numpy()) < stol
    assert rel_err(out.rs.cpu().numpy(), exp.rs.cpu().numpy()) < stol


def test_layernorm_prologue_gelu_rowscale(ctx):
    """Linear(LayerNorm(x)) with GELU epilogue and a row mask (blocks.py:427-434, :440-446)."""
    from jen1_amd import lib as L
    from jen1_amd.engine import OpBuilder
    kc, mode = ctx
    torch.manual_seed(5)
    B, Ln, Ci, Co = 2, 19, 160, 64
    x = torch.randn(B, Ln, Ci, device="cuda") * 2 + 0.5
    gam, bet = torch.rand(Ci, device="cuda") + 0.5, torch.randn(Ci, device="cuda") * 0.1
    w = torch.randn(Co, Ci, device="cuda") / Ci ** 0.5
    bias = torch.randn(Co, device="cuda") * 0.1
    mask = (torch.rand(B * Ln, device="cuda") > 0.3).float()
    ref = F.gelu(F.linear(F.layer_norm(x, (Ci,), gam, bet), w, bias)) * mask.view(B, Ln, 1)
    from jen1_amd.packing import pack_gemm_weight
    from jen1_amd.packing import fold_layernorm as _fold
    wf_, _ = _fold(w, gam, bet)
    u_fold = wf_.to(kc.tdtype).float().sum(1).contiguous()
    for folded, force in ((False, None), (True, None), (False, {"cfg": 0}), (True, {"cfg": 2, "direct": False}),
                          (True, {"cfg": 2, "epi": True}), (True, {"cfg": 4, "epi": True})):
        ob = OpBuilder(kc)
        out = new_out(kc, B, Ln, Co)
        src = to_cl(x.permute(0, 2, 1).contiguous(), kc)
        if folded:
            from jen1_amd.packing import fold_layernorm
            wf, bf = fold_layernorm(w, gam, bet)
            ob.conv(ob.ops, src0=src, w=pack_gemm_weight(wf[None], kc.tdtype), bias=(bf + bias).contiguous(), out=out,
                    pro=L.PRO_LN, ln=(Ci, None, None, u_fold) if (force or {}).get("epi") else (Ci, None, None),
                    act=L.ACT_GELU, row_scale=mask, force=force)
        else:
            ob.conv(ob.ops, src0=src, w=pack_gemm_weight(w[None], kc.tdtype), bias=bias, out=out, pro=L.PRO_LN,
                    ln=(Ci, gam, bet), act=L.ACT_GELU, row_scale=mask, force=force)
        run(ob)
        assert rel_err(out.t[:, :, :Co].float().cpu().numpy(), ref.cpu().numpy()) < TOL[mode], (folded, force)


def test_silu_prologue_f32_output(ctx):
    """MappingToScaleShift: Linear(SiLU(mapping)) for all blocks at once, float32 result (blocks.py:148-165)."""
    from jen1_amd import lib as L
    from jen1_amd.engine import Act, OpBuilder
    from jen1_amd.packing import pack_gemm_weight
    kc, mode = ctx
    torch.manual_seed(9)
    n, Ci, Co = 5, 256, 1184
    m = torch.randn(n, Ci, device="cuda")
    w = torch.randn(Co, Ci, device="cuda") / Ci ** 0.5
    b = torch.randn(Co, device="cuda") * 0.1
    ref = F.linear(F.silu(m), w, b)
    ob = OpBuilder(kc)
    src = Act(m.to(kc.tdtype).view(1, n, Ci).contiguous(), 1, n, Ci, Ci)
    o = torch.zeros((1, n, Co), dtype=torch.float32, device="cuda")
    ob.conv(ob.ops, src0=src, w=pack_gemm_weight(w[None], kc.tdtype), bias=b, out=Act(o, 1, n, Co, Co), pro=L.PRO_SILU,
            y_f32=True)
    run(ob)
    assert rel_err(o[0].cpu().numpy(), ref.cpu().numpy()) < TOL[mode]


# ------------------------------------------------------------------ attention (a7)
@pytest.mark.parametrize("d,Nq,Nk,causal", [(8, 7, 7, False), (8, 7, 7, True), (32, 24, 24, True), (64, 12, 129, False),
                                            (128, 3, 129, False), (16, 45, 45, True)])
def test_attention_core(ctx, d, Nq, Nk, causal):
    from jen1_amd.engine import Act, OpBuilder
    kc, mode = ctx
    torch.manual_seed(d + Nq)
    B, H = 3, 4
    mid = H * d
    q = torch.randn(B, Nq, mid, device="cuda")
    kv = torch.randn(2 * B, Nk, 2 * mid, device="cuda")
    kv_row = torch.tensor([4, 0, 3], dtype=torch.int32, device="cuda")
    extra = torch.randn(5, 3 * mid, device="cuda")
    extra_row = torch.tensor([-1, 2, 4], dtype=torch.int32, device="cuda")
    qt, kvt, ext = q.to(kc.tdtype), kv.to(kc.tdtype), extra.to(kc.tdtype)
    # reference in float32 on the rounded inputs
    kk = kvt.float()[kv_row.long()].clone()
    for b in range(B):
        if extra_row[b] >= 0:
            kk[b, Nk - 1, :mid] = ext.float()[extra_row[b], mid // 2: mid // 2 + mid]
            kk[b, Nk - 1, mid:] = ext.float()[extra_row[b], 2 * mid: 3 * mid]
    qh = qt.float().view(B, Nq, H, d).transpose(1, 2)
    kh = kk[:, :, :mid].reshape(B, Nk, H, d).transpose(1, 2)
    vh = kk[:, :, mid:].reshape(B, Nk, H, d).transpose(1, 2)
    sim = qh @ kh.transpose(-1, -2) * d ** -0.5
    if causal:
        keep = ~torch.ones(Nq, Nk, dtype=torch.bool, device="cuda").triu(Nk - Nq + 1)
        sim = sim.masked_fill(~keep, -torch.finfo(torch.float32).max)
    ref = (sim.softmax(-1) @ vh).transpose(1, 2).reshape(B, Nq, mid)
    ob = OpBuilder(kc)
    out = Act(torch.zeros((B, Nq, mid), dtype=kc.tdtype, device="cuda"), B, Nq, mid, mid)
    ob.attention(ob.ops, q=Act(qt.contiguous(), B, Nq, mid, mid), q_off=0, kv_t=kvt.contiguous(), ldkv=2 * mid, k_off=0,
                 v_off=mid, out=out, H=H, d=d, Nk=Nk, causal=causal, kv_row=kv_row, kv_extra=ext.contiguous(),
                 extra_row=extra_row, ld_extra=3 * mid, kx_off=mid // 2, vx_off=2 * mid)
    run(ob)
    assert rel_err(out.t.float().cpu().numpy(), ref.cpu().numpy()) < (1e-5 if mode == "f32" else 1e-2)


@pytest.mark.parametrize("d,Nq,Nk,causal", [(32, 141, 141, True), (64, 71, 129, False), (128, 5, 129, False), (32, 24, 24, False)])
def test_attention_core_fp8_operands(d, Nq, Nk, causal):
    """JEN1_FP8 mode of jen1_attention_fin (BASELINE configs[4] "fp8 MFMA attention path", blocks.py:355-380): Q K^T and P V on
    e4m3 operands (v_mfma_f32_16x16x32_fp8_fp8), float32 softmax, probabilities stored as 256 p.  q / k / v / out stay bf16 in
    memory.  Against float32 attention on the same bf16 inputs; stated tolerance 1e-1 of the largest output (3 mantissa bits per
    operand; N(0, 1) queries and keys give logits of unit variance, i.e. peaky rows where one probability carries the row:
    measured 2e-2 .. 7e-2), a layout or scaling mistake would be O(1)."""
    from jen1_amd import lib as L
    from jen1_amd.engine import Act, KernelCtx, OpBuilder
    kc = KernelCtx("bf16", "cuda")
    kc.deep_dt = L.FP8
    torch.manual_seed(d + Nq)
    B, H = 2, 8
    mid = H * d
    q = torch.randn(B, Nq, mid, device="cuda").to(torch.bfloat16)
    kv = torch.randn(B, Nk, 2 * mid, device="cuda").to(torch.bfloat16)
    qh = q.float().view(B, Nq, H, d).transpose(1, 2)
    kh = kv.float()[:, :, :mid].reshape(B, Nk, H, d).transpose(1, 2)
    vh = kv.float()[:, :, mid:].reshape(B, Nk, H, d).transpose(1, 2)
    sim = qh @ kh.transpose(-1, -2) * d ** -0.5
    if causal:
        keep = ~torch.ones(Nq, Nk, dtype=torch.bool, device="cuda").triu(Nk - Nq + 1)
        sim = sim.masked_fill(~keep, -torch.finfo(torch.float32).max)
    ref = (sim.softmax(-1) @ vh).transpose(1, 2).reshape(B, Nq, mid)
    outs = {}
    for name, dt in (("fp8", L.FP8), ("bf16", L.BF16)):
        kc.deep_dt = dt
        ob = OpBuilder(kc)
        out = Act(torch.zeros((B, Nq, mid), dtype=torch.bfloat16, device="cuda"), B, Nq, mid, mid)
        ob.attention(ob.ops, q=Act(q.contiguous(), B, Nq, mid, mid), q_off=0, kv_t=kv.contiguous(), ldkv=2 * mid, k_off=0, v_off=mid,
                     out=out, H=H, d=d, Nk=Nk, causal=causal)
        run(ob)
        outs[name] = rel_err(out.t.float().cpu().numpy(), ref.cpu().numpy())
    print(f"attention d={d} Nq={Nq} Nk={Nk}: fp8 {outs['fp8']:.3e}, bf16 {outs['bf16']:.3e}")
    assert outs["fp8"] < 1e-1 and outs["bf16"] < 1e-2


@pytest.mark.parametrize("N,causal", [(1, False), (6, True), (24, False)])
def test_attention_deferred_layernorm_finish(ctx, N, causal):
    """jen1_attention_fin: Q / K / V arrive as raw = W' x of a LayerNorm-folded projection; the kernel applies
    rstd (raw - mean u) + b from the row statistics of x (blocks.py:427-429), then the usual attention."""
    from jen1_amd.engine import Act, OpBuilder
    kc, mode = ctx
    torch.manual_seed(N)
    B, H, d, C = 2, 4, 32, 96
    mid = H * d
    x = torch.randn(B, N, C, device="cuda") * 1.3 + 0.4
    gam, bet = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.2
    W = torch.randn(3 * mid, C, device="cuda") / C ** 0.5
    Wf = (W * gam[None, :]).to(kc.tdtype).float()             # folded weights as the GEMM sees them
    bf = W @ bet
    raw = (x @ Wf.T)
    t = torch.zeros(B, N, C + 3 * mid, device="cuda")
    t[:, :, :C], t[:, :, C:] = x, raw
    tt = t.to(kc.tdtype).contiguous()
    rawr = tt.float()[:, :, C:]
    rs = torch.stack([x.sum(-1), (x * x).sum(-1)], -1).reshape(-1).contiguous()
    u = torch.cat([torch.zeros(C, device="cuda"), Wf.sum(1)]).contiguous()
    bfull = torch.cat([torch.zeros(C, device="cuda"), bf]).contiguous()
    mean, var = x.mean(-1, keepdim=True), x.var(-1, unbiased=False, keepdim=True)
    qkv = (rawr - mean * Wf.sum(1)) / torch.sqrt(var + 1e-5) + bf
    qh, kh, vh = (z.reshape(B, N, H, d).transpose(1, 2) for z in qkv.split(mid, dim=-1))
    sim = qh @ kh.transpose(-1, -2) * d ** -0.5
    if causal:
        sim = sim.masked_fill(torch.ones(N, N, dtype=torch.bool, device="cuda").triu(1), -torch.finfo(torch.float32).max)
    ref = (sim.softmax(-1) @ vh).transpose(1, 2).reshape(B, N, mid)
    ob = OpBuilder(kc)
    out = Act(torch.zeros((B, N, mid), dtype=kc.tdtype, device="cuda"), B, N, mid, mid)
    ob.attention(ob.ops, q=Act(tt, B, N, C + 3 * mid, C + 3 * mid), q_off=C, kv_t=tt, ldkv=C + 3 * mid, k_off=C + mid, v_off=C + 2 * mid,
                 out=out, H=H, d=d, Nk=N, causal=causal, fin=(rs, u, bfull, C, 1e-5, 1, 1))
    run(ob)
    assert rel_err(out.t.float().cpu().numpy(), ref.cpu().numpy()) < (2e-5 if mode == "f32" else 2e-2)


# ------------------------------------------------------------------ boundary kernels
def test_pack_unpack_roundtrip_and_stats(ctx):
    from jen1_amd import lib as L
    kc, mode = ctx
    lib = kc.lib
    B, Cx, Cc, T = 2, 128, 129, 77
    x = dev(fill_normal("pk.x", (B, Cx, T)))
    c = dev(fill_normal("pk.c", (B, Cc, T)))
    ld = 288
    y = torch.zeros((2 * B, T, ld), dtype=kc.tdtype, device="cuda")
    st = torch.zeros(2 * B * 64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    L.check(lib.jen1_pack_input(x.data_ptr(), c.data_ptr(), y.data_ptr(), st.data_ptr(), B, Cx, Cc, T, ld, 2, kc.dt, s))
    torch.cuda.synchronize()
    ref = torch.cat([x, c], 1).permute(0, 2, 1)
    for rep in range(2):
        assert rel_err(y[rep * B:(rep + 1) * B, :, : Cx + Cc].float().cpu().numpy(), ref.cpu().numpy()) < (1e-7 if mode == "f32" else 8e-3)
        assert float(y[rep * B:(rep + 1) * B, :, Cx + Cc:].abs().max()) == 0.0
    full = torch.zeros(B, T, ld, device="cuda")
    full[:, :, : Cx + Cc] = ref
    g = full.reshape(B, T, 32, ld // 32)
    exp = torch.stack([g.sum((1, 3)), (g * g).sum((1, 3))], -1).reshape(B * 64)
    assert rel_err(st[: B * 64].cpu().numpy(), exp.cpu().numpy()) < 1e-4
    assert rel_err(st[B * 64:].cpu().numpy(), exp.cpu().numpy()) < 1e-4
    out = torch.zeros(B, Cx, T, device="cuda")
    L.check(lib.jen1_unpack_output(y.data_ptr(), out.data_ptr(), B, Cx, T, ld, kc.dt, s))
    torch.cuda.synchronize()
    assert rel_err(out.cpu().numpy(), x.cpu().numpy()) < (1e-7 if mode == "f32" else 8e-3)


def test_row_stats_and_time_features(ctx):
    from jen1_amd import lib as L
    kc, mode = ctx
    lib = kc.lib
    s = torch.cuda.current_stream().cuda_stream
    x = torch.randn(37, 1024, device="cuda").to(kc.tdtype)
    st = torch.zeros(37 * 2, device="cuda")
    L.check(lib.jen1_row_stats(x.data_ptr(), st.data_ptr(), 37, 1024, 1024, kc.dt, s))
    xf = x.float()
    exp = torch.stack([xf.sum(-1), (xf * xf).sum(-1)], -1).reshape(-1)
    torch.cuda.synchronize()
    assert rel_err(st.cpu().numpy(), exp.cpu().numpy()) < 1e-5
    # LearnedPositionalEmbedding + Linear (+GELU) vs the golden unit (no GELU in the fixture)
    t = torch.tensor([0, 1, 9, 499, 989, 999], dtype=torch.int64, device="cuda")
    fw = dev(fill("u.time.0.weights", (32,), SEED))
    w = dev(fill("u.time.1.weight", (40, 65), SEED))
    b = dev(fill("u.time.1.bias", (40,), SEED))
    out = torch.zeros(6, 40, device="cuda")
    L.check(lib.jen1_time_features(t.data_ptr(), fw.data_ptr(), w.data_ptr(), b.data_ptr(), out.data_ptr(), 6, 32, 40, s))
    torch.cuda.synchronize()
    ref = F.gelu(torch.from_numpy(golden("units")["time.features"]))
    assert rel_err(out.cpu().numpy(), ref.numpy()) < 1e-4
    # tiny float32 linear
    xx = torch.randn(5, 129, device="cuda")
    ww = torch.randn(77, 129, device="cuda")
    bb = torch.randn(77, device="cuda")
    yy = torch.zeros(5, 77, device="cuda")
    L.check(lib.jen1_linear_f32(xx.data_ptr(), ww.data_ptr(), bb.data_ptr(), yy.data_ptr(), 5, 129, 77, L.ACT_GELU, s))
    torch.cuda.synchronize()
    assert rel_err(yy.cpu().numpy(), F.gelu(F.linear(xx, ww, bb)).cpu().numpy()) < 1e-5


@pytest.mark.parametrize("objective", ["noise", "x0", "v"])
@pytest.mark.parametrize("last", [False, True])
def test_cfg_ddim_step(ctx, objective, last):
    """CFG combine + std rescale (model.py:362-369) + model_predictions + DDIM update (gdm.py:128-140, 212-222)."""
    from jen1_amd import lib as L
    kc, mode = ctx
    lib = kc.lib
    torch.manual_seed(3)
    B, Cc, T = 2, 128, 75
    net = torch.randn(2 * B, T, Cc, device="cuda").to(kc.tdtype)
    x = torch.randn(B, Cc, T, device="cuda")
    noise = torch.randn(B, Cc, T, device="cuda")
    coef = torch.tensor([1.7, 1.3, 0.8, 0.5, 0.3, 1.0 if last else 0.0, 0.6, 0.8], device="cuda")
    xo, eo, x0o = (torch.zeros(B, Cc, T, device="cuda") for _ in range(3))
    s = torch.cuda.current_stream().cuda_stream
    obj = {"noise": 0, "x0": 1, "v": 2}[objective]
    # coefficients / noise as per-step tables indexed by a device-side counter (row 2 holds the real data)
    coef_tab = torch.zeros(4, 8, device="cuda")
    coef_tab[2] = coef
    noise_tab = torch.randn(4, B, Cc, T, device="cuda")
    noise_tab[2] = noise
    step = torch.tensor([2], dtype=torch.int32, device="cuda")
    L.check(lib.jen1_cfg_ddim_step(net.data_ptr(), x.data_ptr(), noise_tab.data_ptr(), coef_tab.data_ptr(), xo.data_ptr(),
                                   eo.data_ptr(), x0o.data_ptr(), step.data_ptr(), B, Cc, T, Cc, 2, 0.8, 1, 0.7, obj, 1, kc.dt, s))
    L.check(lib.jen1_step_advance(step.data_ptr(), s))
    torch.cuda.synchronize()
    assert int(step.item()) == 3
    xo2 = torch.zeros_like(xo)
    L.check(lib.jen1_cfg_ddim_step(net.data_ptr(), x.data_ptr(), noise.data_ptr(), coef.data_ptr(), xo2.data_ptr(), None,
                                   None, None, B, Cc, T, Cc, 2, 0.8, 1, 0.7, obj, 1, kc.dt, s))
    torch.cuda.synchronize()
    assert torch.equal(xo, xo2), "table-indexed and direct coefficient/noise paths must agree bit for bit"
    nf = net.float().permute(0, 2, 1)
    out, outm = nf[:B], nf[B:]
    oc = outm + (out - outm) * 0.8
    o = 0.7 * (oc * (out.std(1, keepdim=True) / oc.std(1, keepdim=True))) + 0.3 * oc
    sr, srm1, san, c, sg, _, sat, s1m = coef.tolist()
    if objective == "noise":
        eps = o
        x0 = (sr * x - srm1 * eps).clamp(-1, 1)
    elif objective == "x0":
        x0 = o.clamp(-1, 1)
        eps = (sr * x - x0) / srm1
    else:
        x0 = (sat * x - s1m * o).clamp(-1, 1)
        eps = (sr * x - x0) / srm1
    xn = x0 if last else x0 * san + c * eps + sg * noise
    assert rel_err(x0o.cpu().numpy(), x0.cpu().numpy()) < 2e-5
    assert rel_err(eo.cpu().numpy(), eps.cpu().numpy()) < 2e-5
    assert rel_err(xo.cpu().numpy(), xn.cpu().numpy()) < 2e-5
    g = torch.zeros(B, Cc, T, device="cuda")
    L.check(lib.jen1_cfg_combine(net.data_ptr(), g.data_ptr(), B, Cc, T, Cc, 0.8, 1, 0.7, kc.dt, s))
    torch.cuda.synchronize()
    assert rel_err(g.cpu().numpy(), o.cpu().numpy()) < 2e-5


@pytest.mark.parametrize("cfg", [5, 6, 7, 10])
@pytest.mark.parametrize("two_extra", [False, True])
def test_tile_kernel_raw_extra_segments_fused_shortcut(ctx, cfg, two_extra):
    """ResnetBlock1d second half on a tiled long level: conv k=3 over GroupNorm + FiLM + SiLU(h) plus the 1x1 shortcut over the raw
    block input [x, skip] as extra K segments staged behind the normalised channels of the same tile (blocks.py:219-231, :732-734)"""
    from jen1_amd import lib as L
    from jen1_amd.engine import OpBuilder
    from jen1_amd.packing import conv_weight_to_gemm, pack_gemm_weight
    kc, mode = ctx
    torch.manual_seed(31 + cfg)
    B, Ch, C0, C1, Co, G, Ln = 2, 128, 128, 64 if two_extra else 0, 128, 8, 150
    h = torch.randn(B, Ch, Ln, device="cuda") * 1.3 + 0.2
    x0 = torch.randn(B, C0, Ln, device="cuda")
    x1 = torch.randn(B, C1, Ln, device="cuda") if two_extra else None
    gam, bet = torch.rand(Ch, device="cuda") + 0.5, torch.randn(Ch, device="cuda") * 0.1
    ftab = torch.randn(3, 2 * Ch + 5, device="cuda") * 0.3
    frow = torch.tensor([2, 0], dtype=torch.int32, device="cuda")
    w2 = torch.randn(Co, Ch, 3, device="cuda") / (Ch * 3) ** 0.5
    ws = torch.randn(Co, C0 + C1, 1, device="cuda") / (C0 + C1) ** 0.5
    b2, bs = torch.randn(Co, device="cuda") * 0.1, torch.randn(Co, device="cuda") * 0.1
    hn = F.group_norm(h, G, gam, bet, 1e-5)
    hn = hn * (ftab[frow.long()][:, 5: 5 + Ch, None] + 1) + ftab[frow.long()][:, 5 + Ch: 5 + 2 * Ch, None]
    xin = x0 if not two_extra else torch.cat([x0, x1], 1)
    ref = F.conv1d(F.pad(F.silu(hn), (1, 1)), w2, b2) + F.conv1d(xin, ws, bs)
    wf = torch.cat([pack_gemm_weight(conv_weight_to_gemm(w2), kc.tdtype).flatten(0, 1),
                    pack_gemm_weight(conv_weight_to_gemm(ws), kc.tdtype).flatten(0, 1)], 0).contiguous()
    ob = OpBuilder(kc)
    out = new_out(kc, B, Ln, Co, gn=True)
    extras = [(to_cl(x0, kc), 0)] + ([(to_cl(x1, kc), 0)] if two_extra else [])
    r = ob.conv(ob.ops, src0=to_cl(h, kc), w=wf, bias=b2 + bs, out=out, taps=3, pad_left=1, pro=L.PRO_GN_SILU,
                gn=(G, Ch, gam, bet, 1e-5), film=(ftab, frow, 5, Ch), extra_segs=extras, force={"cfg": cfg})
    assert r is out
    run(ob)
    y = from_cl(out)
    assert rel_err(y.cpu().numpy(), ref.cpu().numpy()) < TOL[mode]
    assert rel_err(out.gn.cpu().numpy(), to_cl(y, kc).gn.cpu().numpy()) < 2e-3


@pytest.mark.parametrize("B,Ln,ld", [(2, 77, 288), (3, 1500, 128), (1, 5, 1024), (4, 94, 32)])
def test_gn_stats_fixed_order(ctx, B, Ln, ld):
    """jen1_gn_stats: fine-group (sum, sumsq) of a channel-last tensor, written (not accumulated) and bit-reproducible"""
    from jen1_amd import lib as L
    kc, mode = ctx
    torch.manual_seed(B * 1000 + Ln)
    x = (torch.randn(B, Ln, ld, device="cuda") * 1.7 + 0.4).to(kc.tdtype)
    st = torch.full((B * 64,), 123.0, device="cuda")                       # stale contents must not survive
    s = torch.cuda.current_stream().cuda_stream
    L.check(kc.lib.jen1_gn_stats(x.data_ptr(), st.data_ptr(), B, Ln, ld, kc.dt, s))
    st2 = torch.zeros(B * 64, device="cuda")
    L.check(kc.lib.jen1_gn_stats(x.data_ptr(), st2.data_ptr(), B, Ln, ld, kc.dt, s))
    torch.cuda.synchronize()
    assert torch.equal(st, st2)
    g = x.double().reshape(B, Ln, 32, ld // 32)
    exp = torch.stack([g.sum((1, 3)), (g * g).sum((1, 3))], -1).reshape(B * 64)
    assert rel_err(st.cpu().numpy(), exp.float().cpu().numpy()) < 2e-6
    with pytest.raises(L.Jen1HipError):
        L.check(kc.lib.jen1_gn_stats(x.data_ptr(), st.data_ptr(), B, Ln, 48, kc.dt, s))


def test_cfg_step_ddpm_row(ctx):
    """row kind 2 of jen1_cfg_ddim_step: ancestral sampling x_next = coef1 x0 + coef2 x_t + sd noise (gdm.py:144-163)"""
    from jen1_amd import lib as L
    kc, mode = ctx
    torch.manual_seed(4)
    B, Cc, T = 2, 128, 50
    net = torch.randn(B, T, Cc, device="cuda").to(kc.tdtype)
    x = torch.randn(B, Cc, T, device="cuda")
    noise = torch.rand(B, Cc, T, device="cuda")
    coef = torch.tensor([1.4, 0.9, 0.35, 0.6, 0.2, 2.0, 0.0, 0.0], device="cuda")
    xo, x0o = torch.zeros(B, Cc, T, device="cuda"), torch.zeros(B, Cc, T, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    L.check(kc.lib.jen1_cfg_ddim_step(net.data_ptr(), x.data_ptr(), noise.data_ptr(), coef.data_ptr(), xo.data_ptr(), None, x0o.data_ptr(),
                                      None, B, Cc, T, Cc, 1, 1.0, 0, 0.7, 0, 1, kc.dt, s))
    torch.cuda.synchronize()
    eps = net.float().permute(0, 2, 1)
    x0 = (1.4 * x - 0.9 * eps).clamp(-1, 1)
    assert rel_err(x0o.cpu().numpy(), x0.cpu().numpy()) < 2e-5
    assert rel_err(xo.cpu().numpy(), (0.35 * x0 + 0.6 * x + 0.2 * noise).cpu().numpy()) < 2e-5
