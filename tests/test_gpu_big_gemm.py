"""-m gpu: the large-M matrix-core GEMM (csrc/big_gemm.hip, include/jen1_hip.h jen1_big_gemm) through the C ABI against plain
PyTorch float32 of the same product -- the cross-attention ``to_kv`` projection over the text context (reference
jen1/model/blocks.py:402-407, :427-434): the grouped form of the sampling plan (13 layers, row map into [B][129] caches, padding
mask, folded LayerNorm bias) and the plain form of the training pass (2B * 129 rows), ragged M, both compute dtypes."""
import ctypes as C

import numpy as np
import pytest
import torch

from helpers import rel_err  # noqa: F401
from jen1_amd import lib as L

pytestmark = pytest.mark.gpu


def _run(a, b, groups, *, row_scale=None, rows_in=0, rows_out=0, c_f32=False, accumulate=False, alpha=1.0, dtype="bf16", group_align=0):
    lib = L.load()
    tab = L.bgemm_group_table([(c.data_ptr(), None if bias is None else bias.data_ptr(), n0, N, c.stride(-2)) for c, bias, n0, N in groups], a.device)
    g = L.BGemmArgs()
    g.a, g.b, g.groups = a.data_ptr(), b.data_ptr(), tab.data_ptr()
    g.row_scale = None if row_scale is None else row_scale.data_ptr()
    g.M, g.Ntot, g.K, g.lda, g.ldb, g.n_groups = a.shape[0], b.shape[0], a.shape[1], a.stride(0), b.stride(0), len(groups)
    g.rows_in, g.rows_out, g.c_f32, g.accumulate = rows_in, rows_out, int(c_f32), int(accumulate)
    g.dtype = L.F32 if dtype == "f32" else L.BF16
    g.alpha = alpha
    g.group_align = group_align
    L.check(lib.jen1_big_gemm(C.byref(g), torch.cuda.current_stream().cuda_stream), "jen1_big_gemm")
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("M,N,K", [(2064, 2048, 1024), (2064, 1024, 2048), (1024, 512, 1024), (129, 256, 64), (130, 384, 192), (1, 128, 64)])
def test_plain_nt_matches_torch(dtype, M, N, K):
    """C = A B^T (one group, no epilogue extras), asymmetric random operands, ragged M (rows past M read as zeros and are never stored)"""
    if dtype == "f32" and K % 32 or dtype == "bf16" and K % 64:
        pytest.skip("K granularity")
    td = torch.float32 if dtype == "f32" else torch.bfloat16
    gen = torch.Generator(device="cuda").manual_seed(M * 7 + N)
    a = (torch.randn((M, K), device="cuda", generator=gen) * 0.5).to(td)
    b = (torch.randn((N, K), device="cuda", generator=gen) * 0.5).to(td)
    guard = torch.full((M + 3, N), 7.0, device="cuda", dtype=td)                  # rows past M must stay untouched
    c = guard[:M]
    _run(a, b, [(c, None, 0, N)], dtype=dtype)
    ref = a.float() @ b.float().t()
    tol = 1e-5 if dtype == "f32" else 1e-2
    err = float((c.float() - ref).abs().max() / ref.abs().max())
    assert err < tol, err
    assert bool((guard[M:] == 7.0).all())


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_grouped_projection_with_row_map_mask_and_bias(dtype):
    """the sampling plan's form: 3 column groups of different width (layers) over the same standardised rows; bias per group, the padding
    mask as a row scale, rows of batch element b land at rows b * 129 + n of a [B][129] cache whose 129th row is not touched"""
    td = torch.float32 if dtype == "f32" else torch.bfloat16
    B, NL, K = 3, 128, 1024
    widths = [512, 1024, 256]
    gen = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn((B * NL, K), device="cuda", generator=gen)
    xs = torch.empty((B * NL, K), device="cuda", dtype=td)
    lib = L.load()
    L.check(lib.jen1_standardize_rows(x.data_ptr(), xs.data_ptr(), B * NL, K, K, K, 1e-5, L.F32 if dtype == "f32" else L.BF16,
                                      torch.cuda.current_stream().cuda_stream), "jen1_standardize_rows")
    xr = x.to(td).float()
    want_xs = (xr - xr.mean(dim=1, keepdim=True)) / torch.sqrt(xr.var(dim=1, unbiased=False, keepdim=True) + 1e-5)
    assert float((xs.float() - want_xs).abs().max()) < (1e-5 if dtype == "f32" else 2e-2)
    w = (torch.randn((sum(widths), K), device="cuda", generator=gen) * 0.05).to(td)
    mask_b = (torch.rand((B, NL + 1), device="cuda", generator=gen) > 0.3).float()       # indexed by the OUTPUT row: [B][129]
    mask = mask_b[:, :NL].reshape(-1)
    outs, groups, n0 = [], [], 0
    for wd in widths:
        cache = torch.full((B, NL + 1, wd), -3.0, device="cuda", dtype=td)
        bias = torch.randn((wd,), device="cuda", generator=gen)
        outs.append((cache, bias, n0, wd))
        groups.append((cache.view(B * (NL + 1), wd), bias, n0, wd))
        n0 += wd
    _run(xs, w, groups, row_scale=mask_b, rows_in=NL, rows_out=NL + 1, dtype=dtype)
    for cache, bias, n0, wd in outs:
        ref = (xs.float() @ w[n0:n0 + wd].float().t() + bias[None]) * mask[:, None]
        got = cache[:, :NL].reshape(B * NL, wd).float()
        err = float((got - ref).abs().max() / ref.abs().max())
        assert err < (1e-5 if dtype == "f32" else 1e-2), err
        assert bool((cache[:, NL] == -3.0).all())                                 # the time-token row belongs to another kernel


def test_accumulate_alpha_and_float32_output():
    M, N, K = 300, 256, 128
    gen = torch.Generator(device="cuda").manual_seed(9)
    a = torch.randn((M, K), device="cuda", generator=gen).to(torch.bfloat16)
    b = torch.randn((N, K), device="cuda", generator=gen).to(torch.bfloat16)
    c = torch.randn((M, N), device="cuda", generator=gen)
    c0 = c.clone()
    _run(a, b, [(c, None, 0, N)], c_f32=True, accumulate=True, alpha=0.25)
    ref = c0 + 0.25 * (a.float() @ b.float().t())
    assert float((c - ref).abs().max() / ref.abs().max()) < 1e-5


def test_kv_fixed_fill_matches_torch_expression():
    """kv[B:] = fixed[None] * mask[:, :, None] for every layer in one launch (model.py:337)"""
    lib = L.load()
    B, R = 3, 129
    gen = torch.Generator(device="cuda").manual_seed(3)
    mask = (torch.rand((B, R), device="cuda", generator=gen) > 0.4).float()
    ents, keep = [], []
    import numpy as np
    rows = []
    for C2 in (512, 2048, 1024):
        fixed = torch.randn((R, C2), device="cuda", generator=gen).to(torch.bfloat16)
        out = torch.zeros((B, R, C2), device="cuda", dtype=torch.bfloat16)
        keep.append((fixed, out))
        rows.append((fixed.data_ptr(), out.data_ptr(), C2))
    tab = np.zeros((len(rows), 4), dtype=np.int64)
    for i, (f, o, c2) in enumerate(rows):
        tab[i, 0], tab[i, 1], tab[i, 2] = f, o, c2
    tab_d = torch.from_numpy(tab).cuda()
    L.check(lib.jen1_kv_fixed_fill(tab_d.data_ptr(), len(rows), mask.data_ptr(), B, R, L.BF16, torch.cuda.current_stream().cuda_stream), "jen1_kv_fixed_fill")
    torch.cuda.synchronize()
    for fixed, out in keep:
        assert torch.equal(out, fixed[None] * mask[:, :, None].to(torch.bfloat16))


@pytest.mark.parametrize("M,N,K", [(2064, 2048, 1024), (2064, 512, 1024), (130, 128, 256), (64, 256, 128), (1, 128, 128), (200, 136, 264)])
def test_weight_gradient_tn_matches_torch(M, N, K):
    """C[n][k] += sum_m A[m][n] B[m][k] (both operands as they lie in memory, reduction over the rows, float32 atomics into C): ragged
    M (rows past M read as zeros), partial last column tiles inside a wider pitch, accumulation into existing values"""
    lib = L.load()
    gen = torch.Generator(device="cuda").manual_seed(M + N + K)
    lda, ldb = -(-N // 128) * 128, -(-K // 128) * 128
    a_full = (torch.randn((M, lda), device="cuda", generator=gen) * 0.5).to(torch.bfloat16)
    b_full = (torch.randn((M, ldb), device="cuda", generator=gen) * 0.5).to(torch.bfloat16)
    c = torch.randn((N, K), device="cuda", generator=gen)
    c0 = c.clone()
    L.check(lib.jen1_big_gemm_tn(a_full.data_ptr(), b_full.data_ptr(), c.data_ptr(), M, N, K, lda, ldb, K, 0.5,
                                 torch.cuda.current_stream().cuda_stream), "jen1_big_gemm_tn")
    torch.cuda.synchronize()
    ref = c0 + 0.5 * (a_full[:, :N].float().t() @ b_full[:, :K].float())
    err = float((c - ref).abs().max() / ref.abs().max())
    assert err < 2e-5, err
    # the store form: C = the product (unsplit reduction, plain stores); whatever C held before is overwritten
    c2 = torch.full((N, K), float("nan"), device="cuda")
    L.check(lib.jen1_big_gemm_tn_store(a_full.data_ptr(), b_full.data_ptr(), c2.data_ptr(), M, N, K, lda, ldb, K, 0.5,
                                       torch.cuda.current_stream().cuda_stream), "jen1_big_gemm_tn_store")
    torch.cuda.synchronize()
    assert float((c2 - (ref - c0)).abs().max() / ref.abs().max()) < 2e-5


@pytest.mark.parametrize("B,T_in,ci,co,taps,stride,pad,with_bias", [
    (3, 50, 128, 64, 3, 1, 1, True),        # centred k = 3 (ResnetBlock1d conv, blocks.py:137-145)
    (3, 50, 128, 128, 3, 1, 2, False),      # causal k = 3 (left pad 2, blocks.py:45-50)
    (2, 257, 256, 96, 5, 2, 2, True),       # strided k = 5 (Downsample1d, blocks.py:56-66), ragged rows
    (16, 1500, 128, 128, 3, 1, 1, True),    # the level-0 shape of the pass: 24 000 reduction rows
    (4, 129, 200, 72, 1, 1, 0, True),       # Linear over many rows, widths that are no multiple of 128 (inside a wider pitch)
])
def test_conv_weight_gradient_tn_matches_torch(B, T_in, ci, co, taps, stride, pad, with_bias):
    """jen1_big_gemm_tn_conv: gw[co][ci][tap] += sum dy x (tap shift, stride, zero padding at the sequence ends, per batch element) and
    the bias gradient, against torch's conv1d autograd on the same bf16-rounded operands; accumulation into a non-zero gradient"""
    import torch.nn.functional as F
    lib = L.load()
    gen = torch.Generator(device="cuda").manual_seed(B * 131 + T_in + ci + taps)
    T_out = (T_in + (taps - 1 if taps > 1 else 0) - taps) // stride + 1 if taps > 1 else T_in
    if taps > 1:
        T_out = (T_in + taps - 1 - taps) // stride + 1          # total padding taps - 1 (left `pad`, right the rest)
    ldx, ldy = -(-ci // 128) * 128, -(-co // 128) * 128
    x = torch.zeros((B, T_in, ldx), device="cuda", dtype=torch.bfloat16)
    dy = torch.zeros((B, T_out, ldy), device="cuda", dtype=torch.bfloat16)
    x[..., :ci] = (torch.randn((B, T_in, ci), device="cuda", generator=gen) * 0.5).to(torch.bfloat16)
    dy[..., :co] = (torch.randn((B, T_out, co), device="cuda", generator=gen) * 0.5).to(torch.bfloat16)
    x[..., ci:] = 7.0       # pitch padding must never reach the result
    dy[..., co:] = -5.0
    gw0 = torch.randn((co, ci, taps), device="cuda", generator=gen)
    gb0 = torch.randn((co,), device="cuda", generator=gen)
    gw, gb = gw0.clone(), gb0.clone()
    L.check(lib.jen1_big_gemm_tn_conv(dy.data_ptr(), x.data_ptr(), gw.data_ptr(), gb.data_ptr() if with_bias else None, B, T_out, T_in, co, ci, taps,
                                      stride, pad, ldy, ldx, 1.0, None, torch.cuda.current_stream().cuda_stream), "jen1_big_gemm_tn_conv")
    torch.cuda.synchronize()
    w = torch.zeros((co, ci, taps), device="cuda", requires_grad=True)
    bias = torch.zeros((co,), device="cuda", requires_grad=True)
    xin = x[..., :ci].float().permute(0, 2, 1)
    xin = F.pad(xin, (pad, taps - 1 - pad)) if taps > 1 else xin
    y = F.conv1d(xin, w, bias, stride=stride)
    assert y.shape[-1] == T_out, (y.shape, T_out)
    y.backward(dy[..., :co].float().permute(0, 2, 1))
    assert rel_err((gw - gw0).cpu().numpy(), w.grad.cpu().numpy()) < 2e-5
    if with_bias:
        assert rel_err((gb - gb0).cpu().numpy(), bias.grad.cpu().numpy()) < 2e-5
    else:
        assert torch.equal(gb, gb0)


@pytest.mark.parametrize("B,T_in,ci,co,taps,stride,pad,with_bias,with_res", [
    (3, 50, 128, 64, 3, 1, 1, True, False),       # centred k = 3
    (3, 50, 128, 128, 3, 1, 2, False, True),      # causal k = 3 + residual
    (2, 257, 256, 96, 5, 2, 2, True, False),      # strided k = 5, ragged rows, 96 output channels inside a 128-column tile
    (16, 1500, 128, 128, 3, 1, 1, True, True),    # the level-0 shape of the training pass
    (4, 129, 192, 72, 1, 1, 0, True, False),      # 1 x 1
    (3, 200, 264, 128, 3, 1, 1, True, False),     # 257 channels padded to 264: a ragged last K step per tap
    (2, 257, 128, 128, 5, 2, 2, False, False),    # strided: the data gradient divides in its row map
    (2, 300, 128, 128, 9, 4, 4, True, False),     # Downsample1d factor 4 (k = 2 f + 1)
])
def test_conv_form_forward_and_data_gradient_match_torch(B, T_in, ci, co, taps, stride, pad, with_bias, with_res):
    """jen1_big_gemm_conv: y = conv1d(x, W) (+ bias) (+ residual) through the row map of the matrix-core GEMM, and -- for stride 1 -- the
    data gradient as the same kernel on dy with the taps reversed and pad' = taps - 1 - pad, against torch conv1d and its autograd"""
    import torch.nn.functional as F
    lib = L.load()
    gen = torch.Generator(device="cuda").manual_seed(B * 17 + T_in + ci + taps)
    T_out = (T_in - 1) // stride + 1 if taps > 1 else T_in        # total padding taps - 1
    ldy = -(-co // 8) * 8
    x = (torch.randn((B, T_in, ci), device="cuda", generator=gen) * 0.5).to(torch.bfloat16)
    w = (torch.randn((co, ci, taps), device="cuda", generator=gen) * 0.1).to(torch.bfloat16)
    wp = w.permute(2, 0, 1).contiguous()                            # [taps][co][ci]: the forward compute copy
    bias = torch.randn((co,), device="cuda", generator=gen) if with_bias else None
    res = (torch.randn((B, T_out, ldy), device="cuda", generator=gen) * 0.5).to(torch.bfloat16) if with_res else None
    y = torch.full((B, T_out, ldy), 3.0, device="cuda", dtype=torch.bfloat16)
    s = torch.cuda.current_stream().cuda_stream
    L.check(lib.jen1_big_gemm_conv(x.data_ptr(), wp.data_ptr(), None if bias is None else bias.data_ptr(), None if res is None else res.data_ptr(),
                                   y.data_ptr(), B, T_in, T_out, ci, co, taps, stride, pad, 0, ci, ci, co * ci, ldy, None, 1, s), "jen1_big_gemm_conv")
    torch.cuda.synchronize()
    xin = x.float().permute(0, 2, 1).requires_grad_(True)
    xp = F.pad(xin, (pad, taps - 1 - pad)) if taps > 1 else xin
    ref = F.conv1d(xp, w.float(), bias, stride=stride)
    assert ref.shape[-1] == T_out
    want = ref.permute(0, 2, 1) + (res[..., :co].float() if with_res else 0.0)
    assert rel_err(y[..., :co].float().cpu().numpy(), want.detach().cpu().numpy()) < 6e-3          # (bf16 output rounding)
    dy = (torch.randn((B, T_out, co), device="cuda", generator=gen) * 0.5).to(torch.bfloat16)
    if co % 64:
        return                                                     # (the data gradient reduces over co: multiples of 64 only)
    wd = w.permute(2, 1, 0).contiguous()                            # [taps][ci][co]: the data-gradient twin
    dx = torch.zeros((B, T_in, ci), device="cuda", dtype=torch.bfloat16)
    L.check(lib.jen1_big_gemm_conv(dy.data_ptr(), wd.data_ptr(), None, None, dx.data_ptr(), B, T_out, T_in, co, ci, taps, 1, taps - 1 - pad, 1, co, co,
                                   ci * co, ci, None, stride, s), "jen1_big_gemm_conv")
    torch.cuda.synchronize()
    ref.backward(dy.float().permute(0, 2, 1))
    assert rel_err(dx.float().cpu().numpy(), xin.grad.permute(0, 2, 1).cpu().numpy()) < 6e-3


def test_conv_form_with_padding_per_batch_element():
    """causal (left pad k - 1) and centred clips side by side in one pass (trainer.py:189-211 merged; blocks.py:45-50): the row shift of
    jen1_big_gemm_conv / jen1_big_gemm_tn_conv per batch element, forward, data gradient and weight gradient against torch per clip"""
    import torch.nn.functional as F
    lib = L.load()
    gen = torch.Generator(device="cuda").manual_seed(99)
    B, T, ci, co, k = 4, 300, 128, 128, 3
    pads = [2, 1, 2, 1]
    x = (torch.randn((B, T, ci), device="cuda", generator=gen) * 0.5).to(torch.bfloat16)
    w = (torch.randn((co, ci, k), device="cuda", generator=gen) * 0.1).to(torch.bfloat16)
    dy = (torch.randn((B, T, co), device="cuda", generator=gen) * 0.5).to(torch.bfloat16)
    wp, wd = w.permute(2, 0, 1).contiguous(), w.permute(2, 1, 0).contiguous()
    fwd_shift = torch.tensor([-p for p in pads], dtype=torch.int32, device="cuda")
    bwd_shift = torch.tensor(pads, dtype=torch.int32, device="cuda")
    y = torch.zeros((B, T, co), device="cuda", dtype=torch.bfloat16)
    dx = torch.zeros((B, T, ci), device="cuda", dtype=torch.bfloat16)
    gw = torch.zeros((co, ci, k), device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    L.check(lib.jen1_big_gemm_conv(x.data_ptr(), wp.data_ptr(), None, None, y.data_ptr(), B, T, T, ci, co, k, 1, 0, 0, ci, ci, co * ci, co,
                                   fwd_shift.data_ptr(), 1, s), "fwd")
    L.check(lib.jen1_big_gemm_conv(dy.data_ptr(), wd.data_ptr(), None, None, dx.data_ptr(), B, T, T, co, ci, k, 1, k - 1, 1, co, co, ci * co, ci,
                                   bwd_shift.data_ptr(), 1, s), "dgrad")
    L.check(lib.jen1_big_gemm_tn_conv(dy.data_ptr(), x.data_ptr(), gw.data_ptr(), None, B, T, T, co, ci, k, 1, 0, co, ci, 1.0, fwd_shift.data_ptr(), s), "wgrad")
    torch.cuda.synchronize()
    wf = w.float().requires_grad_(True)
    ys, dxs = [], []
    for b in range(B):
        xin = x[b:b + 1].float().permute(0, 2, 1).requires_grad_(True)
        yb = F.conv1d(F.pad(xin, (pads[b], k - 1 - pads[b])), wf)
        yb.backward(dy[b:b + 1].float().permute(0, 2, 1))
        ys.append(yb.detach().permute(0, 2, 1))
        dxs.append(xin.grad.permute(0, 2, 1))
    assert rel_err(y.float().cpu().numpy(), torch.cat(ys).cpu().numpy()) < 6e-3
    assert rel_err(dx.float().cpu().numpy(), torch.cat(dxs).cpu().numpy()) < 6e-3
    assert rel_err(gw.cpu().numpy(), wf.grad.cpu().numpy()) < 2e-5


def test_conv_transpose_weight_gradient_tn_matches_torch():
    """nn.ConvTranspose1d (Upsample1d, blocks.py:80-88) W[ci][co][k]: its weight gradient on jen1_big_gemm_tn_conv with the layer's
    input as the row operand and dY as the shifted, strided one"""
    import torch.nn.functional as F
    lib = L.load()
    gen = torch.Generator(device="cuda").manual_seed(5)
    B, L_in, ci, co, k, stride, padding, opad = 3, 97, 128, 64, 8, 4, 2, 0
    L_out = (L_in - 1) * stride - 2 * padding + k + opad
    x = (torch.randn((B, L_in, ci), device="cuda", generator=gen) * 0.5).to(torch.bfloat16)
    dy = torch.zeros((B, L_out, 128), device="cuda", dtype=torch.bfloat16)
    dy[..., :co] = (torch.randn((B, L_out, co), device="cuda", generator=gen) * 0.5).to(torch.bfloat16)
    gw0 = torch.randn((ci, co, k), device="cuda", generator=gen)
    gw = gw0.clone()
    L.check(lib.jen1_big_gemm_tn_conv(x.data_ptr(), dy.data_ptr(), gw.data_ptr(), None, B, L_in, L_out, ci, co, k, stride, padding, ci, 128, 1.0, None,
                                      torch.cuda.current_stream().cuda_stream), "jen1_big_gemm_tn_conv")
    torch.cuda.synchronize()
    w = torch.zeros((ci, co, k), device="cuda", requires_grad=True)
    y = F.conv_transpose1d(x.float().permute(0, 2, 1), w, None, stride=stride, padding=padding, output_padding=opad)
    assert y.shape[-1] == L_out
    y.backward(dy[..., :co].float().permute(0, 2, 1))
    assert rel_err((gw - gw0).cpu().numpy(), w.grad.cpu().numpy()) < 2e-5


@pytest.mark.parametrize("k", [3, 1])
def test_conv_weight_gradient_tn_ragged_input_channels(k):
    """257 input channels in a pitch of 264 (the latent + context channels of `to_in`, model.py:240): the last 8-channel chunk is read up to
    the pitch and only the real channels reach gw"""
    import torch.nn.functional as F
    lib = L.load()
    gen = torch.Generator(device="cuda").manual_seed(11)
    B, T, ci, co, ldx = 4, 700, 257, 128, 264
    x = torch.full((B, T, ldx), 3.0, device="cuda", dtype=torch.bfloat16)
    x[..., :ci] = (torch.randn((B, T, ci), device="cuda", generator=gen) * 0.5).to(torch.bfloat16)
    dy = (torch.randn((B, T, co), device="cuda", generator=gen) * 0.5).to(torch.bfloat16)
    gw = torch.zeros((co, ci, k), device="cuda")
    gb = torch.zeros((co,), device="cuda")
    L.check(lib.jen1_big_gemm_tn_conv(dy.data_ptr(), x.data_ptr(), gw.data_ptr(), gb.data_ptr(), B, T, T, co, ci, k, 1, k // 2, co, ldx, 1.0, None,
                                      torch.cuda.current_stream().cuda_stream), "jen1_big_gemm_tn_conv")
    torch.cuda.synchronize()
    w = torch.zeros((co, ci, k), device="cuda", requires_grad=True)
    bias = torch.zeros((co,), device="cuda", requires_grad=True)
    y = F.conv1d(F.pad(x[..., :ci].float().permute(0, 2, 1), (k // 2, k // 2)), w, bias)
    y.backward(dy.float().permute(0, 2, 1))
    assert rel_err(gw.cpu().numpy(), w.grad.cpu().numpy()) < 2e-5
    assert rel_err(gb.cpu().numpy(), bias.grad.cpu().numpy()) < 2e-5


def test_conv_transpose_data_gradient_is_a_strided_conv_form():
    """nn.ConvTranspose1d (Upsample1d, blocks.py:80-88): dx[b, t][ci] = sum_tap dy[b, t stride + tap - padding][co] W[ci][co][tap] on
    jen1_big_gemm_conv (stride in the row map, the [k][Ci][Co] copy as the weight), against torch autograd"""
    import torch.nn.functional as F
    lib = L.load()
    gen = torch.Generator(device="cuda").manual_seed(21)
    B, L_in, ci, co, k, stride, padding = 3, 211, 128, 128, 8, 4, 2
    L_out = (L_in - 1) * stride - 2 * padding + k
    w = (torch.randn((ci, co, k), device="cuda", generator=gen) * 0.1).to(torch.bfloat16)
    dy = (torch.randn((B, L_out, co), device="cuda", generator=gen) * 0.5).to(torch.bfloat16)
    wd = w.permute(2, 0, 1).contiguous()                            # [k][ci][co]
    dx = torch.zeros((B, L_in, ci), device="cuda", dtype=torch.bfloat16)
    L.check(lib.jen1_big_gemm_conv(dy.data_ptr(), wd.data_ptr(), None, None, dx.data_ptr(), B, L_out, L_in, co, ci, k, stride, padding, 0, co, co,
                                   ci * co, ci, None, 1, torch.cuda.current_stream().cuda_stream), "jen1_big_gemm_conv")
    torch.cuda.synchronize()
    x = torch.zeros((B, ci, L_in), device="cuda", requires_grad=True)
    y = F.conv_transpose1d(x, w.float(), None, stride=stride, padding=padding)
    assert y.shape[-1] == L_out
    y.backward(dy.float().permute(0, 2, 1))
    assert rel_err(dx.float().cpu().numpy(), x.grad.permute(0, 2, 1).cpu().numpy()) < 6e-3


@pytest.mark.parametrize("stride,k,padding,opad", [(4, 8, 2, 0), (2, 4, 1, 0), (2, 5, 2, 1)])
def test_conv_transpose_forward_on_the_conv_form(stride, k, padding, opad):
    """nn.ConvTranspose1d forward (Upsample1d, blocks.py:80-88) as jen1_big_gemm_conv with the taps reversed and the stride as the
    divisor of its row map, against torch"""
    import torch.nn.functional as F
    lib = L.load()
    gen = torch.Generator(device="cuda").manual_seed(31 + stride + k)
    B, L_in, ci, co = 3, 157, 128, 64
    L_out = (L_in - 1) * stride - 2 * padding + k + opad
    x = (torch.randn((B, L_in, ci), device="cuda", generator=gen) * 0.5).to(torch.bfloat16)
    w = (torch.randn((ci, co, k), device="cuda", generator=gen) * 0.1).to(torch.bfloat16)
    bias = torch.randn((co,), device="cuda", generator=gen)
    wp = w.permute(2, 1, 0).contiguous()                            # [k][co][ci]
    y = torch.zeros((B, L_out, co), device="cuda", dtype=torch.bfloat16)
    L.check(lib.jen1_big_gemm_conv(x.data_ptr(), wp.data_ptr(), bias.data_ptr(), None, y.data_ptr(), B, L_in, L_out, ci, co, k, 1, k - 1 - padding, 1,
                                   ci, ci, co * ci, co, None, stride, torch.cuda.current_stream().cuda_stream), "jen1_big_gemm_conv")
    torch.cuda.synchronize()
    ref = F.conv_transpose1d(x.float().permute(0, 2, 1), w.float(), bias, stride=stride, padding=padding, output_padding=opad)
    assert ref.shape[-1] == L_out
    assert rel_err(y.float().cpu().numpy(), ref.permute(0, 2, 1).cpu().numpy()) < 6e-3


# ---------------------------------------------------------------------------------------------------------------------
# the 256 x 256 tile form (big_gemm_nt256_kernel) and the column split of a product between it and the 128 x 128 form
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(1024, 1024, 1024), (2064, 2048, 512), (300, 768, 128), (257, 256, 64), (1, 512, 64), (1161, 1280, 1024)])
def test_tile256_form_forced_matches_torch(monkeypatch, M, N, K):
    """JEN1_BGEMM_T256=1: every bf16 product runs on the 256 x 256 tiles -- ragged M (rows past M read zeros, are never stored), column
    counts that are not 256 multiples (one group: the last tile is cut by the buffer descriptor and the column test of the epilogue)"""
    monkeypatch.setenv("JEN1_BGEMM_T256", "1")
    gen = torch.Generator(device="cuda").manual_seed(M * 3 + N + K)
    a = (torch.randn((M, K), device="cuda", generator=gen) * 0.5).to(torch.bfloat16)
    b = (torch.randn((N, K), device="cuda", generator=gen) * 0.5).to(torch.bfloat16)
    guard = torch.full((M + 3, N), 7.0, device="cuda", dtype=torch.bfloat16)
    c = guard[:M]
    _run(a, b, [(c, None, 0, N)])
    ref = a.float() @ b.float().t()
    assert float((c.float() - ref).abs().max() / ref.abs().max()) < 1e-2
    assert bool((guard[M:] == 7.0).all())
    # the other form on the same operands: the two kernels accumulate in the same order along K (bit-identical results)
    monkeypatch.setenv("JEN1_BGEMM_T256", "0")
    c2 = torch.empty_like(c)
    _run(a, b, [(c2, None, 0, N)])
    assert torch.equal(c, c2)


def test_tile256_epilogue_accumulate_alpha_float32_output(monkeypatch):
    monkeypatch.setenv("JEN1_BGEMM_T256", "1")
    M, N, K = 515, 512, 256
    gen = torch.Generator(device="cuda").manual_seed(3)
    a = (torch.randn((M, K), device="cuda", generator=gen) * 0.5).to(torch.bfloat16)
    b = (torch.randn((N, K), device="cuda", generator=gen) * 0.5).to(torch.bfloat16)
    c = torch.randn((M, N), device="cuda", generator=gen)
    want = c + 0.25 * (a.float() @ b.float().t())
    _run(a, b, [(c, None, 0, N)], c_f32=True, accumulate=True, alpha=0.25)
    assert float((c - want).abs().max() / want.abs().max()) < 1e-5


@pytest.mark.parametrize("B", [8, 3])
def test_stacked_projection_splits_between_the_two_tile_forms(B):
    """the sampling plan's shape: 13 column groups (widths 512 / 1024 / 2048, 256-column multiples: group_align = 256) over B x 128 rows,
    row map into [B][129] caches, padding mask, bias.  B = 8: 4 x 68 big tiles = one full round on 256 CUs + 64 columns-of-1024 on the
    128 x 128 form (the split lands on a group boundary); B = 3: too few tiles, everything on the small form.  Same results either way."""
    NL, K = 128, 1024
    widths = [512] * 2 + [1024] * 7 + [2048] * 4          # (the full model's 13 cross-attention layers: 17408 columns)
    gen = torch.Generator(device="cuda").manual_seed(11 + B)
    xs = (torch.randn((B * NL, K), device="cuda", generator=gen)).to(torch.bfloat16)
    w = (torch.randn((sum(widths), K), device="cuda", generator=gen) * 0.05).to(torch.bfloat16)
    mask_b = (torch.rand((B, NL + 1), device="cuda", generator=gen) > 0.3).float()
    mask = mask_b[:, :NL].reshape(-1)
    outs, groups, n0 = [], [], 0
    for wd in widths:
        cache = torch.full((B, NL + 1, wd), -3.0, device="cuda", dtype=torch.bfloat16)
        bias = torch.randn((wd,), device="cuda", generator=gen)
        outs.append((cache, bias, n0, wd))
        groups.append((cache.view(B * (NL + 1), wd), bias, n0, wd))
        n0 += wd
    _run(xs, w, groups, row_scale=mask_b, rows_in=NL, rows_out=NL + 1, group_align=256)
    for cache, bias, n0, wd in outs:
        ref = (xs.float() @ w[n0:n0 + wd].float().t() + bias[None]) * mask[:, None]
        got = cache[:, :NL].reshape(B * NL, wd).float()
        assert float((got - ref).abs().max() / ref.abs().max()) < 1e-2, (n0, wd)
        assert bool((cache[:, NL] == -3.0).all())          # the time token's row is not this launch's


# ---------------------------------------------------------------------------------------------------------------------
# the four-stage forms (big_gemm_nt_s4_kernel<256 / 272>): K step 32, three tiles in flight, 16 x 16 x 32 MFMA, 64-byte-row swizzle
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tnw,M,N,K", [(272, 1024, 544, 512), (272, 300, 272, 128), (272, 257, 816, 64), (272, 1, 272, 64), (272, 515, 1088, 192)])
def test_four_stage_form_forced_matches_torch(monkeypatch, tnw, M, N, K):
    """K-step counts around the pipeline depth (2, 4, 6, 16 steps of 32), ragged M, one to four column tiles"""
    monkeypatch.setenv("JEN1_BGEMM_S4", str(tnw))
    gen = torch.Generator(device="cuda").manual_seed(M * 5 + N + K + tnw)
    a = (torch.randn((M, K), device="cuda", generator=gen) * 0.5).to(torch.bfloat16)
    b = (torch.randn((N, K), device="cuda", generator=gen) * 0.5).to(torch.bfloat16)
    guard = torch.full((M + 3, N), 7.0, device="cuda", dtype=torch.bfloat16)
    c = guard[:M]
    _run(a, b, [(c, None, 0, N)])
    ref = a.float() @ b.float().t()
    assert float((c.float() - ref).abs().max() / ref.abs().max()) < 1e-2
    assert bool((guard[M:] == 7.0).all())
    monkeypatch.setenv("JEN1_BGEMM_S4", "0")
    c2 = torch.empty_like(c)
    _run(a, b, [(c2, None, 0, N)])
    assert float((c.float() - c2.float()).abs().max() / ref.abs().max()) < 1e-2


@pytest.mark.parametrize("force", ["auto", "0"])
def test_stacked_projection_on_the_272_column_tiles(monkeypatch, force):
    """B = 8: 4 x 64 tiles of 256 x 272 = one round of one tile per CU (chosen automatically); the column tiles straddle the 13 groups
    (layer widths 512 / 1024 / 2048 are no multiples of 272): the epilogue finds the group of every 16-column block.  Same result as the
    other forms; row map, padding mask, bias, the untouched 129th row as in the sampling plan."""
    if force != "auto":
        monkeypatch.setenv("JEN1_BGEMM_S4", force)
    B, NL, K = 8, 128, 1024
    widths = [512] * 2 + [1024] * 7 + [2048] * 4
    gen = torch.Generator(device="cuda").manual_seed(23)
    xs = (torch.randn((B * NL, K), device="cuda", generator=gen)).to(torch.bfloat16)
    w = (torch.randn((sum(widths), K), device="cuda", generator=gen) * 0.05).to(torch.bfloat16)
    mask_b = (torch.rand((B, NL + 1), device="cuda", generator=gen) > 0.3).float()
    mask = mask_b[:, :NL].reshape(-1)
    outs, groups, n0 = [], [], 0
    for wd in widths:
        cache = torch.full((B, NL + 1, wd), -3.0, device="cuda", dtype=torch.bfloat16)
        bias = torch.randn((wd,), device="cuda", generator=gen)
        outs.append((cache, bias, n0, wd))
        groups.append((cache.view(B * (NL + 1), wd), bias, n0, wd))
        n0 += wd
    _run(xs, w, groups, row_scale=mask_b, rows_in=NL, rows_out=NL + 1, group_align=256)
    for cache, bias, n0, wd in outs:
        ref = (xs.float() @ w[n0:n0 + wd].float().t() + bias[None]) * mask[:, None]
        got = cache[:, :NL].reshape(B * NL, wd).float()
        assert float((got - ref).abs().max() / ref.abs().max()) < 1e-2, (n0, wd)
        assert bool((cache[:, NL] == -3.0).all())
