"""``torch.ops.jen1.*``: the entry points of libjen1_hip.so registered with the PyTorch dispatcher (``torch.library``).

BASELINE.json's north_star asks for "Python host code calling hand-written HIP through PyTorch-ROCm custom ops".  The C ABI
(include/jen1_hip.h, jen1_train.h, jen1_deep.h) is the drop-in boundary; this module puts dispatcher schemas, fake (meta)
implementations and -- for the stateless training operators -- autograd formulas on top of it, so that the ops are visible
to ``torch.compile`` / ``torch.export`` / FakeTensor tracing and show up by name in the profiler:

  jen1::unet_cfg_forward     the whole denoiser (reference jen1/model/model.py:299-376): what ``UNetCFG1d.forward`` runs
  jen1::cfg_combine          CFG combine + std rescale of a stacked (cond, uncond) pair (model.py:362-369)
  jen1::group_norm           GroupNorm (+FiLM) (+SiLU) on channel-last rows (blocks.py:137-143, :509), differentiable
  jen1::layer_norm           LayerNorm over the last axis (blocks.py:400-401), differentiable
  jen1::activation           GELU(erf) / SiLU (blocks.py:443, :158), differentiable
  jen1::conv_forward         _Conv1d / nn.Conv1d / nn.ConvTranspose1d / nn.Linear (blocks.py:34-53, :69-95), differentiable: ONE backward
                             op returns the data, weight and bias gradients (``conv1d_same`` / ``conv_transpose1d`` / ``linear`` below)
  jen1::attention            the attention core of AttentionBase (blocks.py:300-330, :431-434), differentiable

There is no CPU implementation behind any of them: without the HIP extension (or off a ROCm device) they raise ``Jen1HipError``.
The training graph (jen1_amd/train.py) keeps its own ``autograd.Function``s for the parameterised operators: those accumulate
weight gradients in place into the flat ``.grad`` buffer RCCL reduces and read compute copies of the weights that are refreshed once
per optimiser step; the functional ops here run the same kernels through the same host code (train._conv_forward / _conv_dgrad /
_conv_wgrad, AttentionCoreFn) but return the gradients as tensors and pack the weight per call.
"""
from __future__ import annotations

import weakref
from typing import Optional, Tuple

import torch
from torch.library import custom_op

from . import lib as L

_models: "weakref.WeakValueDictionary[int, torch.nn.Module]" = weakref.WeakValueDictionary()
_next_handle = [1]


def register_model(model: torch.nn.Module) -> int:
    """an integer handle for ``jen1::unet_cfg_forward`` (dispatcher schemas carry tensors and scalars, not modules)"""
    h = _next_handle[0]
    _next_handle[0] += 1
    _models[h] = model
    return h


def _model(handle: int):
    m = _models.get(int(handle))
    if m is None:
        raise L.Jen1HipError(f"jen1::unet_cfg_forward: no live model behind handle {handle}")
    return m


def _stream(t: torch.Tensor) -> int:
    if t.device.type != "cuda":
        raise L.Jen1HipError("torch.ops.jen1.*: the operators run on a ROCm GPU only (no CPU path exists in this package)")
    return torch.cuda.current_stream(t.device).cuda_stream


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return L.F32
    if t.dtype == torch.bfloat16:
        return L.BF16
    raise L.Jen1HipError(f"torch.ops.jen1.*: dtype {t.dtype} (float32 or bfloat16)")


# ------------------------------------------------------------------------------------------------------------------ denoiser
@custom_op("jen1::unet_cfg_forward", mutates_args=())
def unet_cfg_forward(handle: int, x: torch.Tensor, time: torch.Tensor, embedding: torch.Tensor, embedding_mask: Optional[torch.Tensor],
                     context: Optional[torch.Tensor], embedding_scale: float, embedding_mask_proba: float, batch_cfg: bool, scale_cfg: bool,
                     scale_phi: float, causal: bool, dropout_rows: Optional[torch.Tensor]) -> torch.Tensor:
    m = _model(handle)
    return m._forward_impl(x, time, embedding=embedding, embedding_mask=embedding_mask, embedding_scale=embedding_scale,
                           embedding_mask_proba=embedding_mask_proba, batch_cfg=batch_cfg, scale_cfg=scale_cfg, scale_phi=scale_phi,
                           channels_list=None if context is None else [context], causal=causal, dropout_rows=dropout_rows)


@unet_cfg_forward.register_fake
def _(handle, x, time, embedding, embedding_mask, context, embedding_scale, embedding_mask_proba, batch_cfg, scale_cfg, scale_phi, causal,
      dropout_rows):
    m = _model(handle)
    return x.new_empty((x.shape[0], m.spec.out_channels, x.shape[2]), dtype=torch.float32)


@custom_op("jen1::cfg_combine", mutates_args=())
def cfg_combine(net: torch.Tensor, channels: int, embedding_scale: float, scale_cfg: bool, scale_phi: float) -> torch.Tensor:
    """net: channel-last [2B, T, ld] (rows [0, B) conditional, [B, 2B) unconditional) -> guided output [B, channels, T] float32"""
    assert net.dim() == 3 and net.shape[0] % 2 == 0 and net.is_contiguous() and channels <= net.shape[2]
    B, T, ld = net.shape[0] // 2, net.shape[1], net.shape[2]
    out = torch.empty((B, channels, T), dtype=torch.float32, device=net.device)
    L.check(L.load().jen1_cfg_combine(net.data_ptr(), out.data_ptr(), B, channels, T, ld, float(embedding_scale), 1 if scale_cfg else 0,
                                      float(scale_phi), _dt(net), _stream(net)), "jen1_cfg_combine")
    return out


@cfg_combine.register_fake
def _(net, channels, embedding_scale, scale_cfg, scale_phi):
    return net.new_empty((net.shape[0] // 2, channels, net.shape[1]), dtype=torch.float32)


# ------------------------------------------------------------------------------------------------------------------ GroupNorm
@custom_op("jen1::group_norm", mutates_args=())
def group_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, film: Optional[torch.Tensor], channels: int, groups: int,
               eps: float, silu: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """x: channel-last [B, L, ld >= channels]; film: [B, >= 2 channels] (scale | shift) in x's dtype.  Returns (y, sums[B][G][2])."""
    assert x.dim() == 3 and x.is_contiguous() and x.shape[2] >= channels
    B, Lx, ld = x.shape
    lib, s, dt = L.load(), _stream(x), _dt(x)
    sums = torch.empty((B, groups, 2), dtype=torch.float32, device=x.device)
    y = (torch.zeros_like if ld != channels else torch.empty_like)(x)
    if film is not None:
        film = film.contiguous()
        assert film.dtype == x.dtype and film.shape[-1] >= 2 * channels
    L.check(lib.jen1_gn_sums(x.data_ptr(), sums.data_ptr(), B, Lx, channels, ld, groups, dt, s), "jen1_gn_sums")
    L.check(lib.jen1_gn_apply(x.data_ptr(), sums.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None if film is None else film.data_ptr(),
                              0 if film is None else film.shape[-1], y.data_ptr(), B, Lx, channels, ld, groups, float(eps), 1 if silu else 0,
                              dt, s), "jen1_gn_apply")
    return y, sums


@group_norm.register_fake
def _(x, gamma, beta, film, channels, groups, eps, silu):
    return torch.empty_like(x), x.new_empty((x.shape[0], groups, 2), dtype=torch.float32)


@custom_op("jen1::group_norm_backward", mutates_args=())
def group_norm_backward(dy: torch.Tensor, x: torch.Tensor, sums: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                        film: Optional[torch.Tensor], channels: int, groups: int, eps: float,
                        silu: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """(dx, dgamma, dbeta, dfilm[B, 2 channels] float32 -- zeros without FiLM)"""
    B, Lx, ld = x.shape
    lib, s, dt = L.load(), _stream(x), _dt(x)
    dy = dy.contiguous()
    dx = (torch.zeros_like if ld != channels else torch.empty_like)(x)
    P = torch.empty((B, channels, 4), dtype=torch.float32, device=x.device)
    Gm = torch.empty((B, groups, 2), dtype=torch.float32, device=x.device)
    dg = torch.zeros((channels,), dtype=torch.float32, device=x.device)
    db = torch.zeros((channels,), dtype=torch.float32, device=x.device)
    dfilm = torch.zeros((B, 2 * channels), dtype=torch.float32, device=x.device)
    if film is not None:
        film = film.contiguous()
    L.check(lib.jen1_gn_backward(dy.data_ptr(), x.data_ptr(), sums.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                 None if film is None else film.data_ptr(), 0 if film is None else film.shape[-1], dx.data_ptr(),
                                 dg.data_ptr(), db.data_ptr(), None if film is None else dfilm.data_ptr(), P.data_ptr(), Gm.data_ptr(), B, Lx,
                                 channels, ld, groups, float(eps), 1 if silu else 0, dt, s), "jen1_gn_backward")
    return dx, dg, db, dfilm


@group_norm_backward.register_fake
def _(dy, x, sums, gamma, beta, film, channels, groups, eps, silu):
    f32 = torch.float32
    return (torch.empty_like(x), x.new_empty((channels,), dtype=f32), x.new_empty((channels,), dtype=f32),
            x.new_empty((x.shape[0], 2 * channels), dtype=f32))


def _gn_setup(ctx, inputs, output):
    x, gamma, beta, film, channels, groups, eps, silu = inputs
    ctx.save_for_backward(x, output[1], gamma, beta, film if film is not None else x.new_empty(0))
    ctx.has_film, ctx.channels, ctx.groups, ctx.eps, ctx.silu = film is not None, channels, groups, eps, silu


def _gn_backward(ctx, dy, dsums):
    x, sums, gamma, beta, film = ctx.saved_tensors
    dx, dg, db, dfilm = torch.ops.jen1.group_norm_backward(dy, x, sums, gamma, beta, film if ctx.has_film else None, ctx.channels,
                                                           ctx.groups, ctx.eps, ctx.silu)
    df = None
    if ctx.has_film:
        df = torch.zeros((x.shape[0], film.shape[-1]), dtype=film.dtype, device=x.device)
        df[:, : 2 * ctx.channels] = dfilm
    pad = gamma.shape[0] - ctx.channels
    if pad > 0:
        dg, db = torch.nn.functional.pad(dg, (0, pad)), torch.nn.functional.pad(db, (0, pad))
    return dx, dg.to(gamma.dtype), db.to(beta.dtype), df, None, None, None, None


group_norm.register_autograd(_gn_backward, setup_context=_gn_setup)


# ------------------------------------------------------------------------------------------------------------------ LayerNorm
@custom_op("jen1::layer_norm", mutates_args=())
def layer_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """LayerNorm over the first ``gamma.numel()`` columns of the last axis.  Returns (y, stats[rows][2] = (mean, rstd))."""
    x = x.contiguous()
    C, ld = gamma.shape[0], x.shape[-1]
    rows = x.numel() // ld
    y = (torch.zeros_like if ld != C else torch.empty_like)(x)
    stats = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
    L.check(L.load().jen1_ln_forward(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), stats.data_ptr(), rows, C, ld, float(eps),
                                     _dt(x), _stream(x)), "jen1_ln_forward")
    return y, stats


@layer_norm.register_fake
def _(x, gamma, beta, eps):
    return torch.empty_like(x), x.new_empty((x.numel() // x.shape[-1], 2), dtype=torch.float32)


@custom_op("jen1::layer_norm_backward", mutates_args=())
def layer_norm_backward(dy: torch.Tensor, x: torch.Tensor, stats: torch.Tensor, gamma: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    x, dy = x.contiguous(), dy.contiguous()
    C, ld = gamma.shape[0], x.shape[-1]
    rows = x.numel() // ld
    dx = (torch.zeros_like if ld != C else torch.empty_like)(x)
    dg = torch.zeros((C,), dtype=torch.float32, device=x.device)
    db = torch.zeros((C,), dtype=torch.float32, device=x.device)
    L.check(L.load().jen1_ln_backward(dy.data_ptr(), x.data_ptr(), stats.data_ptr(), gamma.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                      rows, C, ld, _dt(x), _stream(x)), "jen1_ln_backward")
    return dx, dg, db


@layer_norm_backward.register_fake
def _(dy, x, stats, gamma):
    return torch.empty_like(x), x.new_empty((gamma.shape[0],), dtype=torch.float32), x.new_empty((gamma.shape[0],), dtype=torch.float32)


def _ln_setup(ctx, inputs, output):
    x, gamma, beta, eps = inputs
    ctx.save_for_backward(x, output[1], gamma)


def _ln_backward(ctx, dy, dstats):
    x, stats, gamma = ctx.saved_tensors
    dx, dg, db = torch.ops.jen1.layer_norm_backward(dy, x, stats, gamma)
    return dx, dg.to(gamma.dtype), db.to(gamma.dtype), None


layer_norm.register_autograd(_ln_backward, setup_context=_ln_setup)


# ------------------------------------------------------------------------------------------------------------------ activations
@custom_op("jen1::activation", mutates_args=())
def activation(x: torch.Tensor, mode: int) -> torch.Tensor:
    """mode 0: GELU(erf), 1: SiLU, 2: ELU"""
    x = x.contiguous()
    y = torch.empty_like(x)
    L.check(L.load().jen1_act_forward(x.data_ptr(), y.data_ptr(), x.numel(), int(mode), _dt(x), _stream(x)), "jen1_act_forward")
    return y


@activation.register_fake
def _(x, mode):
    return torch.empty_like(x)


@custom_op("jen1::activation_backward", mutates_args=())
def activation_backward(dy: torch.Tensor, x: torch.Tensor, mode: int) -> torch.Tensor:
    x, dy = x.contiguous(), dy.contiguous()
    dx = torch.empty_like(x)
    L.check(L.load().jen1_act_backward(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), x.numel(), int(mode), _dt(x), _stream(x)), "jen1_act_backward")
    return dx


@activation_backward.register_fake
def _(dy, x, mode):
    return torch.empty_like(x)


def _act_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0])
    ctx.mode = inputs[1]


activation.register_autograd(lambda ctx, dy: (torch.ops.jen1.activation_backward(dy, ctx.saved_tensors[0], ctx.mode), None),
                             setup_context=_act_setup)


# ------------------------------------------------------------------------------------------------------------------ conv / linear
# The GEMM-shaped training operators (a1, a2, a8 of SURVEY.md section 8): _Conv1d (blocks.py:34-53), nn.Conv1d / nn.ConvTranspose1d of
# Upsample1d (blocks.py:69-95) and nn.Linear, forward on jen1_train_gemm / jen1_big_gemm_conv and -- as ONE backward op -- the data
# gradient, the weight gradient and the bias gradient on jen1_train_gemm / jen1_big_gemm_tn_conv / jen1_colsum.  Functional form:
# the gradients come back as tensors (the training graph's ConvFn accumulates them in place into the flat ``.grad`` buffer instead)
# and the compute copy of the weight ([k][C_out][pad8(C_in)], the layout the kernels read) is made per call.
_KINDS = ("conv", "convT", "linear")
_rts: dict = {}


def _rt(t: torch.Tensor):
    from .train import TrainRuntime
    _stream(t)
    key = (t.device.index, t.dtype)
    rt = _rts.get(key)
    if rt is None:
        rt = _rts[key] = TrainRuntime("f32" if _dt(t) == L.F32 else "bf16", t.device)
    return rt


def _pad8(c: int) -> int:
    return (c + 7) // 8 * 8


def _compute_copy(rt, w: torch.Tensor, kind: str, dtype: torch.dtype) -> torch.Tensor:
    d = rt._layout(w, kind)
    k, r, c = d.shape
    out = torch.zeros((k, r, _pad8(c)), dtype=dtype, device=w.device)
    out[:, :, :c] = d
    return out


def _geom(kind: int, weight: torch.Tensor, x: torch.Tensor, stride: int, pad: int, L_out: int):
    from .train import ConvGeom
    if kind == 2:
        co, ci = weight.shape
        rows = x.numel() // x.shape[-1]
        return ConvGeom("linear", 1, 1, 0, rows, rows, ci, co)
    if kind == 0:
        co, ci, k = weight.shape
    else:
        ci, co, k = weight.shape
    return ConvGeom(_KINDS[kind], k, int(stride), int(pad), x.shape[1], int(L_out), ci, co)


@custom_op("jen1::conv_forward", mutates_args=())
def conv_forward(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], kind: int, stride: int, pad: int, L_out: int) -> torch.Tensor:
    """x: channel-last [B, L_in, pad8(C_in)] in the compute dtype (padding lanes zero); weight in the reference layout
    (kind 0 Conv1d [Co, Ci, k], 1 ConvTranspose1d [Ci, Co, k], 2 Linear [Co, Ci]); bias float32 [Co] or None.  kind 0: output position
    t reads input rows t * stride - pad + tap (rows outside [0, L_in) are zeros); kind 1: ``pad`` is ConvTranspose1d's padding;
    kind 2: L_out is ignored (rows in = rows out).  -> [B, L_out, pad8(Co)]"""
    from .train import _conv_forward
    assert 0 <= kind <= 2 and x.dim() == 3
    rt = _rt(x)
    x = x.contiguous()
    g = _geom(kind, weight, x, stride, pad, L_out)
    assert x.shape[-1] == _pad8(g.ci), f"jen1::conv_forward: {x.shape[-1]} input lanes for {g.ci} channels (pad to a multiple of 8)"
    wp = _compute_copy(rt, weight, _KINDS[kind], x.dtype)
    xin = x.view(1, -1, x.shape[-1]) if kind == 2 else x
    y = _conv_forward(rt, xin, wp, None if bias is None else bias.detach().to(torch.float32).contiguous(), g)
    return y.view(x.shape[0], x.shape[1], y.shape[-1]) if kind == 2 else y


def _conv_out_len(kind, x, weight, L_out):
    return x.shape[1] if kind == 2 else int(L_out)


@conv_forward.register_fake
def _(x, weight, bias, kind, stride, pad, L_out):
    co = weight.shape[1] if kind == 1 else weight.shape[0]
    return x.new_empty((x.shape[0], _conv_out_len(kind, x, weight, L_out), _pad8(co)))


@custom_op("jen1::conv_backward", mutates_args=())
def conv_backward(dy: torch.Tensor, x: torch.Tensor, weight: torch.Tensor, kind: int, stride: int, pad: int, has_bias: bool,
                  need_dx: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(dx [x's shape; empty(0) unless need_dx], dweight float32 in the reference layout, dbias float32 [Co]; empty(0) unless has_bias)"""
    from .train import _conv_dgrad, _conv_wgrad
    rt = _rt(x)
    x, dy = x.contiguous(), dy.contiguous()
    g = _geom(kind, weight, x, stride, pad, dy.shape[1])
    xin = x.view(1, -1, x.shape[-1]) if kind == 2 else x
    dyin = dy.view(1, -1, dy.shape[-1]) if kind == 2 else dy
    dw = torch.zeros(weight.shape, dtype=torch.float32, device=x.device)
    db = torch.zeros((g.co,), dtype=torch.float32, device=x.device) if has_bias else x.new_empty((0,), dtype=torch.float32)
    if not _conv_wgrad(rt, xin, dyin, dw, g, db if has_bias else None) and has_bias:
        ldy = dy.shape[-1]
        L.check(rt.lib.jen1_colsum(dy.data_ptr(), db.data_ptr(), dy.numel() // ldy, g.co, ldy, rt.dt_of(dy), rt.stream()), "jen1_colsum")
    if not need_dx:
        return x.new_empty((0,)), dw, db
    wp = _compute_copy(rt, weight, _KINDS[kind], x.dtype)
    wd = _compute_copy(rt, weight, _KINDS[kind] + "D", x.dtype)
    return _conv_dgrad(rt, dyin, wp, g, wd).view(x.shape), dw, db


@conv_backward.register_fake
def _(dy, x, weight, kind, stride, pad, has_bias, need_dx):
    co = weight.shape[1] if kind == 1 else weight.shape[0]
    f32 = torch.float32
    return (torch.empty_like(x) if need_dx else x.new_empty((0,)), weight.new_empty(weight.shape, dtype=f32),
            x.new_empty((co if has_bias else 0,), dtype=f32))


def _conv_setup(ctx, inputs, output):
    x, weight, bias, kind, stride, pad, L_out = inputs
    ctx.save_for_backward(x, weight)
    ctx.kind, ctx.stride, ctx.pad, ctx.has_bias = kind, stride, pad, bias is not None
    ctx.bias_dtype = None if bias is None else bias.dtype


def _conv_backward(ctx, dy):
    x, weight = ctx.saved_tensors
    dx, dw, db = torch.ops.jen1.conv_backward(dy, x, weight, ctx.kind, ctx.stride, ctx.pad, ctx.has_bias, ctx.needs_input_grad[0])
    return (dx if ctx.needs_input_grad[0] else None, dw.to(weight.dtype), db.to(ctx.bias_dtype) if ctx.has_bias else None, None, None, None, None)


conv_forward.register_autograd(_conv_backward, setup_context=_conv_setup)


def conv1d_same(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], stride: int = 1, causal: bool = False) -> torch.Tensor:
    """_Conv1d (blocks.py:34-53): total padding k - 1, all of it on the left when causal, else split evenly"""
    k = weight.shape[2]
    return torch.ops.jen1.conv_forward(x, weight, bias, 0, stride, (k - 1) if causal else (k - 1) // 2, (x.shape[1] - 1) // stride + 1)


def conv_transpose1d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], stride: int, padding: int, output_padding: int) -> torch.Tensor:
    """nn.ConvTranspose1d of Upsample1d (blocks.py:80-88)"""
    k = weight.shape[2]
    return torch.ops.jen1.conv_forward(x, weight, bias, 1, stride, padding, (x.shape[1] - 1) * stride - 2 * padding + k + output_padding)


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.Linear on the last axis of [B, L, pad8(C_in)]"""
    return torch.ops.jen1.conv_forward(x, weight, bias, 2, 1, 0, x.shape[1])


# ------------------------------------------------------------------------------------------------------------------ attention core
class _Ctx:
    """what AttentionCoreFn's static methods need of an autograd context"""

    def save_for_backward(self, *t):
        self.saved_tensors = t


@custom_op("jen1::attention", mutates_args=())
def attention(q: torch.Tensor, kv: torch.Tensor, heads: int, causal: bool, kv_mask: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """softmax(q k^T / sqrt(d) [+ causal mask]) v per head (blocks.py:300-330 AttentionBase; kv = to_kv's output K | V [B, Nk, 2 C],
    kv_mask [B, Nk] multiplied into the rows of K and V, blocks.py:431-434).  -> (out [B, Nq, C], probabilities [B heads, Nq, pad8(Nk)])"""
    from .train import AttentionCoreFn
    rt = _rt(q)
    q, kv = q.contiguous(), kv.contiguous()
    small = bool(rt.small_attn and rt.lib.jen1_attn_small_fits(q.shape[1], kv.shape[1], q.shape[2] // heads, rt.dt_of(q)))
    if kv_mask is not None and not small:
        kv, kv_mask = kv * kv_mask.to(kv.dtype)[:, :, None], None
    ctx = _Ctx()
    out = AttentionCoreFn.forward(ctx, q, kv, rt, int(heads), bool(causal), kv_mask)
    return out, ctx.saved_tensors[2]


@attention.register_fake
def _(q, kv, heads, causal, kv_mask):
    return torch.empty_like(q), q.new_empty((q.shape[0] * heads, q.shape[1], _pad8(kv.shape[1])))


@custom_op("jen1::attention_backward", mutates_args=())
def attention_backward(dout: torch.Tensor, q: torch.Tensor, kv: torch.Tensor, probs: torch.Tensor, heads: int,
                       kv_mask: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """(dq, dkv) from the saved probabilities"""
    from .train import AttentionCoreFn
    rt = _rt(q)
    q, kv = q.contiguous(), kv.contiguous()
    ctx = _Ctx()
    ctx.rt, ctx.heads, ctx.scale = rt, int(heads), (q.shape[2] // heads) ** -0.5
    ctx.small = bool(rt.small_attn and rt.lib.jen1_attn_small_fits(q.shape[1], kv.shape[1], q.shape[2] // heads, rt.dt_of(q)))
    ctx.kv_row = ctx.dkv_slot = None
    mask = None if kv_mask is None else kv_mask.to(torch.float32).contiguous()
    ctx.kv_mask = mask if ctx.small else None
    kvm = kv if (mask is None or ctx.small) else kv * mask.to(kv.dtype)[:, :, None]
    ctx.saved_tensors = (q, kvm, probs)
    dq, dkv = AttentionCoreFn.backward(ctx, dout)[:2]
    if mask is not None and not ctx.small:
        dkv = dkv * mask.to(dkv.dtype)[:, :, None]
    return dq, dkv


@attention_backward.register_fake
def _(dout, q, kv, probs, heads, kv_mask):
    return torch.empty_like(q), torch.empty_like(kv)


def _attn_setup(ctx, inputs, output):
    q, kv, heads, causal, kv_mask = inputs
    ctx.save_for_backward(q, kv, output[1], kv_mask if kv_mask is not None else q.new_empty(0))
    ctx.heads, ctx.has_mask = heads, kv_mask is not None


def _attn_backward(ctx, dout, dprobs):
    q, kv, probs, mask = ctx.saved_tensors
    dq, dkv = torch.ops.jen1.attention_backward(dout, q, kv, probs, ctx.heads, mask if ctx.has_mask else None)
    return dq, dkv, None, None, None


attention.register_autograd(_attn_backward, setup_context=_attn_setup)

OPS = ("unet_cfg_forward", "cfg_combine", "group_norm", "group_norm_backward", "layer_norm", "layer_norm_backward", "activation",
       "activation_backward", "conv_forward", "conv_backward", "attention", "attention_backward")
