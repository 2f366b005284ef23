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

There is no CPU implementation behind any of them: without the HIP extension (or off a ROCm device) they raise ``Jen1HipError``.
The training graph (jen1_amd/train.py) keeps its own ``autograd.Function``s for the parameterised operators: those accumulate
weight gradients in place into the flat ``.grad`` buffer RCCL reduces, which a functional custom op cannot express without a
copy per parameter.
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

OPS = ("unet_cfg_forward", "cfg_combine", "group_norm", "group_norm_backward", "layer_norm", "layer_norm_backward", "activation",
       "activation_backward")
