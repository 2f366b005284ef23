"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY, NOT A PRODUCT PATH.

A plain-numpy restatement of the JEN-1 denoiser hot path of
0417keito/JEN-1-pytorch: UNetCFG1d.forward iterated by the DDIM / DDPM sampler
and the training loss.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module; nothing under
``jen-1-pytorch_amd/`` does, and the product path raises when the HIP library
is missing instead of falling back to this file.

Parity pin: the reference ships no tests or golden vectors, so
this oracle is pinned against outputs of the reference itself, imported on CPU
in the build container by ``tests/golden/make_golden.py``; the resulting
fixtures live in ``tests/golden/*.npz`` and ``tests/test_oracle_golden.py``
checks this file against every one of them.

Every function cites the reference lines it restates (paths relative to
/root/reference).  Tensors are numpy arrays in the reference's own layouts
([B, C, T] for conv / norm, [B, N, C] inside attention).  ``dtype`` selects
float32 (the reference arithmetic) or float64 (a higher-precision witness).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
from scipy.special import erf as _erf

Array = np.ndarray


# ----------------------------------------------------------------------------
# elementary ops
# ----------------------------------------------------------------------------
def silu(x: Array) -> Array:
    """nn.SiLU (jen1/model/blocks.py:126)."""
    return x / (1.0 + np.exp(-x))


def gelu(x: Array) -> Array:
    """nn.GELU, exact erf form (jen1/model/blocks.py:299,444; model.py:77-98)."""
    return (0.5 * x * (1.0 + _erf(x * (1.0 / math.sqrt(2.0))))).astype(x.dtype)


def linear(x: Array, w: Array, b: Optional[Array] = None) -> Array:
    """nn.Linear: x[..., i] @ w[o, i]^T + b."""
    y = x @ w.T
    return y + b if b is not None else y


def group_norm(x: Array, groups: int, gamma: Array, beta: Array, eps: float) -> Array:
    """nn.GroupNorm over (C/G, T) per sample (blocks.py:117-121, :509)."""
    B, C, T = x.shape
    xg = x.reshape(B, groups, (C // groups) * T)
    mu = xg.mean(axis=2, keepdims=True)
    var = ((xg - mu) ** 2).mean(axis=2, keepdims=True)
    xn = ((xg - mu) / np.sqrt(var + x.dtype.type(eps))).reshape(B, C, T)
    return xn * gamma[None, :, None] + beta[None, :, None]


def layer_norm(x: Array, gamma: Array, beta: Array, eps: float = 1e-5) -> Array:
    """nn.LayerNorm over the last dim (blocks.py:400-401)."""
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + x.dtype.type(eps)) * gamma + beta


def conv1d_same(x: Array, w: Array, b: Optional[Array], stride: int, causal: bool) -> Array:
    """``_Conv1d`` (blocks.py:34-53): the constructor's ``padding`` is ignored;
    (k-1) zeros are added all-left when causal, else floor((k-1)/2) on each side;
    then a stride-``stride`` valid convolution.  w: [C_out, C_in, k]."""
    B, C, L = x.shape
    O, Ci, k = w.shape
    assert Ci == C
    pad = k - 1
    if causal:
        xp = np.pad(x, ((0, 0), (0, 0), (pad, 0)))
    else:
        h = pad // 2
        xp = np.pad(x, ((0, 0), (0, 0), (h, h)))
    return _conv1d_valid(xp, w, b, stride)


def conv1d_zero_pad(x: Array, w: Array, b: Optional[Array], padding: int) -> Array:
    """plain nn.Conv1d(k, padding=p), stride 1 (Upsample1d factor==1, blocks.py:72-75)."""
    xp = np.pad(x, ((0, 0), (0, 0), (padding, padding)))
    return _conv1d_valid(xp, w, b, 1)


def _conv1d_valid(xp: Array, w: Array, b: Optional[Array], stride: int) -> Array:
    B, C, Lp = xp.shape
    O, _, k = w.shape
    Lo = (Lp - k) // stride + 1
    s0, s1, s2 = xp.strides
    cols = np.lib.stride_tricks.as_strided(xp, shape=(B, C, k, Lo), strides=(s0, s1, s2, s2 * stride), writeable=False)
    y = np.matmul(w.reshape(O, C * k), cols.reshape(B, C * k, Lo))
    if b is not None:
        y = y + b[None, :, None]
    return y.astype(xp.dtype, copy=False)


def conv_transpose1d(x: Array, w: Array, b: Optional[Array], stride: int, padding: int, output_padding: int) -> Array:
    """nn.ConvTranspose1d (Upsample1d, blocks.py:88-95).  w: [C_in, C_out, k].
    out[t*stride + j - padding] += x[t] . w[:, :, j]."""
    B, C, L = x.shape
    Ci, O, k = w.shape
    assert Ci == C
    Lo = (L - 1) * stride - 2 * padding + k + output_padding
    full = np.zeros((B, O, (L - 1) * stride + k + output_padding), dtype=x.dtype)
    for j in range(k):
        full[:, :, j: j + (L - 1) * stride + 1: stride] += np.matmul(w[:, :, j].T, x)
    y = full[:, :, padding: padding + Lo]
    if b is not None:
        y = y + b[None, :, None]
    return y


def crop_pair(x1: Array, x2: Array) -> Tuple[Array, Array]:
    """``crop`` (utils/module.py:186-204): centre-crop the longer tensor along T."""
    d = x1.shape[-1] - x2.shape[-1]
    if d == 0:
        return x1, x2
    start = d // 2
    end = d - start
    if d > 0:
        return x1[:, :, start: x1.shape[-1] - end], x2
    # python slice semantics of the reference for a negative diff
    return x1, x2[:, :, start: -end]


# ----------------------------------------------------------------------------
# the network, driven by a flat ``state_dict``-style parameter dictionary
# ----------------------------------------------------------------------------
class OracleUNetCFG1d:
    """Restatement of UNetCFG1d / UNet1d (jen1/model/model.py:13-376) and the
    blocks it is built from (jen1/model/blocks.py).  ``params`` uses the
    reference's ``state_dict`` keys."""

    def __init__(self, params: Dict[str, Array], *, channels: int, multipliers: Sequence[int],
                 factors: Sequence[int], num_blocks: Sequence[int], attentions: Sequence[int],
                 attention_heads: int, resnet_groups: int = 8, use_skip_scale: bool = True,
                 use_xattn_time: bool = True, dtype=np.float32, **_unused):
        self.dt = np.dtype(dtype)
        self.p = {k: np.asarray(v).astype(self.dt) for k, v in params.items()}
        self.channels = channels
        self.multipliers = list(multipliers)
        self.factors = list(factors)
        self.num_blocks = list(num_blocks)
        self.attentions = list(attentions)
        self.heads = attention_heads
        self.groups = resnet_groups
        self.skip_scale = self.dt.type(2 ** -0.5) if use_skip_scale else self.dt.type(1.0)
        self.use_xattn_time = use_xattn_time
        self.L = len(self.multipliers) - 1
        self.taps: Dict[str, Array] = {}     # optional intermediate taps for fixtures

    # -- leaves -------------------------------------------------------------
    def _time_features(self, prefix: str, t: Array) -> Array:
        """LearnedPositionalEmbedding + Linear (utils/module.py:58-79).
        ``t`` is the raw integer timestep promoted to float; the phase is
        ((t * w) * 2) * pi evaluated left to right in the working precision."""
        w = self.p[f"{prefix}.0.weights"]
        x = t.astype(self.dt)[:, None]
        freqs = x * w[None, :] * self.dt.type(2) * self.dt.type(math.pi)
        f = np.concatenate([x, np.sin(freqs), np.cos(freqs)], axis=-1).astype(self.dt)
        return linear(f, self.p[f"{prefix}.1.weight"], self.p[f"{prefix}.1.bias"])

    def mapping(self, t: Array) -> Array:
        """UNet1d.get_mapping (model.py:204-223, :75-89)."""
        m = gelu(self._time_features("to_time.0", t))
        m = gelu(linear(m, self.p["to_mapping.0.weight"], self.p["to_mapping.0.bias"]))
        m = gelu(linear(m, self.p["to_mapping.2.weight"], self.p["to_mapping.2.bias"]))
        return m

    def conv_block(self, n: str, x: Array, groups: int, scale_shift, causal: bool) -> Array:
        """ConvBlock1d.forward (blocks.py:137-145)."""
        h = group_norm(x, groups, self.p[f"{n}.groupnorm.weight"], self.p[f"{n}.groupnorm.bias"], 1e-5)
        if scale_shift is not None:
            sc, sh = scale_shift
            h = h * (sc + self.dt.type(1)) + sh
        h = silu(h)
        return conv1d_same(h, self.p[f"{n}.project.conv.weight"], self.p[f"{n}.project.conv.bias"], 1, causal)

    def resnet_block(self, n: str, x: Array, mapping: Array, groups: int, causal: bool) -> Array:
        """ResnetBlock1d.forward (blocks.py:219-231) + MappingToScaleShift (:161-165)."""
        h = self.conv_block(f"{n}.block1", x, groups, None, causal)
        ss = linear(silu(mapping), self.p[f"{n}.to_scale_shift.to_scale_shift.1.weight"],
                    self.p[f"{n}.to_scale_shift.to_scale_shift.1.bias"])[:, :, None]
        c = ss.shape[1] // 2
        h = self.conv_block(f"{n}.block2", h, groups, (ss[:, :c], ss[:, c:]), causal)
        if f"{n}.to_out.conv.weight" in self.p:
            res = conv1d_same(x, self.p[f"{n}.to_out.conv.weight"], self.p[f"{n}.to_out.conv.bias"], 1, causal)
        else:
            res = x
        return h + res

    def attention(self, n: str, x: Array, context: Optional[Array], context_mask: Optional[Array], causal: bool) -> Array:
        """Attention.forward + AttentionBase.forward, math path (blocks.py:415-437, 355-380).
        The padding mask multiplies K and V (it is NOT a -inf logit mask, :431-434)."""
        ctx = x if context is None else context
        xn = layer_norm(x, self.p[f"{n}.norm.weight"], self.p[f"{n}.norm.bias"])
        cn = layer_norm(ctx, self.p[f"{n}.norm_context.weight"], self.p[f"{n}.norm_context.bias"])
        q = linear(xn, self.p[f"{n}.to_q.weight"])
        kv = linear(cn, self.p[f"{n}.to_kv.weight"])
        mid = kv.shape[-1] // 2
        k, v = kv[..., :mid], kv[..., mid:]
        if context_mask is not None:
            m = context_mask.astype(self.dt)[:, :, None]
            k, v = k * m, v * m
        B, N, _ = q.shape
        M = k.shape[1]
        h = self.heads
        d = mid // h
        qh = q.reshape(B, N, h, d).transpose(0, 2, 1, 3)
        kh = k.reshape(B, M, h, d).transpose(0, 2, 1, 3)
        vh = v.reshape(B, M, h, d).transpose(0, 2, 1, 3)
        sim = np.matmul(qh, kh.transpose(0, 1, 3, 2)) * self.dt.type(d ** -0.5)
        if causal:
            # causal_mask (blocks.py:315-319): keep j <= i + (M - N)
            keep = ~np.triu(np.ones((N, M), dtype=bool), k=M - N + 1)
            sim = np.where(keep[None, None], sim, -np.finfo(self.dt).max)
        sim = sim - sim.max(axis=-1, keepdims=True)
        e = np.exp(sim)
        attn = e / e.sum(axis=-1, keepdims=True)
        out = np.matmul(attn, vh).transpose(0, 2, 1, 3).reshape(B, N, mid)
        return linear(out, self.p[f"{n}.attention.to_out.weight"], self.p[f"{n}.attention.to_out.bias"])

    def transformer1d(self, n: str, x: Array, layers: int, embedding: Array, embedding_mask, causal: bool) -> Array:
        """Transformer1d.forward (blocks.py:528-537): GN(32, eps=1e-6), the SAME 1x1 conv
        before and after the blocks; TransformerBlock.forward (:483-489)."""
        w, b = self.p[f"{n}.conv1d.conv.weight"], self.p[f"{n}.conv1d.conv.bias"]
        h = group_norm(x, 32, self.p[f"{n}.group_norm.weight"], self.p[f"{n}.group_norm.bias"], 1e-6)
        h = conv1d_same(h, w, b, 1, causal)
        h = h.transpose(0, 2, 1)
        for l in range(layers):
            bn = f"{n}.blocks.{l}"
            h = self.attention(f"{bn}.attention", h, None, None, causal) + h
            h = self.attention(f"{bn}.cross_attention", h, embedding, embedding_mask, False) + h
            f = gelu(linear(h, self.p[f"{bn}.feed_forward.0.weight"], self.p[f"{bn}.feed_forward.0.bias"]))
            h = linear(f, self.p[f"{bn}.feed_forward.2.weight"], self.p[f"{bn}.feed_forward.2.bias"]) + h
        h = h.transpose(0, 2, 1)
        return conv1d_same(np.ascontiguousarray(h), w, b, 1, causal)

    # -- UNet1d.forward -------------------------------------------------------
    def unet(self, x: Array, t: Array, embedding: Array, embedding_mask, ctx_channels: Optional[Array], causal: bool) -> Array:
        """UNet1d.forward (model.py:225-265)."""
        x = x.astype(self.dt)
        if ctx_channels is not None:
            x = np.concatenate([x, ctx_channels.astype(self.dt)], axis=1)
        mp = self.mapping(t)
        G = self.groups
        # to_in / to_out (Patcher / Unpatcher, blocks.py:256-259, 284-287) are never causal
        x = self.resnet_block("to_in.block", x, mp, 1, False)
        self.taps["to_in"] = x
        skips_list: List = [x]
        for i in range(self.L):
            n = f"downsamples.{i}"
            f = self.factors[i]
            x = conv1d_same(x, self.p[f"{n}.downsample.conv.weight"], self.p[f"{n}.downsample.conv.bias"], f, causal)
            skips = []
            for j in range(self.num_blocks[i]):
                x = self.resnet_block(f"{n}.blocks.{j}", x, mp, G, causal)
                skips.append(x)
            if self.attentions[i]:
                x = self.transformer1d(f"{n}.transformer", x, self.attentions[i], embedding, embedding_mask, causal)
                skips.append(x)
            skips_list.append(skips)
            self.taps[f"down{i}"] = x
        x = self.resnet_block("bottleneck.pre_block", x, mp, G, causal)
        if self.attentions[-1]:                                              # model.py:147
            x = self.transformer1d("bottleneck.transformer", x, self.attentions[-1], embedding, embedding_mask, causal)
        x = self.resnet_block("bottleneck.post_block", x, mp, G, causal)
        self.taps["bottleneck"] = x
        for idx, i in enumerate(reversed(range(self.L))):
            n = f"upsamples.{idx}"
            skips = skips_list.pop()
            nl = self.num_blocks[i] + (1 if self.attentions[i] else 0)
            for j in range(nl):
                xa, sk = crop_pair(x, skips.pop())                         # blocks.py:732-734
                x = np.concatenate([xa, sk * self.skip_scale], axis=1)
                x = self.resnet_block(f"{n}.blocks.{j}", x, mp, G, causal)
            if self.attentions[i]:
                x = self.transformer1d(f"{n}.transformer", x, self.attentions[i], embedding, embedding_mask, causal)
            f = self.factors[i]
            w, b = self.p[f"{n}.upsample.weight"], self.p[f"{n}.upsample.bias"]
            if f == 1:
                x = conv1d_zero_pad(x, w, b, 1)
            else:
                x = conv_transpose1d(x, w, b, f, f // 2 + f % 2, f % 2)
            self.taps[f"up{idx}"] = x
        x = x + skips_list.pop()                                             # model.py:261
        return self.resnet_block("to_out.block", x, mp, 1, False)

    # -- UNetCFG1d.forward ----------------------------------------------------
    def forward(self, x: Array, time: Array, *, embedding: Array, embedding_mask: Optional[Array] = None,
                embedding_scale: float = 1.0, embedding_mask_proba: float = 0.0, batch_cfg: bool = False,
                scale_cfg: bool = False, scale_phi: float = 0.7, channels_list=None, causal: bool = False,
                features=None, dropout_rows: Optional[Array] = None) -> Array:
        """UNetCFG1d.forward (model.py:299-376).  ``dropout_rows`` (bool[B]) injects the
        Bernoulli draw of ``rand_bool`` (utils/module.py:36-42) so that the CFG-dropout
        branch is reproducible; it is required whenever 0 < embedding_mask_proba < 1."""
        dt = self.dt
        B = embedding.shape[0]
        emb = embedding.astype(dt)
        mask = None if embedding_mask is None else embedding_mask
        if self.use_xattn_time:
            tok = gelu(self._time_features("to_time_embedding.0", time))
            emb = np.concatenate([emb, tok[:, None, :]], axis=1)
            if mask is not None:
                mask = np.concatenate([mask.astype(dt), np.ones((B, 1), dtype=dt)], axis=1)
        fixed = np.broadcast_to(self.p["fixed_embedding.embedding.weight"][None, : emb.shape[1]], emb.shape)
        if embedding_mask_proba > 0.0:
            if embedding_mask_proba >= 1.0:
                rows = np.ones((B,), dtype=bool)
            else:
                assert dropout_rows is not None, "inject the Bernoulli draw for 0 < proba < 1"
                rows = np.asarray(dropout_rows, dtype=bool)
            emb = np.where(rows[:, None, None], fixed, emb)
        ctx = None if not channels_list else channels_list[0]
        if embedding_scale != 1.0:
            if batch_cfg:
                out_all = self.unet(np.concatenate([x, x], 0), np.concatenate([time, time], 0),
                                    np.concatenate([emb, fixed], 0),
                                    None if mask is None else np.concatenate([mask, mask], 0),
                                    None if ctx is None else np.concatenate([ctx, ctx], 0), causal)
                out, out_masked = out_all[:B], out_all[B:]
            else:
                out = self.unet(x, time, emb, mask, ctx, causal)
                out_masked = self.unet(x, time, fixed, mask, ctx, causal)
            out_cfg = out_masked + (out - out_masked) * dt.type(embedding_scale)
            if scale_cfg:
                s0 = out.std(axis=1, ddof=1, keepdims=True)
                s1 = out_cfg.std(axis=1, ddof=1, keepdims=True)
                return dt.type(scale_phi) * (out_cfg * (s0 / s1)) + dt.type(1 - scale_phi) * out_cfg
            return out_cfg
        return self.unet(x, time, emb, mask, ctx, causal)

    __call__ = forward


# ----------------------------------------------------------------------------
# noise schedule + Gaussian diffusion
# ----------------------------------------------------------------------------
def linspace_f32(start: float, end: float, steps: int) -> Array:
    """torch.linspace for float32: step = (end-start)/(steps-1) in float32, the first
    half counts up from ``start`` and the second half counts down from ``end``
    (ATen RangeFactories).  Used by get_beta_schedule('linear')."""
    s, e = np.float32(start), np.float32(end)
    step = np.float32((e - s) / np.float32(steps - 1))
    idx = np.arange(steps)
    half = steps // 2
    up = (s + step * idx.astype(np.float32)).astype(np.float32)
    dn = (e - step * (steps - 1 - idx).astype(np.float32)).astype(np.float32)
    return np.where(idx < half, up, dn).astype(np.float32)


def get_beta_schedule(name: str, n: int) -> Array:
    """jen1/diffusion/gdm/noise_schedule.py:7-31 ('linear' and 'cosine')."""
    if name == "linear":
        scale = 1000 / n
        return linspace_f32(scale * 0.0001, scale * 0.02, n)
    if name == "cosine":
        ab = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
        return np.array([min(1 - ab((i + 1) / n) / ab(i / n), 0.999) for i in range(n)], dtype=np.float64).astype(np.float32)
    raise NotImplementedError(f"unknown beta schedule: {name}")


class OracleGaussianDiffusion:
    """GaussianDiffusion (jen1/diffusion/gdm/gdm.py:14-272).  All tables are float32
    like the reference's; the sampler injects every random draw so that runs are
    reproducible (the reference draws with torch's global RNG)."""

    def __init__(self, *, steps: int, betas: Array, objective: str = "noise", loss_type: str = "l2",
                 cfg_dropout_proba: float = 0.1, embedding_scale: float = 0.8, batch_cfg: bool = False,
                 scale_cfg: bool = False, sampling_timesteps: Optional[int] = None, ddim_sampling_eta: float = 1.0):
        assert objective in {"noise", "x0", "v"}
        assert loss_type in {"l1", "l2"}
        self.objective, self.loss_type = objective, loss_type
        self.cfg_dropout_proba, self.embedding_scale = cfg_dropout_proba, embedding_scale
        self.batch_cfg, self.scale_cfg = batch_cfg, scale_cfg
        self.num_timesteps = steps
        self.sampling_timesteps = steps if sampling_timesteps is None else sampling_timesteps
        assert self.sampling_timesteps <= steps
        self.is_ddim_sampling = self.sampling_timesteps < steps
        self.eta = ddim_sampling_eta
        f = np.float32
        betas = np.asarray(betas, dtype=f)
        assert betas.ndim == 1 and (betas > 0).all() and (betas <= 1).all()
        self.betas = betas
        alphas = (f(1) - betas).astype(f)
        # torch.cumprod on CPU accumulates float32 inputs in double (at::acc_type) and
        # rounds each prefix product to float32
        ac = np.cumprod(alphas.astype(np.float64)).astype(f)
        self.alphas_cumprod = ac
        self.alphas_cumprod_prev = np.concatenate([[f(1)], ac[:-1]]).astype(f)
        self.sqrt_alphas_cumprod = np.sqrt(ac)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(f(1) - ac)
        self.sqrt_recip_alphas_cumprod = np.sqrt(f(1) / ac)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(f(1) / ac - f(1))
        self.posterior_variance = betas * (f(1) - self.alphas_cumprod_prev) / (f(1) - ac)
        pv = self.posterior_variance
        self.posterior_log_variance_clipped = np.log(np.concatenate([pv[1:2], pv[1:]]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (f(1) - ac)
        self.posterior_mean_coef2 = (f(1) - self.alphas_cumprod_prev) * np.sqrt(alphas) / (f(1) - ac)

    @staticmethod
    def _ex(a: Array, t: Array, ndim: int) -> Array:
        """extract (utils/script_util.py:43-46)."""
        return a[t].reshape((-1,) + (1,) * (ndim - 1))

    def ddim_times(self) -> List[Tuple[int, int]]:
        """gdm.py:190-193: linspace(-1, N-1, S+1) truncated to int, reversed, paired."""
        ts = linspace_f32(-1.0, self.num_timesteps - 1, self.sampling_timesteps + 1)
        times = list(reversed([int(v) for v in ts]))          # .int() truncates toward zero
        return list(zip(times[:-1], times[1:]))

    def ddim_coeffs(self, time: int, time_next: int) -> Tuple[np.float32, np.float32, np.float32]:
        """gdm.py:212-216 -> (sqrt(alpha_next), c, sigma) in float32."""
        f = np.float32
        a, an = self.alphas_cumprod[time], self.alphas_cumprod[time_next]
        sigma = f(self.eta) * np.sqrt((f(1) - a / an) * (f(1) - an) / (f(1) - a))
        c = np.sqrt(f(1) - an - sigma ** 2)
        return f(np.sqrt(an)), f(c), f(sigma)

    def _model_call(self, model, x, t, conditioning, causal, dropout_rows):
        return model(x, t, embedding=conditioning["cross_attn_cond"],
                     embedding_mask=conditioning["cross_attn_masks"],
                     embedding_scale=self.embedding_scale, embedding_mask_proba=self.cfg_dropout_proba,
                     features=conditioning.get("global_cond"),
                     channels_list=[conditioning["input_concat_cond"]],
                     batch_cfg=self.batch_cfg, scale_cfg=self.scale_cfg, causal=causal,
                     dropout_rows=dropout_rows)

    def model_predictions(self, x, t, model, conditioning, clip_x_start=False, causal=False, dropout_rows=None):
        """gdm.py:116-142."""
        out = self._model_call(model, x, t, conditioning, causal, dropout_rows)
        nd = x.ndim
        clip = (lambda v: np.clip(v, -1.0, 1.0)) if clip_x_start else (lambda v: v)
        if self.objective == "noise":
            eps = out
            x0 = clip(self._ex(self.sqrt_recip_alphas_cumprod, t, nd) * x - self._ex(self.sqrt_recipm1_alphas_cumprod, t, nd) * eps)
        elif self.objective == "x0":
            x0 = clip(out)
            eps = (self._ex(self.sqrt_recip_alphas_cumprod, t, nd) * x - x0) / self._ex(self.sqrt_recipm1_alphas_cumprod, t, nd)
        else:
            x0 = clip(self._ex(self.sqrt_alphas_cumprod, t, nd) * x - self._ex(self.sqrt_one_minus_alphas_cumprod, t, nd) * out)
            eps = (self._ex(self.sqrt_recip_alphas_cumprod, t, nd) * x - x0) / self._ex(self.sqrt_recipm1_alphas_cumprod, t, nd)
        return eps.astype(x.dtype), x0.astype(x.dtype)

    def ddim_sample(self, model, shape, conditioning, *, init_noise: Array, step_noises: Sequence[Array],
                    dropout_rows: Optional[Sequence[Array]] = None, causal=False, init_data=None,
                    return_all_timesteps=False):
        """gdm.py:181-225 with every torch RNG draw (``randn(shape)``, ``randn_like``,
        the CFG-dropout Bernoulli) supplied by the caller."""
        audio = np.asarray(init_noise, dtype=np.float32).reshape(shape)
        if init_data is not None:
            audio = audio + init_data
        audios = [audio]
        B = shape[0]
        for i, (time, time_next) in enumerate(self.ddim_times()):
            t = np.full((B,), time, dtype=np.int64)
            eps, x0 = self.model_predictions(audio, t, model, conditioning, clip_x_start=True, causal=causal,
                                             dropout_rows=None if dropout_rows is None else dropout_rows[i])
            audios.append(audio)
            if time_next < 0:
                audio = x0
                continue
            sa, c, sigma = self.ddim_coeffs(time, time_next)
            audio = (x0 * sa + c * eps + sigma * np.asarray(step_noises[i], dtype=np.float32)).astype(np.float32)
        return audio if not return_all_timesteps else np.stack(audios, axis=1)

    def p_sample_loop(self, model, shape, conditioning, *, init_noise: Array, step_noises: Sequence[Array],
                      dropout_rows=None, init_data=None):
        """DDPM ancestral loop (gdm.py:144-179).  NOTE the reference draws the per-step
        noise with ``torch.rand_like`` (uniform, :161) and does not forward ``causal``."""
        audio = np.asarray(init_noise, dtype=np.float32).reshape(shape)
        if init_data is not None:
            audio = audio + init_data
        B = shape[0]
        for i, t_int in enumerate(reversed(range(self.num_timesteps))):
            t = np.full((B,), t_int, dtype=np.int64)
            _, x0 = self.model_predictions(audio, t, model, conditioning, clip_x_start=False, causal=False,
                                           dropout_rows=None if dropout_rows is None else dropout_rows[i])
            x0 = np.clip(x0, -1.0, 1.0)
            nd = audio.ndim
            mean = self._ex(self.posterior_mean_coef1, t, nd) * x0 + self._ex(self.posterior_mean_coef2, t, nd) * audio
            logvar = self._ex(self.posterior_log_variance_clipped, t, nd)
            noise = np.asarray(step_noises[i], dtype=np.float32) if t_int > 0 else 0.0
            audio = (mean + np.exp(0.5 * logvar) * noise).astype(np.float32)
        return audio

    def q_sample(self, x_start, t, noise):
        """gdm.py:232-243."""
        nd = x_start.ndim
        return self._ex(self.sqrt_alphas_cumprod, t, nd) * x_start + self._ex(self.sqrt_one_minus_alphas_cumprod, t, nd) * noise

    def training_losses(self, model, x_start, t, conditioning, noise, causal=False, dropout_rows=None):
        """``training_loosses`` (gdm.py:245-272): mean over (C,T) then over B."""
        x_t = self.q_sample(x_start, t, noise).astype(np.float32)
        out = self._model_call(model, x_t, t, conditioning, causal, dropout_rows)
        nd = x_start.ndim
        if self.objective == "noise":
            target = noise
        elif self.objective == "x0":
            target = x_start
        else:
            target = self._ex(self.sqrt_alphas_cumprod, t, nd) * noise - self._ex(self.sqrt_one_minus_alphas_cumprod, t, nd) * x_start
        d = out - target
        per = (d * d) if self.loss_type == "l2" else np.abs(d)
        return per.reshape(per.shape[0], -1).mean(axis=1).mean()


class OracleVDM:
    """``VDM`` (/root/reference/jen1/diffusion/vdm/vdm.py:12-110) with the repairs of SURVEY.md Appendix A-3 / A-4 that
    jen1_amd/vdm.py documents (the shipped class cannot run): the model gets a batch vector of the step's continuous time,
    alphas / sigmas are read by step index, the loss broadcasts them as [B, 1, 1].  Pinned by tests/golden/tiny_vdm.npz, which
    make_golden.py produces from a subclass of the reference's own class with the same three repairs."""

    def __init__(self, *, loss_type: str = "l2", cfg_dropout_proba: float = 0.1, embedding_scale: float = 0.8,
                 batch_cfg: bool = False, scale_cfg: bool = False):
        self.loss_type, self.cfg_dropout_proba, self.embedding_scale = loss_type, cfg_dropout_proba, embedding_scale
        self.batch_cfg, self.scale_cfg = batch_cfg, scale_cfg

    _model_call = OracleGaussianDiffusion._model_call

    def sample(self, model, shape, conditioning, *, step: int, init_noise: Array, causal=False, init_data=None,
               return_all_timesteps=False, dropout_rows=None):
        """vdm.py:42-79 (repaired): x_pred = alpha x - sigma v; noise_pred = sigma x + alpha v; x = alpha' x_pred + sigma' noise_pred"""
        f = np.float32
        audio = np.asarray(init_noise, dtype=f).reshape(shape)
        if init_data is not None:
            audio = audio + init_data
        steps = linspace_f32(1.0, 0.0, step + 1)
        al = np.cos(steps * f(math.pi / 2)).astype(f)
        sg = np.sin(steps * f(math.pi / 2)).astype(f)
        audios = [audio]
        for i in range(step):
            t = np.full((shape[0],), steps[i], dtype=f)
            v = self._model_call(model, audio, t, conditioning, causal, None if dropout_rows is None else dropout_rows[i])
            x_pred = al[i] * audio - sg[i] * v
            noise_pred = sg[i] * audio + al[i] * v
            audio = (al[i + 1] * x_pred + sg[i + 1] * noise_pred).astype(f)
            audios.append(audio)
        return audio if not return_all_timesteps else np.stack(audios, axis=1)

    def training_losses(self, model, x_start, conditioning, noise, times, causal=False, dropout_rows=None):
        """vdm.py:81-110 (repaired); the target uses x_t as written (vdm.py:106)"""
        f = np.float32
        times = np.asarray(times, dtype=f)
        al = np.cos(times * f(math.pi / 2)).astype(f).reshape(-1, 1, 1)
        sg = np.sin(times * f(math.pi / 2)).astype(f).reshape(-1, 1, 1)
        x_t = (x_start * al + noise * sg).astype(f)
        out = self._model_call(model, x_t, times, conditioning, causal, dropout_rows)
        d = out - (noise * al - x_t * sg)
        per = (d * d) if self.loss_type == "l2" else np.abs(d)
        return per.reshape(per.shape[0], -1).mean(axis=1).mean()


# ----------------------------------------------------------------------------
# host-side helpers restated from files that cannot be imported here
# (trainer.py / generation.py need encodec + tensorboard)
# ----------------------------------------------------------------------------
def task_mask(task: str, T: int, *, mask_length: Optional[int] = None, mask_start: Optional[int] = None) -> Tuple[Array, Optional[bool]]:
    """UnifiedMultiTaskTrainer.random_mask with the ``random`` draws injected
    (trainer.py:215-247).  Returns (mask[1,1,T] with 1 = keep, causal or None when
    the reference flips a coin)."""
    task = task.lower()
    m = np.ones((1, 1, T), dtype=np.float32)
    if task == "text_guided":
        return np.zeros((1, 1, T), dtype=np.float32), None
    if task == "music_inpaint":
        m[:, :, mask_start: mask_start + mask_length] = 0
        return m, False
    if task == "music_cont":
        m[:, :, T - mask_length:] = 0
        return m, True
    raise ValueError(task)


def input_concat_cond(masked_input: Array, mask: Array) -> Array:
    """cat([x*mask, mask], dim=1) -> [b, 129, T] (trainer.py:271; generation.py:171-185)."""
    return np.concatenate([masked_input, np.broadcast_to(mask, (masked_input.shape[0], 1, masked_input.shape[2]))], axis=1)


# ---------------------------------------------------------------------------------------------------
# optimiser step: trainer.py:144-149, train.py:56-60, :84
def clip_grad_norm(grads, max_norm: float):
    """nn.utils.clip_grad_norm_ (L2, error_if_nonfinite=False): returns (clipped grads, total_norm)."""
    total = float(np.sqrt(sum(float(np.sum(g.astype(np.float64) ** 2)) for g in grads)))
    coef = min(1.0, max_norm / (total + 1e-6))
    return [(g * np.float32(coef)).astype(np.float32) for g in grads], total


def adamw_step(p, g, m, v, step: int, lr=3e-5, betas=(0.9, 0.95), eps=1e-8, weight_decay=0.1):
    """one torch.optim.AdamW step (decoupled weight decay, single-tensor formulation), float32 like the reference;
    returns (p, m, v).  ``step`` is 1-based."""
    f = np.float32
    b1, b2 = f(betas[0]), f(betas[1])
    p = (p * f(1.0 - lr * weight_decay)).astype(f)
    m = (b1 * m + (f(1) - b1) * g).astype(f)
    v = (b2 * v + (f(1) - b2) * g * g).astype(f)
    bc1 = 1.0 - betas[0] ** step
    bc2 = 1.0 - betas[1] ** step
    denom = (np.sqrt(v) / f(np.sqrt(bc2)) + f(eps)).astype(f)
    p = (p - f(lr / bc1) * (m / denom)).astype(f)
    return p, m, v


def linear_lr_factor(it: int, start_factor: float = 1.0 / 3, end_factor: float = 1.0, total_iters: int = 5) -> float:
    """torch.optim.lr_scheduler.LinearLR closed form (train.py:84): factor after ``it`` scheduler steps."""
    t = min(it, total_iters)
    return start_factor + (end_factor - start_factor) * t / total_iters
