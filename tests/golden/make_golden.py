#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference, read-only).  The
reference is imported unmodified on CPU with two shim modules for third-party
imports that are absent offline (``einops_exts.rearrange_many`` -- a pure
reshape helper -- and ``dac.nn.layers.Snake1d`` -- only constructed when
use_snake=True, never on this path), exactly as SURVEY.md section 8c / Appendix D
describe.  Fixtures hold DATA only: outputs of the reference (sub-sampled where
large) plus the scalar settings of each case.  Inputs and weights are NOT
stored; they are regenerated from ``jen1_amd.init_fill`` / ``jen1_amd.synth``
by the tests, here and on the GPU box.

    python tests/golden/make_golden.py            # all fixtures
    python tests/golden/make_golden.py units tiny # a subset
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "jen-1-pytorch_amd"))

import einops  # noqa: E402
import numpy as np  # noqa: E402
import torch  # noqa: E402

# ---- shims for absent third-party imports (see module docstring) -------------
_m = types.ModuleType("einops_exts")
_m.rearrange_many = lambda ts, pat, **kw: tuple(einops.rearrange(t, pat, **kw) for t in ts)
sys.modules["einops_exts"] = _m
_dac, _dacnn, _dacl = types.ModuleType("dac"), types.ModuleType("dac.nn"), types.ModuleType("dac.nn.layers")


class _Snake1d(torch.nn.Module):
    def __init__(self, *a, **k):
        raise RuntimeError("Snake1d is not on the JEN-1 path")


_dacl.Snake1d = _Snake1d
sys.modules.update({"dac": _dac, "dac.nn": _dacnn, "dac.nn.layers": _dacl})
sys.path.insert(0, "/root/reference")

from jen1.model import blocks as rb  # noqa: E402
from jen1.model.model import UNetCFG1d  # noqa: E402
from jen1.diffusion.gdm.gdm import GaussianDiffusion  # noqa: E402
from jen1.diffusion.gdm.noise_schedule import get_beta_schedule  # noqa: E402
import utils.module as rmod  # noqa: E402

from jen1_amd import synth  # noqa: E402
from jen1_amd.config import UNetSpec, full_model_config, tiny_model_config  # noqa: E402
from jen1_amd.init_fill import fill, fill_normal, fill_uniform  # noqa: E402

torch.set_grad_enabled(False)
torch.manual_seed(0)
SEED = 1234


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def load_filled(module: torch.nn.Module, prefix: str = "", seed: int = SEED):
    sd = module.state_dict()
    new = {k: T(fill(prefix + k, tuple(v.shape), seed)) for k, v in sd.items()}
    module.load_state_dict(new)
    return module.eval()


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"  wrote {name}.npz ({os.path.getsize(path) / 1024:.0f} KiB)")


# ------------------------------------------------------------------------------
def gen_schedule():
    out = {}
    for name in ("linear", "cosine"):
        betas, _ = get_beta_schedule(name, 1000)
        betas = betas.to(torch.float32)
        gd = GaussianDiffusion(steps=1000, betas=betas, objective="noise", loss_type="l2", device="cpu",
                               sampling_timesteps=100)
        for attr in ("betas", "alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
                     "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "posterior_variance",
                     "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"):
            out[f"{name}.{attr}"] = getattr(gd, attr).numpy()
    betas, _ = get_beta_schedule("linear", 1000)
    for S in (10, 100):
        gd = GaussianDiffusion(steps=1000, betas=betas.float(), objective="noise", loss_type="l2", device="cpu",
                               sampling_timesteps=S)
        times = torch.linspace(-1, 999, steps=S + 1)
        times = list(reversed(times.int().tolist()))
        out[f"ddim_times.{S}"] = np.array(times, dtype=np.int64)
        co = []
        for t, tn in zip(times[:-1], times[1:]):
            if tn < 0:
                continue
            a, an = gd.alphas_cumprod[t], gd.alphas_cumprod[tn]
            sigma = 1.0 * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
            c = (1 - an - sigma ** 2).sqrt()
            co.append([an.sqrt().item(), c.item(), sigma.item()])
        out[f"ddim_coeffs.{S}"] = np.array(co, dtype=np.float32)
    save("schedule", **out)


# ------------------------------------------------------------------------------
def gen_units():
    out = {}
    meta = {}
    B, L = 2, 37
    # a1: _Conv1d (blocks.py:34-53)
    for (ci, co, k, s) in ((8, 12, 1, 1), (8, 12, 3, 1), (8, 12, 5, 2), (8, 12, 9, 4)):
        m = load_filled(rb.Conv1d(in_channels=ci, out_channels=co, kernel_size=k, stride=s), f"u.conv.k{k}s{s}.")
        x = T(fill_normal(f"u.conv.x.{ci}", (B, ci, L)))
        for causal in (False, True):
            out[f"conv.k{k}s{s}.c{int(causal)}"] = m(x, causal).numpy()
    # a2: Upsample1d (blocks.py:69-95)
    for f in (1, 2, 4):
        m = load_filled(rb.Upsample1d(in_channels=8, out_channels=12, factor=f), f"u.up.f{f}.upsample.")
        x = T(fill_normal("u.up.x", (B, 8, 11)))
        out[f"upsample.f{f}"] = m(x).numpy()
    # a5: ResnetBlock1d
    for (ci, co, g) in ((16, 16, 8), (24, 16, 8), (17, 16, 1)):
        m = load_filled(rb.ResnetBlock1d(in_channels=ci, out_channels=co, num_groups=g, context_mapping_features=32),
                        f"u.res.{ci}.{co}.{g}.")
        x = T(fill_normal(f"u.res.x.{ci}", (B, ci, L)))
        mp = T(fill_normal("u.res.map", (B, 32)))
        for causal in (False, True):
            out[f"res.{ci}.{co}.{g}.c{int(causal)}"] = m(x, mp, causal).numpy()
    # a7: Attention
    feat, heads, hf = 32, 4, 8
    att = load_filled(rb.Attention(features=feat, head_features=hf, num_heads=heads), "u.att.self.")
    x = T(fill_normal("u.att.x", (B, 7, feat)))
    for causal in (False, True):
        out[f"att.self.c{int(causal)}"] = att(x, causal=causal).numpy()
    xatt = load_filled(rb.Attention(features=feat, head_features=hf, num_heads=heads, context_features=48), "u.att.cross.")
    ctx = T(fill_normal("u.att.ctx", (B, 9, 48)))
    cm = torch.tensor([[1, 1, 1, 1, 0, 0, 0, 0, 1], [1, 1, 1, 1, 1, 1, 1, 0, 1]], dtype=torch.float32)
    out["att.cross.masked"] = xatt(x, context=ctx, context_mask=cm).numpy()
    out["att.cross.nomask"] = xatt(x, context=ctx).numpy()
    # a9: Transformer1d (shared 1x1 conv, GN(32))
    tr = load_filled(rb.Transformer1d(num_layers=1, channels=64, num_heads=4, head_features=16, multiplier=1,
                                      context_features=48), "u.tr.")
    xt = T(fill_normal("u.tr.x", (B, 64, 7)))
    for causal in (False, True):
        out[f"tr.c{int(causal)}"] = tr(xt, context=ctx, context_mask=cm, causal=causal).numpy()
    # a10: Down / Up / Bottleneck blocks with odd lengths -> crop
    kw = dict(num_groups=8, context_mapping_features=32, context_embedding_features=48,
              attention_heads=4, attention_multiplier=1)
    dn = load_filled(rb.DownsampleBlock1d(in_channels=16, out_channels=32, factor=2, num_layers=2, use_skip=True,
                                          num_transformer_blocks=1, **kw), "u.down.")
    xd = T(fill_normal("u.down.x", (B, 16, L)))
    for causal in (False, True):
        y, skips = dn(xd, mapping=mp, embedding=ctx, embedding_mask=cm, causal=causal)
        out[f"down.c{int(causal)}.y"] = y.numpy()
        for i, s in enumerate(skips):
            out[f"down.c{int(causal)}.skip{i}"] = s.numpy()
    up = load_filled(rb.UpsampleBlock1d(in_channels=32, out_channels=16, factor=2, num_layers=3, use_skip=True,
                                        skip_channels=32, use_skip_scale=True, num_transformer_blocks=1, **kw), "u.up.")
    xu = T(fill_normal("u.upb.x", (B, 32, 20)))          # longer than the skips (19) -> crop
    sk = [T(fill_normal(f"u.upb.skip{i}", (B, 32, 19))) for i in range(3)]
    for causal in (False, True):
        out[f"upblock.c{int(causal)}"] = up(xu, skips=list(sk), mapping=mp, embedding=ctx, embedding_mask=cm, causal=causal).numpy()
    bt = load_filled(rb.BottleneckBlock1d(channels=32, num_transformer_blocks=1, **kw), "u.bott.")
    xb = T(fill_normal("u.bott.x", (B, 32, 5)))
    out["bottleneck.c0"] = bt(xb, mapping=mp, embedding=ctx, embedding_mask=cm, causal=False).numpy()
    # a11: time features (fp32 sin/cos of large arguments)
    te = load_filled(rmod.TimePositionalEmbedding(dim=64, out_features=40), "u.time.")
    tt = torch.tensor([0, 1, 9, 499, 989, 999], dtype=torch.long)
    out["time.features"] = te(tt).numpy()
    save("units", **out)


# ------------------------------------------------------------------------------
def _build(cfg):
    model = UNetCFG1d(**cfg)
    spec = UNetSpec(**cfg)
    sd = model.state_dict()
    got = [(k, tuple(v.shape)) for k, v in sd.items()]
    assert got == spec.param_shapes(), "UNetSpec.param_shapes() disagrees with the reference state_dict"
    load_filled(model)
    return model, spec


def _inputs(B, T_, task="text_guided"):
    cond = synth.conditioning(B, T_, task)
    x = synth.latents(B, T_)
    return x, cond


def gen_tiny():
    cfg = tiny_model_config()
    model, spec = _build(cfg)
    B, T_ = 2, 300
    x, cond = _inputs(B, T_)
    xi, cond_i = _inputs(B, T_, "music_inpaint")
    emb, mask = T(cond["cross_attn_cond"]), T(cond["cross_attn_masks"])
    t = torch.tensor([999, 499], dtype=torch.long)
    out = {"schema": np.array(json.dumps([[k, list(s)] for k, s in spec.param_shapes()]))}
    cases = []
    for scale in (1.0, 0.8):
        for batch_cfg in (True, False):
            for scale_cfg in (True, False):
                for causal in (False, True):
                    if scale == 1.0 and (not batch_cfg or scale_cfg):
                        continue       # flags are ignored when embedding_scale == 1.0 (model.py:375-376)
                    cases.append((scale, batch_cfg, scale_cfg, causal))
    for (scale, batch_cfg, scale_cfg, causal) in cases:
        y = model(T(x), t, embedding=emb, embedding_mask=mask, embedding_scale=scale, embedding_mask_proba=0.0,
                  batch_cfg=batch_cfg, scale_cfg=scale_cfg, channels_list=[T(cond["input_concat_cond"])],
                  features=None, causal=causal)
        out[f"y.s{scale}.b{int(batch_cfg)}.r{int(scale_cfg)}.c{int(causal)}"] = y.numpy()[:, :, ::3]
    # inpaint-style context channels (non-zero masked input + mask channel), full output kept
    y = model(T(xi), t, embedding=emb, embedding_mask=mask, embedding_scale=0.8, embedding_mask_proba=0.0,
              batch_cfg=True, scale_cfg=True, channels_list=[T(cond_i["input_concat_cond"])], features=None, causal=False)
    out["y.inpaint"] = y.numpy()
    # injected CFG-dropout rows (model.py:323-328): row 1 swapped to the fixed embedding
    real = rmod.rand_bool
    import jen1.model.model as rmodel
    rmodel.rand_bool = lambda shape, proba, device=None: torch.tensor([False, True]).reshape(shape)
    y = model(T(x), t, embedding=emb, embedding_mask=mask, embedding_scale=0.8, embedding_mask_proba=0.2,
              batch_cfg=True, scale_cfg=True, channels_list=[T(cond["input_concat_cond"])], features=None, causal=False)
    rmodel.rand_bool = real
    out["y.dropout_row1"] = y.numpy()[:, :, ::3]
    # no embedding mask at all
    y = model(T(x), t, embedding=emb, embedding_mask=None, embedding_scale=0.8, batch_cfg=True, scale_cfg=False,
              channels_list=[T(cond["input_concat_cond"])], features=None, causal=False)
    out["y.nomask"] = y.numpy()[:, :, ::3]
    save("tiny_unet", **out)
    return model


def gen_tiny_sampler(model=None):
    cfg = tiny_model_config()
    if model is None:
        model, _ = _build(cfg)
    B, T_, S = 2, 300, 10
    _, cond = _inputs(B, T_)
    cond_t = {k: (None if v is None else T(v)) for k, v in cond.items()}
    betas, _ = get_beta_schedule("linear", 1000)
    shape = (B, 128, T_)
    init = synth.noise_list(1, shape, seed=7)[0]
    noises = synth.noise_list(S, shape, seed=11)
    out = {}

    def run(proba, scale, batch_cfg, scale_cfg, causal, drops=None, objective="noise"):
        gd = GaussianDiffusion(steps=1000, betas=betas.float(), objective=objective, loss_type="l2", device="cpu",
                               cfg_dropout_proba=proba, embedding_scale=scale, batch_cfg=batch_cfg,
                               scale_cfg=scale_cfg, sampling_timesteps=S)
        seq = [T(init)] + [T(n) for n in noises]
        it = iter(seq)
        r_randn, r_randn_like, r_bern = torch.randn, torch.randn_like, torch.bernoulli
        torch.randn = lambda *a, **k: next(it).clone()
        torch.randn_like = lambda *a, **k: next(it).clone()
        if drops is not None:
            dit = iter(drops)
            torch.bernoulli = lambda p: T(np.asarray(next(dit), dtype=np.float32)).reshape(p.shape)
        try:
            y = gd.sample(model, shape, cond_t, causal=causal)
        finally:
            torch.randn, torch.randn_like, torch.bernoulli = r_randn, r_randn_like, r_bern
        return y.numpy()

    out["ddim10.cfg"] = run(0.0, 0.8, True, True, False)
    out["ddim10.nocfg.causal"] = run(0.0, 1.0, False, False, True)[:, :, ::3]
    drops = [[(i + b) % 3 == 0 for b in range(B)] for i in range(S)]
    out["ddim10.dropout"] = run(0.2, 0.8, True, True, False, drops=drops)[:, :, ::3]
    out["ddim10.dropout.rows"] = np.array(drops, dtype=bool)
    out["ddim10.x0"] = run(0.0, 0.8, True, True, False, objective="x0")[:, :, ::3]
    out["ddim10.v"] = run(0.0, 0.8, True, True, False, objective="v")[:, :, ::3]
    save("tiny_sampler", **out)


def gen_tiny_train(model=None):
    cfg = tiny_model_config()
    if model is None:
        model, _ = _build(cfg)
    B, T_ = 2, 300
    out = {}
    betas, _ = get_beta_schedule("linear", 1000)
    names = ["to_in.block.block1.project.conv.weight", "downsamples.1.transformer.blocks.0.cross_attention.to_kv.weight",
             "bottleneck.transformer.conv1d.conv.weight", "upsamples.0.upsample.weight",
             "to_out.block.block2.groupnorm.weight", "to_time.0.0.weights", "fixed_embedding.embedding.weight",
             "to_mapping.0.bias"]
    out["grad_names"] = np.array(json.dumps(names))
    out["grad_names_all"] = np.array(json.dumps([n for n, _ in model.named_parameters()]))
    for task, causal in (("text_guided", False), ("music_inpaint", False), ("music_cont", True)):
        x0 = synth.latents(B, T_, key="clip")
        cond = synth.conditioning(B, T_, task)
        cond_t = {k: (None if v is None else T(v)) for k, v in cond.items()}
        noise = fill_uniform(f"synth.trainnoise.{task}", (B, 128, T_), 3, 0.0, 1.0)
        t = torch.tensor([17, 801], dtype=torch.long)
        for objective in ("noise", "x0", "v"):
            gd = GaussianDiffusion(steps=1000, betas=betas.float(), objective=objective, loss_type="l2", device="cpu",
                                   cfg_dropout_proba=0.0, embedding_scale=0.8, batch_cfg=True, scale_cfg=True)
            with torch.enable_grad():
                model.zero_grad(set_to_none=True)
                model.train()
                loss = gd.training_loosses(model, T(x0), t, cond_t, noise=T(noise), causal=causal)
                loss.backward()
            out[f"loss.{task}.{objective}"] = np.float32(loss.item())
            params = dict(model.named_parameters())
            out[f"gradnorm.{task}.{objective}"] = np.array([params[n].grad.norm().item() for n in names], dtype=np.float32)
            if objective == "noise":
                # every parameter's gradient norm + a strided sample of its entries, in named_parameters() order
                out[f"gradnorm_all.{task}"] = np.array([p_.grad.norm().item() for _, p_ in model.named_parameters()], dtype=np.float32)
                out[f"gradsample_all.{task}"] = np.concatenate(
                    [p_.grad.reshape(-1)[:: max(1, p_.numel() // 16)][:16].numpy() for _, p_ in model.named_parameters()]).astype(np.float32)
            if task == "text_guided" and objective == "noise":
                out["grad.to_time.0.0.weights"] = params["to_time.0.0.weights"].grad.numpy().copy()
                out["grad.to_out.block.block2.project.conv.bias"] = params["to_out.block.block2.project.conv.bias"].grad.numpy().copy()
    model.eval()
    save("tiny_train", **out)


def gen_full():
    cfg = full_model_config()
    model, spec = _build(cfg)
    assert spec.num_params() == 296_543_106, spec.num_params()
    B, T_ = 2, 1500
    x, cond = _inputs(B, T_)
    t = torch.tensor([999, 9], dtype=torch.long)
    taps = {}

    def hook(name):
        def f(_m, _i, o):
            o = o[0] if isinstance(o, tuple) else o
            taps[name] = np.array([o.norm().item(), o.abs().max().item(), o.shape[-1]], dtype=np.float64)
        return f
    model.to_in.register_forward_hook(hook("to_in"))
    for i, d in enumerate(model.downsamples):
        d.register_forward_hook(hook(f"down{i}"))
    model.bottleneck.register_forward_hook(hook("bottleneck"))
    for i, u in enumerate(model.upsamples):
        u.register_forward_hook(hook(f"up{i}"))
    out = {}
    y = model(T(x), t, embedding=T(cond["cross_attn_cond"]), embedding_mask=T(cond["cross_attn_masks"]),
              embedding_scale=0.8, embedding_mask_proba=0.0, batch_cfg=True, scale_cfg=True,
              channels_list=[T(cond["input_concat_cond"])], features=None, causal=False)
    out["y.cfg"] = y.numpy()[:, :, ::16]
    out["y.cfg.norm"] = np.array([y.norm().item(), y.abs().max().item()])
    for k, v in taps.items():
        out[f"tap.cfg.{k}"] = v
    taps.clear()
    y = model(T(x), t, embedding=T(cond["cross_attn_cond"]), embedding_mask=T(cond["cross_attn_masks"]),
              embedding_scale=1.0, channels_list=[T(cond["input_concat_cond"])], features=None, causal=True)
    out["y.nocfg.causal"] = y.numpy()[:, :, ::16]
    for k, v in taps.items():
        out[f"tap.nocfg.{k}"] = v
    save("full_unet", **out)


def _hook_taps(model, taps):
    def hook(name):
        def f(_m, _i, o):
            o = o[0] if isinstance(o, tuple) else o
            taps[name] = np.array([o.norm().item(), o.abs().max().item(), o.shape[-1]], dtype=np.float64)
        return f
    hs = [model.to_in.register_forward_hook(hook("to_in"))]
    for i, d in enumerate(model.downsamples):
        hs.append(d.register_forward_hook(hook(f"down{i}")))
    hs.append(model.bottleneck.register_forward_hook(hook("bottleneck")))
    for i, u in enumerate(model.upsamples):
        hs.append(u.register_forward_hook(hook(f"up{i}")))
    return hs


def gen_full_bench():
    """The exact shapes bench.py times (BASELINE configs[1], [2], [4]) through the reference: full model at B=8, T=1500
    without CFG and with the CFG pair (2B = 16), B=1, T=9000 for the continuation (causal) and inpaint tasks, a 10-step
    DDIM at the full configuration (B=2) and a 2-step DDIM at B=8 with and without the CFG pair.  Outputs are sub-sampled
    along T; per-level norms are kept for the forwards."""
    cfg = full_model_config()
    model, spec = _build(cfg)
    out = {}
    taps = {}
    _hook_taps(model, taps)
    # ---- configs[1] / [2]: B = 8, T = 1500 --------------------------------------------------------------------
    B, T_ = 8, 1500
    x, cond = _inputs(B, T_)
    t = torch.tensor([999, 989, 499, 259, 129, 59, 9, 0], dtype=torch.long)
    out["B8.t"] = t.numpy()
    kw = dict(embedding=T(cond["cross_attn_cond"]), embedding_mask=T(cond["cross_attn_masks"]), embedding_mask_proba=0.0,
              channels_list=[T(cond["input_concat_cond"])], features=None)
    y = model(T(x), t, embedding_scale=1.0, causal=False, **kw)
    out["B8.y.nocfg"] = y.numpy()[:, :, ::16]
    for k, v in taps.items():
        out[f"B8.tap.nocfg.{k}"] = v
    taps.clear()
    y = model(T(x), t, embedding_scale=0.8, batch_cfg=True, scale_cfg=True, causal=False, **kw)
    out["B8.y.cfg"] = y.numpy()[:, :, ::16]
    for k, v in taps.items():
        out[f"B8.tap.cfg.{k}"] = v
    taps.clear()
    # ---- configs[4]: B = 1, T = 9000, continuation (causal) and inpaint (non-causal) -------------------------------
    B, T_ = 1, 9000
    t1 = torch.tensor([499], dtype=torch.long)
    for task, causal in (("music_cont", True), ("music_inpaint", False)):
        x, cond = _inputs(B, T_, task)
        kw = dict(embedding=T(cond["cross_attn_cond"]), embedding_mask=T(cond["cross_attn_masks"]), embedding_mask_proba=0.0,
                  channels_list=[T(cond["input_concat_cond"])], features=None)
        y = model(T(x), t1, embedding_scale=0.8, batch_cfg=True, scale_cfg=True, causal=causal, **kw)
        out[f"T9000.y.{task}"] = y.numpy()[:, :, ::24]
        out[f"T9000.y.{task}.norm"] = np.array([y.norm().item(), y.abs().max().item()])
        for k, v in taps.items():
            out[f"T9000.tap.{task}.{k}"] = v
        taps.clear()
    # ---- DDIM through the reference sampler with injected noise ----------------------------------------------------------
    betas, _ = get_beta_schedule("linear", 1000)

    def ddim(B, T_, S, scale, causal=False, task="text_guided"):
        _, cond = _inputs(B, T_, task)
        cond_t = {k: (None if v is None else T(v)) for k, v in cond.items()}
        shape = (B, 128, T_)
        init = synth.noise_list(1, shape, seed=7)[0]
        noises = synth.noise_list(S, shape, seed=11)
        gd = GaussianDiffusion(steps=1000, betas=betas.float(), objective="noise", loss_type="l2", device="cpu",
                               cfg_dropout_proba=0.0, embedding_scale=scale, batch_cfg=True, scale_cfg=True,
                               sampling_timesteps=S)
        it = iter([T(init)] + [T(n) for n in noises])
        r_randn, r_randn_like = torch.randn, torch.randn_like
        torch.randn = lambda *a, **k: next(it).clone()
        torch.randn_like = lambda *a, **k: next(it).clone()
        try:
            y = gd.sample(model, shape, cond_t, causal=causal)
        finally:
            torch.randn, torch.randn_like = r_randn, r_randn_like
        return y.numpy()

    out["ddim10.B2.cfg"] = ddim(2, 1500, 10, 0.8)[:, :, ::8]
    out["ddim2.B8.cfg"] = ddim(8, 1500, 2, 0.8)[:, :, ::16]
    out["ddim2.B8.nocfg"] = ddim(8, 1500, 2, 1.0)[:, :, ::16]
    out["ddim2.T9000.cont"] = ddim(1, 9000, 2, 0.8, causal=True, task="music_cont")[:, :, ::24]
    save("full_bench", **out)


def gen_full_train():
    """full configuration, one clip through the CFG pair: loss + every parameter's gradient from the reference's autograd"""
    cfg = full_model_config()
    model, spec = _build(cfg)
    B, T_ = 1, 1500
    betas, _ = get_beta_schedule("linear", 1000)
    out = {"grad_names_all": np.array(json.dumps([n for n, _ in model.named_parameters()]))}
    x0 = synth.latents(B, T_, key="clip")
    cond = synth.conditioning(B, T_, "music_cont")
    cond_t = {k: (None if v is None else T(v)) for k, v in cond.items()}
    noise = fill_uniform("synth.trainnoise.full", (B, 128, T_), 3, 0.0, 1.0)
    t = torch.tensor([417], dtype=torch.long)
    gd = GaussianDiffusion(steps=1000, betas=betas.float(), objective="noise", loss_type="l2", device="cpu",
                           cfg_dropout_proba=0.0, embedding_scale=0.8, batch_cfg=True, scale_cfg=True)
    with torch.enable_grad():
        model.zero_grad(set_to_none=True)
        model.train()
        loss = gd.training_loosses(model, T(x0), t, cond_t, noise=T(noise), causal=True)
        loss.backward()
    out["loss"] = np.float32(loss.item())
    out["gradnorm_all"] = np.array([p_.grad.norm().item() for _, p_ in model.named_parameters()], dtype=np.float32)
    out["gradsample_all"] = np.concatenate(
        [p_.grad.reshape(-1)[:: max(1, p_.numel() // 16)][:16].numpy() for _, p_ in model.named_parameters()]).astype(np.float32)
    save("full_train", **out)



def _ref_ddim(model, B, T_, S, scale, eta=1.0, causal=False, task="text_guided", init_eps=0.0, all_steps=False):
    """the reference's GaussianDiffusion.sample (DDIM) on the full model with injected start / per-step noise"""
    betas, _ = get_beta_schedule("linear", 1000)
    _, cond = _inputs(B, T_, task)
    cond_t = {k: (None if v is None else T(v)) for k, v in cond.items()}
    shape = (B, 128, T_)
    init = synth.noise_list(1, shape, seed=7)[0]
    if init_eps:
        init = (init * np.float32(1.0 + init_eps)).astype(np.float32)
    noises = synth.noise_list(S, shape, seed=11)
    gd = GaussianDiffusion(steps=1000, betas=betas.float(), objective="noise", loss_type="l2", device="cpu",
                           cfg_dropout_proba=0.0, embedding_scale=scale, batch_cfg=True, scale_cfg=True,
                           sampling_timesteps=S, ddim_sampling_eta=eta)
    it = iter([T(init)] + [T(n) for n in noises])
    r_randn, r_randn_like = torch.randn, torch.randn_like
    torch.randn = lambda *a, **k: next(it).clone()
    torch.randn_like = lambda *a, **k: next(it).clone()
    try:
        y = gd.sample(model, shape, cond_t, return_all_timesteps=all_steps, causal=causal)
    finally:
        torch.randn, torch.randn_like = r_randn, r_randn_like
    return y.numpy()


def gen_ddim100():
    """BASELINE configs[1] / [2] at their own schedule length: the reference's 100-step DDIM of the full model.
      ddim100.B2.cfg[.stepK]   B = 2, CFG pair (scale 0.8, rescale), eta = 1 with injected noise; the trajectory at K = 10/25/50/75
      ddim100.B8.nocfg[.stepK] B = 8, no CFG, eta = 0 -- exactly what bench.py times
      *.sens[.stepK]           the REFERENCE's own response to a start noise scaled by (1 + 1e-6), as (max-abs, max-ref, rel-L2):
                               the x0 clamp makes the 100-step chain amplify rounding-level differences, so the final latents are
                               pinned only as tightly as the reference pins itself; the early trajectory is pinned tightly
    Outputs sub-sampled along T."""
    import time as _time
    cfg = full_model_config()
    model, _ = _build(cfg)
    out = {}
    taps = (10, 25, 50, 75)

    def sens(a, b):
        return np.array([np.abs(a - b).max(), np.abs(b).max(), np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum())])

    for key, B, scale, eta, sub in (("ddim100.B2.cfg", 2, 0.8, 1.0, 8), ("ddim100.B8.nocfg", 8, 1.0, 0.0, 16)):
        t0 = _time.time()
        traj = _ref_ddim(model, B, 1500, 100, scale, eta=eta, all_steps=True)          # [B, S + 1, C, T]; entry k = input of step k
        print(f"  {key}: {_time.time() - t0:.0f} s")
        pert = _ref_ddim(model, B, 1500, 100, scale, eta=eta, init_eps=1e-6, all_steps=True)
        # gd.sample(return_all_timesteps=True) stacks the INPUT of every step (gdm.py:205) and never the final output; run the
        # last entry again without the stack for the final latents
        fin = _ref_ddim(model, B, 1500, 100, scale, eta=eta)
        fin_p = _ref_ddim(model, B, 1500, 100, scale, eta=eta, init_eps=1e-6)
        out[key] = fin[:, :, ::sub]
        out[key + ".sens"] = sens(fin_p, fin)
        print("   final sensitivity (max-abs, max-ref, rel-L2):", out[key + ".sens"])
        for k in taps:
            out[f"{key}.step{k}"] = traj[:, k, :, ::sub]
            out[f"{key}.sens.step{k}"] = sens(pert[:, k], traj[:, k])
            print(f"   step {k} sensitivity:", out[f"{key}.sens.step{k}"])
        del traj, pert
    save("full_ddim100", **out)


TRAIN8_TASKS = synth.TRAIN8_TASKS
train8_inputs = synth.train8_inputs


def gen_full_train8():
    """full configuration, 8 clips as the 3 / 3 / 2 task sub-batches of one micro-batch: per-task losses, the summed loss
    and every parameter's gradient of ``sum(task losses).backward()`` from the reference's autograd"""
    cfg = full_model_config()
    model, spec = _build(cfg)
    betas, _ = get_beta_schedule("linear", 1000)
    gd = GaussianDiffusion(steps=1000, betas=betas.float(), objective="noise", loss_type="l2", device="cpu",
                           cfg_dropout_proba=0.0, embedding_scale=0.8, batch_cfg=True, scale_cfg=True)
    out = {"grad_names_all": np.array(json.dumps([n for n, _ in model.named_parameters()]))}
    with torch.enable_grad():
        model.zero_grad(set_to_none=True)
        model.train()
        for task, x0, t, cond, noise, causal in train8_inputs():
            cond_t = {k: (None if v is None else T(v)) for k, v in cond.items()}
            loss = gd.training_loosses(model, T(x0), T(t), cond_t, noise=T(noise), causal=causal)
            loss.backward()                                            # gradients of the three tasks accumulate (trainer.py:141)
            out[f"loss.{task}"] = np.float32(loss.item())
            print("  ", task, loss.item())
    out["loss"] = np.float32(sum(float(out[f"loss.{t}"]) for t, _, _ in TRAIN8_TASKS))
    out["gradnorm_all"] = np.array([p_.grad.norm().item() for _, p_ in model.named_parameters()], dtype=np.float32)
    out["gradsample_all"] = np.concatenate(
        [p_.grad.reshape(-1)[:: max(1, p_.numel() // 16)][:16].numpy() for _, p_ in model.named_parameters()]).astype(np.float32)
    save("full_train8", **out)


def encodec_decoder_shapes():
    """(key, shape) of transformers' EncodecDecoder for the 48 kHz configuration (the names jen1_amd/encodec.py reads)"""
    from transformers import EncodecConfig, EncodecModel
    cfg = EncodecConfig(sampling_rate=48000, audio_channels=2, normalize=True, chunk_length_s=1.0, overlap=0.01,
                        use_causal_conv=False, norm_type="time_group_norm", upsampling_ratios=[8, 5, 4, 2],
                        target_bandwidths=[3.0, 6.0, 12.0, 24.0])
    return EncodecModel(cfg), cfg


def gen_encodec():
    """SEANet decoder + RVQ decode of the Hugging Face port of Encodec 48 kHz (architecture oracle; the ``encodec``
    package the reference imports is not installed), synthetic weights by key name."""
    model, cfg = encodec_decoder_shapes()
    dec = model.decoder
    sd = dec.state_dict()
    dec.load_state_dict({k: T(fill("encodec.decoder." + k, tuple(v.shape), SEED)) for k, v in sd.items()})
    dec.eval()
    out = {"schema": np.array(json.dumps([(k, list(v.shape)) for k, v in sd.items()]))}
    emb = fill_normal("encodec.emb", (2, 128, 37), 5)
    taps = {}
    dec.layers[0].register_forward_hook(lambda m, i, o: taps.__setitem__("conv0", o.numpy().copy()))
    dec.layers[1].register_forward_hook(lambda m, i, o: taps.__setitem__("lstm", o.numpy().copy()))
    y = dec(T(emb)).numpy()
    assert y.shape == (2, 2, 37 * 320), y.shape
    out["decoder.y"] = y
    out["decoder.tap.conv0"] = taps["conv0"][:, ::8, :]
    out["decoder.tap.lstm"] = taps["lstm"][:, ::8, :]
    # quantizer.decode: 16 codebooks of 1024 x 128
    q = model.quantizer
    nq = len(q.layers)
    for i, layer in enumerate(q.layers):
        layer.codebook.embed.copy_(T(fill_normal(f"encodec.quantizer.layers.{i}.codebook.embed", (1024, 128), SEED)))
    g = np.random.Generator(np.random.Philox(key=[77, 0x6A656E31]))
    codes = g.integers(0, 1024, size=(nq, 2, 53), dtype=np.int64)
    out["rvq.codes"] = codes
    out["rvq.y"] = q.decode(T(codes)).numpy()
    out["rvq.n_q"] = np.int64(nq)
    # encoder half: SEANet encoder on one normalised segment, then the RVQ search with all 16 codebooks (24 kbps)
    enc = model.encoder
    esd = enc.state_dict()
    enc.load_state_dict({k: T(fill("encodec.encoder." + k, tuple(v.shape), SEED)) for k, v in esd.items()})
    enc.eval()
    out["enc_schema"] = np.array(json.dumps([(k, list(v.shape)) for k, v in esd.items()]))
    audio = fill_normal("encodec.audio", (2, 2, 9600 + 123), 5) * 0.3
    e = enc(T(audio))
    assert e.shape == (2, 128, -(-(9600 + 123) // 320)), e.shape
    out["encoder.y"] = e.numpy()
    out["encoder.codes"] = q.encode(e, 24.0).numpy()
    save("encodec", **out)


def gen_ddpm(model=None):
    """ancestral sampling (gdm.py:144-179) of the reference: GaussianDiffusion(steps=20).p_sample_loop called directly (sample()
    passes causal= to it and raises TypeError, gdm.py:229-230); start noise (randn) and the per-step noise (rand_like: UNIFORM, as
    written) are injected"""
    cfg = tiny_model_config()
    if model is None:
        model, _ = _build(cfg)
    B, T_, S = 2, 300, 20
    _, cond = _inputs(B, T_)
    cond_t = {k: (None if v is None else T(v)) for k, v in cond.items()}
    betas = torch.linspace(1e-3, 0.35, S, dtype=torch.float32)       # (the 'linear' schedule reaches beta = 1 at 20 steps: 1 / alphas_cumprod = inf)
    shape = (B, 128, T_)
    init = synth.noise_list(1, shape, seed=17)[0]
    noises = synth.noise_list(S, shape, seed=19, uniform=True)
    out = {"betas": betas.numpy()}

    def run(scale, batch_cfg, scale_cfg, all_steps=False):
        gd = GaussianDiffusion(steps=S, betas=betas, objective="noise", loss_type="l2", device="cpu", cfg_dropout_proba=0.0,
                               embedding_scale=scale, batch_cfg=batch_cfg, scale_cfg=scale_cfg)
        assert not gd.is_ddim_sampling
        it = iter([T(n) for n in noises])
        r_randn, r_rand_like = torch.randn, torch.rand_like
        torch.randn = lambda *a, **k: T(init).clone()
        torch.rand_like = lambda *a, **k: next(it).clone()
        try:
            try:
                gd.sample(model, shape, cond_t)
                raise AssertionError("the reference's sample() was expected to fail for non-DDIM sampling")
            except TypeError:
                pass
            y = gd.p_sample_loop(model, shape, cond_t, return_all_timesteps=all_steps)
        finally:
            torch.randn, torch.rand_like = r_randn, r_rand_like
        return y.numpy()

    out["ddpm20.cfg"] = run(0.8, True, True)
    out["ddpm20.nocfg"] = run(1.0, False, False)[:, :, ::3]
    traj = run(0.8, True, True, all_steps=True)
    assert traj.shape == (B, S + 1, 128, T_)
    out["ddpm20.cfg.traj"] = traj[:, :, ::8, ::15]
    save("tiny_ddpm", **out)


def gen_vdm(model=None):
    """The reference's VDM (jen1/diffusion/vdm/vdm.py) cannot run as shipped (both failures are
    asserted below).  The fixture pins the REPAIRED form: a subclass of the reference's own class that overrides the two broken
    methods with the same formulas and the three repairs jen1_amd/vdm.py documents -- the model gets ``time.expand(B)``,
    ``alphas`` / ``sigmas`` are read by step index, and the loss reshapes them to [B, 1, 1].  Tiny configuration: sampling with
    10 and 4 steps (CFG pair; no CFG + causal), the whole trajectory of the 10-step run, the loss and every gradient norm."""
    import math
    from jen1.diffusion.vdm.vdm import VDM
    cfg = tiny_model_config()
    if model is None:
        model, _ = _build(cfg)
    B, T_ = 2, 300
    _, cond = _inputs(B, T_)
    cond_t = {k: (None if v is None else T(v)) for k, v in cond.items()}
    shape = (B, 128, T_)
    init = synth.noise_list(1, shape, seed=23)[0]

    # ---- the reference as shipped fails (A-3, A-4) --------------------------------------------------------------------------
    ref = VDM(loss_type="l2", device="cpu", cfg_dropout_proba=0.0, embedding_scale=0.8, batch_cfg=True, scale_cfg=True)
    for what, fn in (("sample", lambda: ref.sample(model, shape, cond_t, step=2)),
                     ("training_loosses", lambda: ref.training_loosses(model, T(synth.latents(B, T_, key="clip")), cond_t))):
        try:
            fn()
            raise AssertionError(f"the reference's VDM.{what} was expected to fail")
        except AssertionError:
            raise
        except Exception as e:                                        # noqa: BLE001
            print(f"  reference VDM.{what} fails as documented: {type(e).__name__}")

    class RepairedVDM(VDM):
        def p_sample_i(self, x, i, model, conditioning, causal):
            time = self.steps[i].expand(x.shape[0])                    # repair: a batch vector of the step's time
            v_pred = model(x, time, embedding=conditioning['cross_attn_cond'], embedding_mask=conditioning['cross_attn_masks'],
                           embedding_scale=self.embedding_scale, embedding_mask_proba=self.cfg_dropout_proba,
                           features=conditioning['global_cond'], channels_list=[conditioning['input_concat_cond']],
                           batch_cfg=self.batch_cfg, scale_cfg=self.scale_cfg, causal=causal)
            x_pred = self.alphas[i] * x - self.sigmas[i] * v_pred      # repair: indexed by the step, not by its float time
            noise_pred = self.sigmas[i] * x + self.alphas[i] * v_pred
            return self.alphas[i + 1] * x_pred + self.sigmas[i + 1] * noise_pred

        def p_sample_loop(self, model, shape, conditioning, step=1000, return_all_timesteps=False, init_data=None, causal=False):
            audio = torch.randn(shape, device=self.device)
            if init_data is not None:
                audio = audio + init_data
            audios = [audio]
            self.steps = torch.linspace(1., 0., step + 1, device=self.device)
            self.get_alpha_sigma(self.steps)
            for i in range(step):
                audio = self.p_sample_i(audio, i, model, conditioning, causal)
                audios.append(audio)
            return audio if not return_all_timesteps else torch.stack(audios, dim=1)

        def training_loosses(self, model, x_start, conditioning, noise=None, causal=False, times=None):
            if noise is None:
                noise = torch.rand_like(x_start)
            if times is None:
                times = torch.rand(x_start.shape[0])
            x_t, alphas, sigmas = self.q_sample(x_start, times.reshape(-1, 1, 1), noise=noise)       # repair: [B, 1, 1]
            model_out = model(x_t, times, embedding=conditioning['cross_attn_cond'], embedding_mask=conditioning['cross_attn_masks'],
                              embedding_scale=self.embedding_scale, embedding_mask_proba=self.cfg_dropout_proba,
                              features=conditioning['global_cond'], channels_list=[conditioning['input_concat_cond']],
                              batch_cfg=self.batch_cfg, scale_cfg=self.scale_cfg, causal=causal)
            target = noise * alphas - x_t * sigmas
            loss = self.loss_fn(model_out, target, reduction='none')
            return loss.reshape(loss.shape[0], -1).mean(dim=1).mean()

    out = {}

    def run(scale, step, causal=False, all_steps=False):
        vd = RepairedVDM(loss_type="l2", device="cpu", cfg_dropout_proba=0.0, embedding_scale=scale, batch_cfg=True, scale_cfg=True)
        r_randn = torch.randn
        torch.randn = lambda *a, **k: T(init).clone()
        try:
            y = vd.sample(model, shape, cond_t, step=step, return_all_timesteps=all_steps, causal=causal)
        finally:
            torch.randn = r_randn
        return y.numpy()

    out["vdm10.cfg"] = run(0.8, 10)
    out["vdm10.cfg.traj"] = run(0.8, 10, all_steps=True)[:, :, ::8, ::15]
    out["vdm4.nocfg.causal"] = run(1.0, 4, causal=True)[:, :, ::3]
    # ---- loss + gradients ----------------------------------------------------------------------------------------------------
    x0 = synth.latents(B, T_, key="clip")
    noise = fill_uniform("synth.trainnoise.vdm", (B, 128, T_), 3, 0.0, 1.0)
    times = torch.tensor([0.137, 0.803], dtype=torch.float32)
    out["times"] = times.numpy()
    vd = RepairedVDM(loss_type="l2", device="cpu", cfg_dropout_proba=0.0, embedding_scale=0.8, batch_cfg=True, scale_cfg=True)
    with torch.enable_grad():
        model.zero_grad(set_to_none=True)
        model.train()
        loss = vd.training_loosses(model, T(x0), cond_t, noise=T(noise), causal=False, times=times)
        loss.backward()
    out["loss"] = np.float32(loss.item())
    out["grad_names_all"] = np.array(json.dumps([n for n, _ in model.named_parameters()]))
    out["gradnorm_all"] = np.array([p_.grad.norm().item() for _, p_ in model.named_parameters()], dtype=np.float32)
    model.eval()
    save("tiny_vdm", **out)


def gen_host():
    """The host functions either side of the denoiser, from the reference's own trainer.py / generation.py (imported with stand-ins
    for the absent ``encodec`` package; called unbound on a namespace that carries the attributes they read):
      trainer.py:215-247 random_mask (the ``random`` module seeded per case), :249-278 get_conditioning,
      generation.py:134-143 get_mask, :152-192 get_conditioning,
    and the checkpoint format (script_util.py:79-124): the reference's save_checkpoint writes a tiny-config file with its
    model + torch AdamW, the build's load_checkpoint reads it; the build's save_checkpoint writes one, the reference's
    load_checkpoint reads it.  Stored: the key list, per-tensor checksums and the outcomes -- no weights."""
    import hashlib
    import logging
    import random
    import tempfile
    enc = types.ModuleType("encodec")
    enc.EncodecModel = type("EncodecModel", (), {})
    encu = types.ModuleType("encodec.utils")
    encu.convert_audio = lambda wav, sr, target_sr, target_channels: wav
    sys.modules.setdefault("encodec", enc)
    sys.modules.setdefault("encodec.utils", encu)
    import trainer as rtr
    import generation as rgen
    from utils import script_util as rsu
    out = {}
    # ---- random_mask ---------------------------------------------------------------------------------------------
    cases = []
    for L_ in (1500, 375, 300):
        seq = T(synth.latents(3, L_, key="clip"))
        for task in ("text_guided", "music_inpaint", "music_cont"):
            for seed in (0, 1, 2, 3, 4, 5, 6):
                random.seed(seed)
                try:
                    masked, mask, causal = rtr.UnifiedMultiTaskTrainer.random_mask(None, seq, L_, task)
                except Exception as e:      # non-integral float bounds: random.randint refuses them on this Python
                    cases.append([L_, task, seed, type(e).__name__])
                    continue
                assert torch.equal(masked, seq * mask) and mask.shape == (3, 1, L_) and bool((mask[0] == mask[2]).all())
                k = f"mask.{L_}.{task}.{seed}"
                out[k] = mask[0, 0].numpy().astype(np.uint8)
                out[k + ".causal"] = np.bool_(causal)
                cases.append([L_, task, seed, "ok"])
    out["mask.cases"] = np.array(json.dumps(cases))
    # ---- get_conditioning (trainer form and generation form) ---------------------------------------------------------
    B, T_ = 3, 40
    g = np.random.Generator(np.random.Philox(key=[5, 0x6A656E31]))
    emb = g.standard_normal((B, 16, 24)).astype(np.float32)
    emb2 = g.standard_normal((B, 4, 24)).astype(np.float32)
    msk = (g.random((B, 16)) > 0.3)
    msk2 = (g.random((B, 4)) > 0.3)
    glob = g.standard_normal((B, 1, 12)).astype(np.float32)
    masked_in = g.standard_normal((B, 8, T_)).astype(np.float32)
    keep = (g.random((B, 1, T_)) > 0.5).astype(np.float32)
    for k, v in (("emb", emb), ("emb2", emb2), ("msk", msk), ("msk2", msk2), ("glob", glob), ("masked_in", masked_in), ("keep", keep)):
        out["cond.in." + k] = v
    cond = {"prompt": (T(emb), T(msk)), "style": (T(emb2), T(msk2)), "g": (T(glob), None), "masked_input": T(masked_in), "mask": T(keep)}
    ns = types.SimpleNamespace(cross_attn_cond_ids=["prompt", "style"], global_cond_ids=["g"], input_concat_ids=["masked_input", "mask"])
    r = rtr.UnifiedMultiTaskTrainer.get_conditioning(ns, cond)
    for k, v in r.items():
        out["cond.trainer." + k] = v.numpy()
    ns.batch_size = B
    r = rgen.Jen1.get_conditioning(ns, cond)
    for k, v in r.items():
        out["cond.generation." + k] = v.numpy()
    # ---- get_mask ----------------------------------------------------------------------------------------------------
    ns = types.SimpleNamespace(sample_rate=48000)
    gm = []
    for n, a, b, bs in ((96000, 0.0, 2.0, 2), (96000, 0.5, 1.5, 1), (48000, 0.33333, 0.77777, 3), (1000, 0.0101, 0.0199, 1)):
        m = rgen.Jen1.get_mask(ns, n, a, b, bs)
        assert m.shape == (bs, 1, n)
        z = np.flatnonzero(m[0, 0].numpy() == 0)
        gm.append([n, a, b, bs, int(z[0]) if len(z) else -1, int(z[-1]) if len(z) else -1, int(len(z))])
    out["get_mask.cases"] = np.array(json.dumps(gm))
    # ---- checkpoint format, both directions ----------------------------------------------------------------------------
    from jen1_amd import checkpoint as mck
    from jen1_amd.model import UNetCFG1d as MyUNet
    from jen1_amd.optim import FusedAdamW
    cfg = tiny_model_config()
    rmodel, spec = _build(cfg)
    rmodel.train()
    params = list(rmodel.parameters())
    ropt = torch.optim.AdamW(params, lr=3e-5, betas=(0.9, 0.95), weight_decay=0.1)
    with torch.enable_grad():
        for it in range(2):
            ropt.zero_grad()
            for i, p_ in enumerate(params):
                p_.grad = T(fill_normal(f"ckpt.grad.{it}.{i}", tuple(p_.shape), 3)) * 1e-2
            ropt.step()
    digest = lambda t: hashlib.sha256(np.ascontiguousarray(t.detach().numpy()).tobytes()).hexdigest()[:16]
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "G_2.pth")
        rsu.save_checkpoint(rmodel, ropt, 3e-5, 2, path, logging.getLogger("golden"))
        # the reference's file into the build
        mine = MyUNet(**cfg, compute_dtype="f32", device="cpu", init_seed=None)
        mopt = FusedAdamW(mine.parameters())
        _, _, lr, epoch = mck.load_checkpoint(path, mine, optimizer=mopt)
        rsd = rmodel.state_dict()
        assert lr == 3e-5 and epoch == 2 and mopt.step_count == 2
        assert all(torch.equal(v, rsd[k]) for k, v in mine.state_dict().items()) and list(mine.state_dict()) == list(rsd)
        rst = ropt.state_dict()["state"]
        for i, (p_, o) in enumerate(zip(mopt.params, mopt.offsets)):
            assert torch.equal(mopt.exp_avg[o:o + p_.numel()].view_as(p_), rst[i]["exp_avg"])
            assert torch.equal(mopt.exp_avg_sq[o:o + p_.numel()].view_as(p_), rst[i]["exp_avg_sq"])
        out["ckpt.keys"] = np.array(json.dumps(list(rsd)))
        out["ckpt.file_keys"] = np.array(json.dumps(sorted(torch.load(path, map_location="cpu", weights_only=False))))
        out["ckpt.opt_group_keys"] = np.array(json.dumps(sorted(k for k in ropt.state_dict()["param_groups"][0])))
        out["ckpt.param_digest"] = np.array(json.dumps({k: digest(v) for k, v in rsd.items()}))
        out["ckpt.exp_avg_digest"] = np.array(json.dumps([digest(rst[i]["exp_avg"]) for i in range(len(params))]))
        out["ckpt.exp_avg_sq_digest"] = np.array(json.dumps([digest(rst[i]["exp_avg_sq"]) for i in range(len(params))]))
        # the build's file into the reference
        path2 = os.path.join(d, "G_3.pth")
        mck.save_checkpoint(mine, mopt, 1e-5, 3, path2)
        r2, _ = _build(cfg)
        for p_ in r2.parameters():
            p_.data.zero_()
        ropt2 = torch.optim.AdamW(r2.parameters(), lr=1.0)
        _, _, lr2, epoch2 = rsu.load_checkpoint(path2, r2, logger=None, optimizer=ropt2)
        assert lr2 == 1e-5 and epoch2 == 3
        assert all(torch.equal(v, rsd[k]) for k, v in r2.state_dict().items())
        st2 = ropt2.state_dict()
        assert all(torch.equal(st2["state"][i]["exp_avg"], rst[i]["exp_avg"]) and float(st2["state"][i]["step"]) == 2.0 for i in range(len(params)))
        assert st2["param_groups"][0]["lr"] == 3e-5 and st2["param_groups"][0]["betas"] == (0.9, 0.95)
        out["ckpt.cross_load"] = np.array(json.dumps({"reference_file_into_build": True, "build_file_into_reference": True}))
    save("host_pins", **out)


if __name__ == "__main__":
    which = set(sys.argv[1:]) or {"schedule", "units", "tiny", "sampler", "train", "full", "fullbench", "fulltrain", "encodec", "ddpm", "host", "ddim100", "fulltrain8", "vdm"}
    model = None
    if "schedule" in which:
        print("schedule"); gen_schedule()
    if "units" in which:
        print("units"); gen_units()
    if "tiny" in which:
        print("tiny"); model = gen_tiny()
    if "sampler" in which:
        print("sampler"); gen_tiny_sampler(model)
    if "train" in which:
        print("train"); gen_tiny_train(model)
    if "full" in which:
        print("full"); gen_full()
    if "fullbench" in which:
        print("fullbench"); gen_full_bench()
    if "fulltrain" in which:
        print("fulltrain"); gen_full_train()
    if "ddim100" in which:
        print("ddim100"); gen_ddim100()
    if "fulltrain8" in which:
        print("fulltrain8"); gen_full_train8()
    if "encodec" in which:
        print("encodec"); gen_encodec()
    if "ddpm" in which:
        print("ddpm"); gen_ddpm(model)
    if "vdm" in which:
        print("vdm"); gen_vdm(model)
    if "host" in which:
        print("host"); gen_host()
