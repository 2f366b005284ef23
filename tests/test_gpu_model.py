"""-m gpu: whole-denoiser parity on the MI355X, through the reference-shaped host API.

float32 mode is the parity gate BASELINE.json states: max-abs error / max-abs reference
<= 1e-3 against (a) the golden fixtures produced by the reference itself and (b) the CPU
oracle on the same seeded inputs.  bf16 mode is reported against a looser, separately stated
tolerance (the reference's own bf16 autocast drifts 9.5e-3).
"""
import numpy as np
import pytest
import torch

from helpers import bf16_gate, record_parity, filled, golden, rel_err
from jen1_amd import synth
from jen1_amd.config import UNetSpec, full_model_config, tiny_model_config

pytestmark = pytest.mark.gpu

F32_TOL = 1e-3
BF16_TOL = 5e-2


def dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def tiny_models():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from jen1_amd.model import UNetCFG1d
    cfg = tiny_model_config()
    return {m: UNetCFG1d(**cfg, init_seed=1234, compute_dtype=m, device="cuda") for m in ("f32", "bf16")}


@pytest.fixture(scope="module")
def oracle_tiny():
    from oracle import jen1_oracle as O
    cfg = tiny_model_config()
    return O.OracleUNetCFG1d(filled(UNetSpec(**cfg).param_shapes()), **cfg)


def _inputs(task="text_guided", B=2, T=300):
    return synth.latents(B, T), np.array([999, 499][:B] + [7] * max(0, B - 2), dtype=np.int64), synth.conditioning(B, T, task)


def _fwd(model, x, t, cond, **kw):
    y = model(dev(x), dev(t), embedding=dev(cond["cross_attn_cond"]),
              embedding_mask=dev(cond["cross_attn_masks"]) if kw.pop("use_mask", True) else None,
              channels_list=[dev(cond["input_concat_cond"])], **kw)
    torch.cuda.synchronize()
    return y.cpu().numpy()


def test_state_dict_schema_matches_reference(tiny_models):
    import json
    sch = json.loads(str(golden("tiny_unet")["schema"]))
    sd = tiny_models["f32"].state_dict()
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == [(k, tuple(s)) for k, s in sch]


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_tiny_unet_all_cfg_branches_vs_golden(tiny_models, mode):
    g = golden("tiny_unet")
    x, t, cond = _inputs()
    tol = F32_TOL if mode == "f32" else BF16_TOL
    worst = 0.0
    for key in [k for k in g.files if k.startswith("y.s")]:
        scale_s, rest = key[3:].split(".b")
        b, r, c = rest[0] == "1", rest[3] == "1", rest[6] == "1"
        y = _fwd(tiny_models[mode], x, t, cond, embedding_scale=float(scale_s), embedding_mask_proba=0.0, batch_cfg=b,
                 scale_cfg=r, causal=c)
        e = rel_err(y[:, :, ::3], g[key])
        worst = max(worst, e)
        assert e < tol, (key, e)
    print(f"[{mode}] worst max-abs/max-ref over 10 CFG/causal cases: {worst:.3e}")


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_tiny_unet_inpaint_dropout_nomask_vs_golden(tiny_models, mode):
    g = golden("tiny_unet")
    tol = F32_TOL if mode == "f32" else BF16_TOL
    m = tiny_models[mode]
    x, t, cond = _inputs()
    xi, _, cond_i = _inputs("music_inpaint")
    y = _fwd(m, xi, t, cond_i, embedding_scale=0.8, batch_cfg=True, scale_cfg=True)
    assert rel_err(y, g["y.inpaint"]) < tol
    y = _fwd(m, x, t, cond, embedding_scale=0.8, batch_cfg=True, scale_cfg=True, embedding_mask_proba=0.2,
             dropout_rows=torch.tensor([False, True]))
    assert rel_err(y[:, :, ::3], g["y.dropout_row1"]) < tol
    y = _fwd(m, x, t, cond, embedding_scale=0.8, batch_cfg=True, scale_cfg=False, use_mask=False)
    assert rel_err(y[:, :, ::3], g["y.nomask"]) < tol


def test_tiny_unet_levels_vs_oracle(tiny_models, oracle_tiny):
    """per-level taps of the fp32 engine against the oracle: localises a failing block."""
    x, t, cond = _inputs()
    m = tiny_models["f32"]
    _fwd(m, x, t, cond, embedding_scale=1.0, causal=True)
    oracle_tiny(x, t, embedding=cond["cross_attn_cond"], embedding_mask=cond["cross_attn_masks"], embedding_scale=1.0,
                channels_list=[cond["input_concat_cond"]], causal=True)
    plan = m.engine().plan(2, 300, 1, True)
    for name in ("to_in", "down0", "down1", "bottleneck", "up0", "up1"):
        a = plan.taps[name]
        got = a.t[:, :, : a.C].float().permute(0, 2, 1).cpu().numpy()
        ref = oracle_tiny.taps[name]
        if name == "up1":      # the engine fuses `x += skips_list.pop()` (model.py:261) into the last upsample
            ref = ref + oracle_tiny.taps["to_in"]
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        assert rel_err(got, ref) < F32_TOL, (name, rel_err(got, ref))


@pytest.mark.parametrize("B,T", [(1, 64), (3, 301), (5, 97)])
def test_tiny_unet_odd_shapes_vs_oracle(tiny_models, oracle_tiny, B, T):
    """ragged batch / odd lengths (crop path, partial tiles) against the oracle in fp32."""
    x, cond = synth.latents(B, T), synth.conditioning(B, T, "music_cont")
    t = np.array([(37 * i + 5) % 1000 for i in range(B)], dtype=np.int64)
    ref = oracle_tiny(x, t, embedding=cond["cross_attn_cond"], embedding_mask=cond["cross_attn_masks"], embedding_scale=0.8,
                      batch_cfg=True, scale_cfg=True, channels_list=[cond["input_concat_cond"]], causal=True)
    y = _fwd(tiny_models["f32"], x, t, cond, embedding_scale=0.8, batch_cfg=True, scale_cfg=True, causal=True)
    assert rel_err(y, ref) < F32_TOL


def test_forward_is_pure_and_repeatable(tiny_models):
    """inputs are not mutated and a repeated call returns the same values (atomics only reorder fp32 sums)."""
    x, t, cond = _inputs()
    m = tiny_models["f32"]
    xd = dev(x)
    x0 = xd.clone()
    kw = dict(embedding=dev(cond["cross_attn_cond"]), embedding_mask=dev(cond["cross_attn_masks"]),
              channels_list=[dev(cond["input_concat_cond"])], embedding_scale=0.8, batch_cfg=True, scale_cfg=True)
    y1 = m(xd, dev(t), **kw)
    y2 = m(xd, dev(t), **kw)
    torch.cuda.synchronize()
    assert torch.equal(xd, x0)
    assert y1.data_ptr() != y2.data_ptr()
    assert rel_err(y1.cpu().numpy(), y2.cpu().numpy()) < 1e-5


# ------------------------------------------------------------------ sampler
@pytest.mark.parametrize("use_graph", [False, True])
def test_tiny_ddim_sampler_vs_golden(tiny_models, use_graph):
    from jen1_amd.diffusion import GaussianDiffusion, get_beta_schedule
    g = golden("tiny_sampler")
    B, T, S = 2, 300, 10
    cond = {k: dev(v) for k, v in synth.conditioning(B, T).items()}
    shape = (B, 128, T)
    init = dev(synth.noise_list(1, shape, seed=7)[0])
    noises = [dev(n) for n in synth.noise_list(S, shape, seed=11)]
    betas, _ = get_beta_schedule("linear", 1000)
    m = tiny_models["f32"]

    def run(proba, scale, bcfg, rcfg, causal, drops=None, objective="noise"):
        gd = GaussianDiffusion(steps=1000, betas=betas, objective=objective, loss_type="l2", device="cuda",
                               cfg_dropout_proba=proba, embedding_scale=scale, batch_cfg=bcfg, scale_cfg=rcfg,
                               sampling_timesteps=S)
        y = gd.sample(m, shape, cond, causal=causal, init_noise=init, step_noises=noises, dropout_rows=drops, use_graph=use_graph)
        torch.cuda.synchronize()
        return y.cpu().numpy()

    assert rel_err(run(0.0, 0.8, True, True, False), g["ddim10.cfg"]) < F32_TOL
    assert rel_err(run(0.0, 1.0, False, False, True)[:, :, ::3], g["ddim10.nocfg.causal"]) < F32_TOL
    assert rel_err(run(0.2, 0.8, True, True, False, drops=g["ddim10.dropout.rows"])[:, :, ::3], g["ddim10.dropout"]) < F32_TOL
    assert rel_err(run(0.0, 0.8, True, True, False, objective="x0")[:, :, ::3], g["ddim10.x0"]) < F32_TOL
    assert rel_err(run(0.0, 0.8, True, True, False, objective="v")[:, :, ::3], g["ddim10.v"]) < F32_TOL


@pytest.mark.parametrize("use_graph", [False, True])
def test_tiny_ddpm_sampler_vs_golden(tiny_models, use_graph):
    """ancestral sampling (gdm.py:144-179) on the fused stepper against the reference's p_sample_loop with
    GaussianDiffusion(steps=20): final latents with and without CFG, the whole trajectory, and the literal (unfused) loop"""
    from jen1_amd.diffusion import GaussianDiffusion
    g = golden("tiny_ddpm")
    B, T, S = 2, 300, 20
    cond = {k: dev(v) for k, v in synth.conditioning(B, T).items()}
    shape = (B, 128, T)
    init = dev(synth.noise_list(1, shape, seed=17)[0])
    noises = [dev(n) for n in synth.noise_list(S, shape, seed=19, uniform=True)]
    betas = torch.from_numpy(g["betas"])
    m = tiny_models["f32"]

    def run(scale, bcfg, rcfg, **kw):
        gd = GaussianDiffusion(steps=S, betas=betas, objective="noise", loss_type="l2", device="cuda", cfg_dropout_proba=0.0,
                               embedding_scale=scale, batch_cfg=bcfg, scale_cfg=rcfg)
        assert not gd.is_ddim_sampling
        y = gd.sample(m, shape, cond, init_noise=init, step_noises=noises, use_graph=use_graph, **kw)
        torch.cuda.synchronize()
        return y.cpu().numpy()

    # 20 chained steps with clamping: 1e-3 (north_star's gate)
    assert rel_err(run(0.8, True, True), g["ddpm20.cfg"]) < 1e-3
    assert rel_err(run(1.0, False, False)[:, :, ::3], g["ddpm20.nocfg"]) < 1e-3
    traj = run(0.8, True, True, return_all_timesteps=True)
    assert traj.shape == (B, S + 1, 128, T)
    assert rel_err(traj[:, :, ::8, ::15], g["ddpm20.cfg.traj"]) < 1e-3
    if not use_graph:
        assert rel_err(run(0.8, True, True, fused=False), g["ddpm20.cfg"]) < 1e-3


def test_tiny_training_loss_value_vs_golden(tiny_models):
    """forward value of ``training_loosses`` (gdm.py:245-272) for the three tasks x objectives."""
    from jen1_amd.diffusion import GaussianDiffusion, get_beta_schedule
    from jen1_amd.init_fill import fill_uniform
    g = golden("tiny_train")
    B, T = 2, 300
    betas, _ = get_beta_schedule("linear", 1000)
    t = torch.tensor([17, 801], dtype=torch.long, device="cuda")
    for task, causal in (("text_guided", False), ("music_inpaint", False), ("music_cont", True)):
        x0 = dev(synth.latents(B, T, key="clip"))
        cond = {k: dev(v) for k, v in synth.conditioning(B, T, task).items()}
        noise = dev(fill_uniform(f"synth.trainnoise.{task}", (B, 128, T), 3, 0.0, 1.0))
        for objective in ("noise", "x0", "v"):
            gd = GaussianDiffusion(steps=1000, betas=betas, objective=objective, loss_type="l2", device="cuda",
                                   cfg_dropout_proba=0.0, embedding_scale=0.8, batch_cfg=True, scale_cfg=True)
            loss = float(gd.training_loosses(tiny_models["f32"], x0, t, cond, noise=noise, causal=causal).detach())
            ref = float(g[f"loss.{task}.{objective}"])
            assert abs(loss - ref) <= 1e-3 * abs(ref), (task, objective, loss, ref)


# ------------------------------------------------------------------ full configuration (BASELINE configs[1], [2])
@pytest.fixture(scope="module")
def full_model_f32():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from jen1_amd.model import UNetCFG1d
    return UNetCFG1d(**full_model_config(), init_seed=1234, compute_dtype="f32", device="cuda")


def test_full_unet_vs_golden_f32(full_model_f32):
    g = golden("full_unet")
    B, T = 2, 1500
    x, cond = synth.latents(B, T), synth.conditioning(B, T)
    t = np.array([999, 9], dtype=np.int64)
    y = _fwd(full_model_f32, x, t, cond, embedding_scale=0.8, batch_cfg=True, scale_cfg=True, causal=False)
    e = rel_err(y[:, :, ::16], g["y.cfg"])
    print(f"full UNet fp32, CFG pair: max-abs/max-ref = {e:.3e}, max-abs err = {np.abs(y[:, :, ::16] - g['y.cfg']).max():.3e}")
    assert e < F32_TOL
    plan = full_model_f32.engine().plan(B, T, 2, False)
    for k in [k for k in g.files if k.startswith("tap.cfg.")]:
        a = plan.taps[k[len("tap.cfg."):]]
        ref = g[k]
        if a.L != int(ref[2]):
            # up-path taps: the engine stores the upsample output already centre-cropped to the skip
            # length (utils/module.py:186-204 folded into the producer); the reference tap is pre-crop
            assert k.startswith("tap.cfg.up") and 0 < int(ref[2]) - a.L < 4, k
            continue
        if k == "tap.cfg.up8":
            continue                      # `x += skips_list.pop()` (model.py:261) is fused into this tensor
        n = float(torch.linalg.vector_norm(a.t[:, :, : a.C].double()))
        assert abs(n - ref[0]) <= 1e-3 * ref[0], (k, n, ref[0])
    y = _fwd(full_model_f32, x, t, cond, embedding_scale=1.0, causal=True)
    e = rel_err(y[:, :, ::16], g["y.nocfg.causal"])
    print(f"full UNet fp32, no CFG, causal: max-abs/max-ref = {e:.3e}")
    assert e < F32_TOL


def test_full_unet_bf16_reported_error():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from jen1_amd.model import UNetCFG1d
    g = golden("full_unet")
    m = UNetCFG1d(**full_model_config(), init_seed=1234, compute_dtype="bf16", device="cuda")
    B, T = 2, 1500
    x, cond = synth.latents(B, T), synth.conditioning(B, T)
    t = np.array([999, 9], dtype=np.int64)
    y = _fwd(m, x, t, cond, embedding_scale=0.8, batch_cfg=True, scale_cfg=True, causal=False)
    e = rel_err(y[:, :, ::16], g["y.cfg"])
    print(f"full UNet bf16, CFG pair: max-abs/max-ref = {e:.3e}")
    assert e < BF16_TOL


@pytest.fixture(scope="module")
def full_model_bf16():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from jen1_amd.model import UNetCFG1d
    return UNetCFG1d(**full_model_config(), init_seed=1234, compute_dtype="bf16", device="cuda")


def _check_taps(plan, g, prefix):
    for k in [k for k in g.files if k.startswith(prefix)]:
        name = k[len(prefix):]
        a, ref = plan.taps[name], g[k]
        if a.L != int(ref[2]) or name == "up8":
            continue      # up-path taps are stored already cropped / with the final skip add fused (see test_full_unet_vs_golden_f32)
        n = float(torch.linalg.vector_norm(a.t[:, :, : a.C].double()))
        assert abs(n - ref[0]) <= 1e-3 * ref[0], (k, n, ref[0])


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_full_unet_bench_shapes_B8_vs_golden(full_model_f32, full_model_bf16, mode):
    """BASELINE configs[1] / [2]: the B = 8 (no CFG) and 2B = 16 (CFG pair) launch plans bench.py times, against what the
    reference produced on the same inputs (tests/golden/full_bench.npz: B=8, T=1500, eight different timesteps)."""
    g = golden("full_bench")
    m, tol = (full_model_f32, F32_TOL) if mode == "f32" else (full_model_bf16, BF16_TOL)
    B, T = 8, 1500
    x, cond = synth.latents(B, T), synth.conditioning(B, T)
    t = g["B8.t"]
    y = _fwd(m, x, t, cond, embedding_scale=1.0, causal=False)
    e = rel_err(y[:, :, ::16], g["B8.y.nocfg"])
    print(f"full UNet {mode}, B=8 no CFG: max-abs/max-ref = {e:.3e}")
    assert e < tol
    if mode == "f32":
        _check_taps(m.engine().plan(B, T, 1, False), g, "B8.tap.nocfg.")
    y = _fwd(m, x, t, cond, embedding_scale=0.8, batch_cfg=True, scale_cfg=True, causal=False)
    e = rel_err(y[:, :, ::16], g["B8.y.cfg"])
    print(f"full UNet {mode}, B=8 CFG pair (2B=16): max-abs/max-ref = {e:.3e}")
    assert e < tol
    if mode == "f32":
        _check_taps(m.engine().plan(B, T, 2, False), g, "B8.tap.cfg.")


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_full_unet_long_sequence_T9000_vs_golden(full_model_f32, full_model_bf16, mode):
    """BASELINE configs[4] shape (B=1, 128x9000: self-attention over 141 positions, 6x the conv work), continuation
    (causal) and inpainting (non-causal) conditioning through the CFG pair, against the reference's output."""
    g = golden("full_bench")
    m, tol = (full_model_f32, F32_TOL) if mode == "f32" else (full_model_bf16, BF16_TOL)
    B, T = 1, 9000
    t = np.array([499], dtype=np.int64)
    for task, causal in (("music_cont", True), ("music_inpaint", False)):
        x, cond = synth.latents(B, T), synth.conditioning(B, T, task)
        y = _fwd(m, x, t, cond, embedding_scale=0.8, batch_cfg=True, scale_cfg=True, causal=causal)
        assert y.shape == (B, 128, T)
        e = rel_err(y[:, :, ::24], g[f"T9000.y.{task}"])
        print(f"full UNet {mode}, T=9000 {task}: max-abs/max-ref = {e:.3e}")
        assert e < tol, (task, e)
        if mode == "f32":
            _check_taps(m.engine().plan(B, T, 2, causal), g, f"T9000.tap.{task}.")


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_full_ddim_vs_golden(full_model_f32, full_model_bf16, mode):
    """The sampler at the full configuration against the reference's own ``GaussianDiffusion.sample`` with injected noise:
    10 DDIM steps at B=2, and 2 steps at the two bench shapes (B=8 with and without the CFG pair) and at T=9000 (causal),
    every one as a replayed hipGraph -- the same stepper objects and plans bench.py times."""
    from jen1_amd.diffusion import GaussianDiffusion, get_beta_schedule
    g = golden("full_bench")
    m, tol = (full_model_f32, F32_TOL) if mode == "f32" else (full_model_bf16, BF16_TOL)
    betas, _ = get_beta_schedule("linear", 1000)
    for key, B, T, S, scale, causal, task, sub in (("ddim10.B2.cfg", 2, 1500, 10, 0.8, False, "text_guided", 8),
                                                  ("ddim2.B8.cfg", 8, 1500, 2, 0.8, False, "text_guided", 16),
                                                  ("ddim2.B8.nocfg", 8, 1500, 2, 1.0, False, "text_guided", 16),
                                                  ("ddim2.T9000.cont", 1, 9000, 2, 0.8, True, "music_cont", 24)):
        cond = {k: dev(v) for k, v in synth.conditioning(B, T, task).items()}
        shape = (B, 128, T)
        init = dev(synth.noise_list(1, shape, seed=7)[0])
        noises = [dev(n) for n in synth.noise_list(S, shape, seed=11)]
        gd = GaussianDiffusion(steps=1000, betas=betas, objective="noise", loss_type="l2", device="cuda", cfg_dropout_proba=0.0,
                               embedding_scale=scale, batch_cfg=True, scale_cfg=True, sampling_timesteps=S)
        y = gd.sample(m, shape, cond, causal=causal, init_noise=init, step_noises=noises, use_graph=True)
        torch.cuda.synchronize()
        got, ref = y.cpu().numpy()[:, :, ::sub].astype(np.float64), g[key].astype(np.float64)
        e = rel_err(got, ref)
        l2 = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
        print(f"{key} {mode}: max-abs/max-ref = {e:.3e}, relative L2 = {l2:.3e}")
        record_parity("full_ddim", key, mode, max_abs_rel=e, l2=l2)
        # Conditioning: the first step at t = 999 forms x0 = 157 * (x - 0.99998 * eps), clamped to [-1, 1].  Measured on the
        # reference itself (CPU, float32): a 1e-6 RELATIVE change of the initial noise moves its own 2-step output by 5.9e-4
        # and its 10-step output by 2.8e-5 (max-abs / max-ref).  A single forward of this build agrees with the reference to
        # 1e-6 in float32 (tests above), so float32 is gated at 5e-3 for the 2-step cases and at the 1e-3 parity gate for the
        # 10-step case.  bf16 storage (7e-3 per forward) is amplified the same way: single entries near a clamp boundary move by
        # tenths of the [-1, 1] range, so bf16 is gated on the relative L2 error of the whole sample (<= 1e-1).
        if mode == "f32":
            assert e < (tol if S >= 10 else 5e-3), (key, e)
        else:
            assert l2 < bf16_gate("full_ddim", key, "l2"), (key, l2, e)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_full_ddim100_vs_golden(full_model_f32, full_model_bf16, mode):
    """BASELINE configs[1] / [2] at their own schedule length: 100 DDIM steps of the full model as a replayed hipGraph against
    the reference's ``GaussianDiffusion.sample`` (tests/golden/make_golden.py ddim100): B = 2 through the CFG pair with eta = 1
    and injected per-step noise, and B = 8 without CFG at eta = 0 -- the exact plan bench.py times.  The trajectory is compared
    at steps 10 / 25 / 50 / 75 and at the end.  The reference's own response to a 1e-6 relative change of the start noise is
    stored next to every tap (<= 1.8e-4 of the value range at the end, <= 1.5e-5 up to step 25), so float32 is held to the
    1e-3 parity gate everywhere; bf16 storage is gated on the relative L2 error of the whole sample."""
    from jen1_amd.diffusion import GaussianDiffusion, get_beta_schedule
    g = golden("full_ddim100")
    m = full_model_f32 if mode == "f32" else full_model_bf16
    betas, _ = get_beta_schedule("linear", 1000)
    for key, B, scale, eta, sub in (("ddim100.B2.cfg", 2, 0.8, 1.0, 8), ("ddim100.B8.nocfg", 8, 1.0, 0.0, 16)):
        T, S = 1500, 100
        cond = {k: dev(v) for k, v in synth.conditioning(B, T).items()}
        shape = (B, 128, T)
        init = dev(synth.noise_list(1, shape, seed=7)[0])
        noises = [dev(n) for n in synth.noise_list(S, shape, seed=11)] if eta else None
        gd = GaussianDiffusion(steps=1000, betas=betas, objective="noise", loss_type="l2", device="cuda", cfg_dropout_proba=0.0,
                               embedding_scale=scale, batch_cfg=True, scale_cfg=True, sampling_timesteps=S, ddim_sampling_eta=eta)
        traj = gd.sample(m, shape, cond, return_all_timesteps=True, init_noise=init, step_noises=noises, use_graph=True)
        assert traj.shape == (B, S + 1, 128, T)
        fin = gd.sample(m, shape, cond, init_noise=init, step_noises=noises, use_graph=True)
        torch.cuda.synchronize()
        for name, got in [(f"{key}.step{k}", traj[:, k]) for k in (10, 25, 50, 75)] + [(key, fin)]:
            got, ref = got.cpu().numpy()[:, :, ::sub].astype(np.float64), g[name].astype(np.float64)
            e = rel_err(got, ref)
            l2 = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
            sens = g[name.replace(key, key + ".sens")]
            print(f"{name} {mode}: max-abs/max-ref = {e:.3e}, relative L2 = {l2:.3e}   (reference's own 1e-6 sensitivity: {sens[0] / sens[1]:.1e})")
            record_parity("full_ddim100", name, mode, max_abs_rel=e, l2=l2)
            if mode == "f32":
                assert e < F32_TOL, (name, e)
            else:
                assert l2 < bf16_gate("full_ddim100", name, "l2"), (name, l2, e)


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_vdm_sampler_vs_repaired_reference(tiny_models, mode):
    """``VDM`` (jen1_amd/vdm.py: the reference's vdm.py with the A-3 / A-4 repairs) on the fused stepper (row kind 3 of
    jen1_cfg_ddim_step, continuous times through jen1_time_features_f32), eager and as a replayed graph, and through the literal
    loop over ``model(...)``, against the fixture made from the reference's own class with the same repairs."""
    from jen1_amd.vdm import VDM
    g = golden("tiny_vdm")
    m, tol = (tiny_models["f32"], F32_TOL) if mode == "f32" else (tiny_models["bf16"], BF16_TOL)
    B, T = 2, 300
    cond = {k: dev(v) for k, v in synth.conditioning(B, T).items()}
    shape = (B, 128, T)
    init = dev(synth.noise_list(1, shape, seed=23)[0])
    vd = VDM(loss_type="l2", device="cuda", cfg_dropout_proba=0.0, embedding_scale=0.8, batch_cfg=True, scale_cfg=True)
    for kw in (dict(use_graph=True), dict(use_graph=False), dict(fused=False)):
        traj = vd.sample(m, shape, cond, step=10, return_all_timesteps=True, init_noise=init, **kw)
        torch.cuda.synchronize()
        assert traj.shape == (B, 11, 128, T)
        e = rel_err(traj[:, -1].cpu().numpy(), g["vdm10.cfg"])
        et = rel_err(traj.cpu().numpy()[:, :, ::8, ::15], g["vdm10.cfg.traj"])
        print(f"VDM 10 steps {mode} {kw}: final {e:.3e}, trajectory {et:.3e}")
        assert e < tol and et < tol, (kw, e, et)
    vd1 = VDM(loss_type="l2", device="cuda", cfg_dropout_proba=0.0, embedding_scale=1.0, batch_cfg=True, scale_cfg=True)
    y = vd1.sample(m, shape, cond, step=4, causal=True, init_noise=init)
    assert rel_err(y.cpu().numpy()[:, :, ::3], g["vdm4.nocfg.causal"]) < tol


def test_sampler_full_size_properties(full_model_f32):
    """size-independent properties at BASELINE size (B=2 to stay in memory/time): every x0 prediction is
    clamped to [-1, 1] so the final DDIM sample is; graph replay == eager; finite everywhere."""
    from jen1_amd.diffusion import GaussianDiffusion, get_beta_schedule
    B, T, S = 2, 1500, 4
    cond = {k: dev(v) for k, v in synth.conditioning(B, T).items()}
    betas, _ = get_beta_schedule("linear", 1000)
    shape = (B, 128, T)
    init = dev(synth.noise_list(1, shape, seed=7)[0])
    noises = [dev(n) for n in synth.noise_list(S, shape, seed=11)]
    outs = []
    for use_graph in (False, True):
        gd = GaussianDiffusion(steps=1000, betas=betas, objective="noise", loss_type="l2", device="cuda",
                               cfg_dropout_proba=0.0, embedding_scale=0.8, batch_cfg=True, scale_cfg=True, sampling_timesteps=S)
        y = gd.sample(full_model_f32, shape, cond, init_noise=init, step_noises=noises, use_graph=use_graph)
        torch.cuda.synchronize()
        assert torch.isfinite(y).all()
        assert float(y.abs().max()) <= 1.0 + 1e-6
        outs.append(y.cpu().numpy())
    # graph replay vs eager: same kernels, same order.  A single forward repeats to ~4e-7 (float atomics reorder
    # the GroupNorm/LayerNorm sums of the launch-per-layer levels); four steps from t=999, where x0 = 157*(...) is clamped,
    # amplify that to ~1e-4 run to run (eager vs eager shows the same spread), so the check uses the 1e-3 parity gate.
    # With fixed-order statistics the two are bit-identical: test_deterministic_statistics_mode_is_bit_reproducible.
    assert rel_err(outs[1], outs[0]) < F32_TOL


def test_deterministic_statistics_mode_is_bit_reproducible(full_model_f32, tiny_models):
    """Plan(deterministic=True): the GroupNorm / LayerNorm sums of the levels that run one launch per layer come from fixed-order statistics
    launches (jen1_gn_stats, jen1_row_stats) instead of float atomics.  Two 100-step DDIM runs of the full model from the same
    inputs are bit-identical (graph replay and eager), and the mode changes nothing beyond the summation order (1e-5 against the
    default plan after one forward).  The tiny configuration covers attention on the launch path (row statistics)."""
    from jen1_amd.diffusion import GaussianDiffusion, get_beta_schedule
    betas, _ = get_beta_schedule("linear", 1000)
    for m, (B, T, S) in ((full_model_f32, (2, 1500, 100)), (tiny_models["f32"], (2, 300, 20))):
        cond = {k: dev(v) for k, v in synth.conditioning(B, T).items()}
        shape = (B, 128, T)
        init = dev(synth.noise_list(1, shape, seed=7)[0])
        x = dev(synth.latents(B, T))
        t = torch.tensor([999, 499], device="cuda")
        kw = dict(embedding=cond["cross_attn_cond"], embedding_mask=cond["cross_attn_masks"], channels_list=[cond["input_concat_cond"]],
                  embedding_scale=0.8, batch_cfg=True, scale_cfg=True)
        y_default = m(x, t, **kw).clone()
        m.deterministic = True
        try:
            y_det = [m(x, t, **kw).clone() for _ in range(2)]
            assert torch.equal(y_det[0], y_det[1])
            assert rel_err(y_det[0].cpu().numpy(), y_default.cpu().numpy()) < 1e-5
            plan = m.engine().plan(B, T, 2, False)
            # (the full model's long levels run as sample-resident launches with fixed-order statistics of their own and the pack kernel's
            # sums are fixed-order by default: no statistics launch is left to add there; the tiny configuration runs launch per layer)
            assert plan.det and (plan.long_levels >= 1 or any(getattr(op, "kind", "") == "stats" for op in plan.ops))
            outs = []
            for use_graph in (True, False, True):
                gd = GaussianDiffusion(steps=1000, betas=betas, objective="noise", loss_type="l2", device="cuda", cfg_dropout_proba=0.0,
                                       embedding_scale=0.8, batch_cfg=True, scale_cfg=True, sampling_timesteps=S, ddim_sampling_eta=0.0)
                y = gd.sample(m, shape, cond, init_noise=init, use_graph=use_graph)
                torch.cuda.synchronize()
                assert torch.isfinite(y).all()
                outs.append(y.clone())
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
        finally:
            m.deterministic = False


def test_text_conditioner_tail_projection_and_mask():
    """T5Conditioner.forward tail: proj_out(last_hidden_state) * attention_mask (conditioners.py:106-111) on the HIP path."""
    from jen1_amd.tasks import TextConditionerTail
    torch.manual_seed(3)
    B, N, F = 3, 128, 1024
    w, b = torch.randn(F, F) / F ** 0.5, torch.randn(F) * 0.1
    hid = torch.randn(B, N, F, device="cuda")
    mask = (torch.arange(N, device="cuda")[None, :] < torch.tensor([5, 77, 128], device="cuda")[:, None])
    ref = (hid @ w.cuda().T + b.cuda()) * mask[..., None].float()
    for dtype, tol in (("f32", F32_TOL), ("bf16", BF16_TOL)):
        tail = TextConditionerTail(w, b, dtype=dtype)
        emb, m = tail(hid, mask)
        assert m is mask and emb.dtype == torch.float32
        assert rel_err(emb.cpu().numpy(), ref.cpu().numpy()) < tol
        assert float(emb[0, 5:].abs().max()) == 0.0


def test_independent_batches_in_flight_match_solo_runs(tiny_models):
    """serving mode (bench.py extra.concurrent_batches): samplers of the same shape with different ``plan_slot`` own separate
    buffers, so several batches can be in flight on different streams; each must end exactly where it ends when run alone"""
    from jen1_amd.diffusion import GaussianDiffusion, get_beta_schedule
    B, T, S = 2, 300, 6
    betas, _ = get_beta_schedule("linear", 1000)
    m = tiny_models["f32"]
    # fixed-order statistics: a trajectory is then bit-reproducible, so "the same as alone" can be checked exactly (with the default
    # float atomics six steps from t = 999 amplify the summation-order noise to ~1e-4 and the comparison would be a coin toss)
    m.deterministic = True
    gd = GaussianDiffusion(steps=1000, betas=betas, objective="noise", loss_type="l2", device="cuda", cfg_dropout_proba=0.0,
                           embedding_scale=0.8, batch_cfg=True, scale_cfg=True, sampling_timesteps=S)
    conds = [{k: dev(v) for k, v in synth.conditioning(B, T, task).items()} for task in ("text_guided", "music_inpaint", "music_cont")]
    inits = [dev(n) for n in synth.noise_list(3, (B, 128, T), seed=21)]
    noises = [dev(n) for n in synth.noise_list(S, (B, 128, T), seed=22)]

    def solo(i):
        st = gd.stepper(m, (B, 128, T), conds[i], causal=False, use_graph=True, plan_slot=10 + i)
        st.reset(inits[i], fresh_noise=False)
        for k in range(S):
            st.step(k, noise=noises[k])
        torch.cuda.synchronize()
        return st, st.x.clone()

    sts, want = zip(*[solo(i) for i in range(3)])
    assert sts[0].plan is not sts[1].plan and sts[0].plan.x_in.data_ptr() != sts[1].plan.x_in.data_ptr()
    streams = [torch.cuda.Stream() for _ in range(3)]
    for i, st in enumerate(sts):
        st.reset(inits[i], fresh_noise=False)
    torch.cuda.synchronize()
    for k in range(S):                                   # interleave the three trajectories step by step on three streams
        for st, s in zip(sts, streams):
            with torch.cuda.stream(s):
                st.step(k, noise=noises[k])
    torch.cuda.synchronize()
    m.deterministic = False
    for i, st in enumerate(sts):
        assert torch.equal(st.x, want[i]), (i, rel_err(st.x.cpu().numpy(), want[i].cpu().numpy()))
    assert rel_err(want[0].cpu().numpy(), want[1].cpu().numpy()) > 1e-2      # the three really are different trajectories


def test_context_of_another_shape_is_refused_not_read_out_of_range(tiny_models):
    """``Plan.set_context`` hands the caller's embedding pointer to a launch that reads B x max_length x features floats: a shorter token
    axis, another batch or a mask of another shape must raise (ValueError), as the ``copy_`` of earlier rounds did -- never a silent
    out-of-range device read.  The conditioner pads to max_length (conditioners.py:84-111), so this is the contract."""
    m = tiny_models["f32"]
    x, t, cond = _inputs()
    good = _fwd(m, x, t, cond)
    emb, msk = dev(cond["cross_attn_cond"]), dev(cond["cross_attn_masks"])
    kw = dict(embedding_scale=1.0, channels_list=[dev(cond["input_concat_cond"])], causal=False)
    with pytest.raises(ValueError, match="context_embedding_max_length"):
        m(dev(x), dev(t), embedding=emb[:, :64].contiguous(), embedding_mask=msk[:, :64].contiguous(), **kw)
    with pytest.raises(ValueError):
        m(dev(x), dev(t), embedding=emb[:1].contiguous(), embedding_mask=msk[:1].contiguous(), **kw)
    with pytest.raises(ValueError, match="embedding_mask"):
        m(dev(x), dev(t), embedding=emb, embedding_mask=msk[:, :64].contiguous(), **kw)
    again = _fwd(m, x, t, cond)
    assert rel_err(again, good) < 1e-5
