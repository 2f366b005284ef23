"""-m gpu: the persistent deep-level kernel (include/jen1_deep.h, csrc/deep_kernel.hip) against the launch-per-layer path
and the oracle.

The levels with few positions (T' <= 24 at T = 1500: reference jen1/model/model.py:246-259, blocks.py:540-830) run as ONE
launch whose workgroups exchange activations through global memory inside the launch.  Besides the golden / oracle parity of
the whole model (tests/test_gpu_model.py runs the same plans), these tests compare EVERY intermediate activation of the two
execution paths, replay the launch many times (a stale read anywhere in the exchange shows up as a changed bit: the launch
is bit-reproducible by construction) and check the error word of the bounded spins.
"""
import numpy as np
import pytest
import torch

from helpers import filled, rel_err
from jen1_amd import synth
from jen1_amd.config import UNetSpec, full_model_config, tiny_model_config

pytestmark = pytest.mark.gpu


def dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def tiny_f32():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from jen1_amd.model import UNetCFG1d
    return UNetCFG1d(**tiny_model_config(), init_seed=1234, compute_dtype="f32", device="cuda")


@pytest.fixture(scope="module")
def tiny_bf16():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from jen1_amd.model import UNetCFG1d
    return UNetCFG1d(**tiny_model_config(), init_seed=1234, compute_dtype="bf16", device="cuda")


@pytest.fixture(scope="module")
def full_f32():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from jen1_amd.model import UNetCFG1d
    return UNetCFG1d(**full_model_config(), init_seed=1234, compute_dtype="f32", device="cuda")


@pytest.fixture(scope="module")
def full_bf16():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from jen1_amd.model import UNetCFG1d
    return UNetCFG1d(**full_model_config(), init_seed=1234, compute_dtype="bf16", device="cuda")


@pytest.fixture(scope="module")
def full_fp8():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from jen1_amd.model import UNetCFG1d
    return UNetCFG1d(**full_model_config(), init_seed=1234, compute_dtype="fp8", device="cuda")


def run_plan(model, plan, x, t, cond, drop=None):
    s = torch.cuda.current_stream().cuda_stream
    model._prepare(plan, dev(x), dev(t), dev(cond["cross_attn_cond"]), dev(cond["cross_attn_masks"]), [dev(cond["input_concat_cond"])], drop)
    plan.run(s)
    torch.cuda.synchronize()


def compare_paths(model, B, T, nrep, causal, task="music_cont", tol=2e-5):
    """run the same forward on the plan with the persistent launch and on the launch-per-layer plan; compare every activation"""
    eng = model.engine()
    pd = eng.plan(B, T, nrep, causal, deep=True)
    pl = eng.plan(B, T, nrep, causal, deep=False)
    assert pl.deep_level is None
    x, cond = synth.latents(B, T), synth.conditioning(B, T, task)
    t = np.array([(131 * i + 7) % 1000 for i in range(B)], dtype=np.int64)
    run_plan(model, pl, x, t, cond)
    run_plan(model, pd, x, t, cond)
    if pd.deep_level is None:
        return pd, None
    assert pd.deep.error() == 0, "a dependency wait of the persistent launch timed out"
    worst = 0.0
    pairs = list(zip(pd.acts, pl.acts))
    if len(pd.acts) != len(pl.acts):
        # the launch-per-layer plan took another fusion decision somewhere (e.g. its tile kernel at >= 512 rows keeps the 1x1
        # shortcut as a launch of its own): the activation lists do not line up, compare the per-level outputs instead
        pairs = [(pd.taps[k], pl.taps[k]) for k in pd.taps]
    for i, (a, b) in enumerate(pairs):
        assert a.t.shape == b.t.shape, (i, a.t.shape, b.t.shape)
        ra, rb = a.t[:, :, : a.C].float(), b.t[:, :, : b.C].float()
        if not torch.isfinite(rb).all():
            continue                                   # buffers the launch path allocates but never writes
        den = float(rb.abs().max())
        if den == 0.0:
            continue
        e = float((ra - rb).abs().max()) / den
        worst = max(worst, e)
        assert e < tol, f"activation {i} of {len(pairs)} (shape {tuple(a.t.shape)}): max-abs/max-ref {e:.3e}; deep phases: {pd.deep.labels[:4]} ..."
    return pd, worst


@pytest.mark.parametrize("B,T,nrep,causal", [(2, 64, 1, False), (3, 95, 2, True)])
def test_tiny_config_falls_back_to_launch_path(tiny_f32, B, T, nrep, causal):
    """the tiny configuration has GroupNorm groups of 2 / 4 channels (narrower than the 8-channel vectors the persistent kernel
    stages): every level is refused with a message and the plan keeps one launch per layer"""
    pd, worst = compare_paths(tiny_f32, B, T, nrep, causal)
    assert pd.deep_level is None and pd.deep_errors and all("groups" in e for e in pd.deep_errors), pd.deep_errors


@pytest.mark.parametrize("B,T", [(1, 64), (3, 95), (4, 90)])
def test_tiny_deep_vs_oracle(tiny_f32, tiny_bf16, B, T):
    from oracle import jen1_oracle as O
    cfg = tiny_model_config()
    onet = O.OracleUNetCFG1d(filled(UNetSpec(**cfg).param_shapes()), **cfg)
    x, cond = synth.latents(B, T), synth.conditioning(B, T, "music_inpaint")
    t = np.array([(37 * i + 5) % 1000 for i in range(B)], dtype=np.int64)
    ref = onet(x, t, embedding=cond["cross_attn_cond"], embedding_mask=cond["cross_attn_masks"], embedding_scale=0.8, batch_cfg=True,
               scale_cfg=True, channels_list=[cond["input_concat_cond"]], causal=False)
    for m, tol in ((tiny_f32, 1e-3), (tiny_bf16, 5e-2)):
        y = m(dev(x), dev(t), embedding=dev(cond["cross_attn_cond"]), embedding_mask=dev(cond["cross_attn_masks"]), embedding_scale=0.8,
              batch_cfg=True, scale_cfg=True, channels_list=[dev(cond["input_concat_cond"])], causal=False)
        torch.cuda.synchronize()
        assert rel_err(y.cpu().numpy(), ref) < tol


@pytest.mark.parametrize("B,T,nrep,causal", [(2, 1500, 2, False), (8, 1500, 1, False), (8, 1500, 2, True), (1, 9000, 2, True)])
def test_full_deep_equals_launch_path_f32(full_f32, B, T, nrep, causal):
    pd, worst = compare_paths(full_f32, B, T, nrep, causal, task="music_cont" if causal else "text_guided")
    assert pd.deep_level is not None, pd.deep_errors
    print(f"full B={B} T={T} nrep={nrep} causal={causal}: deep from level {pd.deep_level}, {len(pd.deep)} phases, {pd.n_launch} launches, "
          f"worst activation difference {worst:.2e}")
    if T == 1500:
        assert pd.deep_level == 3
        assert pd.n_launch <= 120


def test_full_deep_bf16_close_to_launch_path(full_bf16):
    pd, worst = compare_paths(full_bf16, 8, 1500, 1, False, task="text_guided", tol=6e-2)
    assert pd.deep_level == 3
    print(f"full bf16 B=8: worst activation difference between the two paths {worst:.2e}")


def test_deep_launch_replays_bit_identically(full_f32, full_bf16):
    """staleness stress: the persistent launch alone, 200 replays on fixed inputs, with a bandwidth hog on a second stream
    during half of them (uneven load); every output bit of the chain's last tensor and of two tensors in the middle must
    repeat -- the launch has no atomics on data, so any difference is a stale or torn read in the exchange"""
    for model in (full_f32, full_bf16):
        B, T = 8, 1500
        plan = model.engine().plan(B, T, 1, False, deep=True)
        x, cond = synth.latents(B, T), synth.conditioning(B, T)
        t = np.array([999, 989, 499, 259, 129, 59, 9, 0], dtype=np.int64)
        run_plan(model, plan, x, t, cond)
        prog = plan.deep
        assert prog.error() == 0
        outs = [a for a in prog.outs if a is not None]
        watch = [outs[-1], outs[len(outs) // 2], outs[len(outs) // 5]]
        want = [a.t.clone() for a in watch]
        s = torch.cuda.current_stream().cuda_stream
        hog_stream = torch.cuda.Stream()
        hog = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
        for rep in range(200):
            for a in watch:
                a.t.fill_(float("nan"))                 # poison: a skipped store would show
            prog.poison(s)                            # every tensor of the launch back to the sentinel (a skipped store would hang a consumer -> error word)
            if rep % 2:
                with torch.cuda.stream(hog_stream):
                    hog.add_(1.0)
            prog.launch(s)
            torch.cuda.synchronize()
            assert prog.error() == 0
            for a, w in zip(watch, want):
                assert torch.equal(a.t, w), f"replay {rep}: output of the persistent launch changed"


def test_deep_sampler_graph_replay_matches_eager(tiny_f32):
    """the persistent launch inside a captured sampler step (hipGraph replay) ends where the eager steps end"""
    from jen1_amd.diffusion import GaussianDiffusion, get_beta_schedule
    B, T, S = 2, 64, 6
    betas, _ = get_beta_schedule("linear", 1000)
    cond = {k: dev(v) for k, v in synth.conditioning(B, T).items()}
    init = dev(synth.noise_list(1, (B, 128, T), seed=7)[0])
    noises = [dev(n) for n in synth.noise_list(S, (B, 128, T), seed=11)]
    outs = []
    for use_graph in (False, True):
        gd = GaussianDiffusion(steps=1000, betas=betas, objective="noise", loss_type="l2", device="cuda", cfg_dropout_proba=0.0,
                               embedding_scale=0.8, batch_cfg=True, scale_cfg=True, sampling_timesteps=S)
        y = gd.sample(tiny_f32, (B, 128, T), cond, init_noise=init, step_noises=noises, use_graph=use_graph)
        torch.cuda.synchronize()
        outs.append(y.cpu().numpy())
    assert rel_err(outs[1], outs[0]) < 1e-3


# ---------------------------------------------------------------------------------------------------------------------------
# JEN1_FP8 (BASELINE configs[4] "fp8 MFMA attention path"): OCP e4m3 operands on the matrix cores of the persistent launch
# ---------------------------------------------------------------------------------------------------------------------------
# Stated tolerance of the JEN1_FP8 mode, denoiser output against the reference's float32 output.  Measured at T = 9000: 6.9e-3
# max-abs / max-ref and 5.7e-3 relative L2 -- the same as bf16 (6.8e-3 / 5.7e-3): the e4m3 operands live in the levels with few
# positions, whose contribution to the output is small; INSIDE those levels single activations differ from the bf16 path by up to
# 0.3 of their range (test_fp8_deep_activations_track_the_bf16_path), so the gate leaves room for other weights / inputs.
FP8_TOL = 3e-2
FP8_L2 = 3e-2


def _golden(name):
    from helpers import golden
    return golden(name)


def test_fp8_weights_are_e4m3_with_row_scales(full_fp8):
    """packing: the e4m3 bytes times the row scale reproduce the bf16 weight to e4m3 precision (2^-4 relative per element), the
    row maximum maps to 448 exactly, and only layers inside the persistent launch own an fp8 copy"""
    eng = full_fp8.engine()
    plan = eng.plan(1, 9000, 2, True)
    assert plan.deep_level is not None and eng.deep_dt == 2 and eng.tdtype == torch.bfloat16
    W = eng.W
    assert len(W._fp8) > 50
    w = W.w["bottleneck.pre_block.conv1"]
    q, sc = W.fp8(w)
    assert q.dtype == torch.uint8 and q.numel() == w.numel() and sc.numel() == w.shape[-3] * 16
    MT = w.shape[-3]
    deq = q.view(torch.float8_e4m3fn).float().reshape(-1, MT, 4, 16, 8) * sc.reshape(MT, 16)[None, :, None, :, None]
    ref = w.float().reshape(-1, MT, 4, 16, 8)
    rowmax = ref.abs().amax(dim=(0, 2, 4))
    assert float(((deq - ref).abs() / rowmax[None, :, None, :, None]).max()) <= 2.0 ** -4
    assert float(q.view(torch.float8_e4m3fn).float().abs().amax()) == 448.0
    assert not any(id(W.w[k]) in W._fp8 for k in W.w if k.startswith("to_in.") or k.startswith("downsamples.0."))


@pytest.mark.parametrize("B,T,nrep,causal", [(8, 1500, 1, False), (1, 9000, 2, True)])
def test_fp8_deep_activations_track_the_bf16_path(full_fp8, B, T, nrep, causal):
    """every activation of the fp8 persistent launch against the launch-per-layer plan of the same engine (bf16 everywhere):
    e4m3 operands cost ~3 % per GEMM; the error must stay at that level through the whole chain (a wrong fragment layout, scale
    or P scaling would be O(1))"""
    # (gate per activation: the worst one -- a [B, 2, 1024] tensor of the deepest level -- sits at 0.33 .. 0.37 from run to run, the
    # bf16 side's GroupNorm statistics are float atomics; a wrong layout or scale gives > 1)
    pd, worst = compare_paths(full_fp8, B, T, nrep, causal, task="music_cont" if causal else "text_guided", tol=5e-1)
    assert pd.deep_level is not None, pd.deep_errors
    print(f"fp8 B={B} T={T}: deep from level {pd.deep_level}, {len(pd.deep)} phases, worst activation difference to the bf16 path {worst:.2e}")


def test_fp8_long_form_vs_golden(full_fp8, full_bf16):
    """BASELINE configs[4] as named: B = 1, T = 9000, CFG pair, continuation (causal) and inpainting, in JEN1_FP8 mode against the
    reference's float32 output (tests/golden/full_bench.npz); bf16 is printed next to it.  Stated tolerance: FP8_TOL / FP8_L2."""
    g = _golden("full_bench")
    B, T = 1, 9000
    t = np.array([499], dtype=np.int64)
    for task, causal in (("music_cont", True), ("music_inpaint", False)):
        x, cond = synth.latents(B, T), synth.conditioning(B, T, task)
        res = {}
        for name, m in (("fp8", full_fp8), ("bf16", full_bf16)):
            y = m(dev(x), dev(t), embedding=dev(cond["cross_attn_cond"]), embedding_mask=dev(cond["cross_attn_masks"]), embedding_scale=0.8,
                  batch_cfg=True, scale_cfg=True, channels_list=[dev(cond["input_concat_cond"])], causal=causal)
            torch.cuda.synchronize()
            got, ref = y.cpu().numpy()[:, :, ::24].astype(np.float64), g[f"T9000.y.{task}"].astype(np.float64)
            res[name] = (rel_err(got, ref), float(np.linalg.norm(got - ref) / np.linalg.norm(ref)))
        print(f"T=9000 {task}: fp8 max-abs/max-ref {res['fp8'][0]:.3e} rel-L2 {res['fp8'][1]:.3e} | bf16 {res['bf16'][0]:.3e} / {res['bf16'][1]:.3e}")
        assert res["fp8"][0] < FP8_TOL and res["fp8"][1] < FP8_L2, (task, res)
        plan = full_fp8.engine().plan(B, T, 2, causal)
        assert plan.deep_level is not None and plan.deep.error() == 0


def test_fp8_long_form_ddim_vs_golden(full_fp8):
    """the 2-step T = 9000 continuation DDIM golden (make_golden.py fullbench) through the fp8 plan as a replayed graph: gated on
    the relative L2 of the sample like bf16 (the x0 clamp turns single-entry errors into tenths of the range)"""
    from jen1_amd.diffusion import GaussianDiffusion, get_beta_schedule
    g = _golden("full_bench")
    betas, _ = get_beta_schedule("linear", 1000)
    B, T, S = 1, 9000, 2
    cond = {k: dev(v) for k, v in synth.conditioning(B, T, "music_cont").items()}
    shape = (B, 128, T)
    init = dev(synth.noise_list(1, shape, seed=7)[0])
    noises = [dev(n) for n in synth.noise_list(S, shape, seed=11)]
    gd = GaussianDiffusion(steps=1000, betas=betas, objective="noise", loss_type="l2", device="cuda", cfg_dropout_proba=0.0,
                           embedding_scale=0.8, batch_cfg=True, scale_cfg=True, sampling_timesteps=S)
    y = gd.sample(full_fp8, shape, cond, causal=True, init_noise=init, step_noises=noises, use_graph=True)
    torch.cuda.synchronize()
    got, ref = y.cpu().numpy()[:, :, ::24].astype(np.float64), g["ddim2.T9000.cont"].astype(np.float64)
    l2 = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
    print(f"ddim2.T9000.cont fp8: max-abs/max-ref {rel_err(got, ref):.3e}, relative L2 {l2:.3e}")
    assert l2 < 1e-1


def test_fp8_launch_replays_bit_identically(full_fp8):
    B, T = 1, 9000
    plan = full_fp8.engine().plan(B, T, 2, True, deep=True)
    x, cond = synth.latents(B, T), synth.conditioning(B, T, "music_cont")
    run_plan(full_fp8, plan, x, np.array([499], dtype=np.int64), cond)
    prog = plan.deep
    assert prog.error() == 0
    outs = [a for a in prog.outs if a is not None]
    watch = [outs[-1], outs[len(outs) // 2]]
    want = [a.t.clone() for a in watch]
    s = torch.cuda.current_stream().cuda_stream
    for rep in range(50):
        for a in watch:
            a.t.fill_(float("nan"))
        prog.poison(s)
        prog.launch(s)
        torch.cuda.synchronize()
        assert prog.error() == 0
        for a, w in zip(watch, want):
            assert torch.equal(a.t, w), f"replay {rep}: output of the fp8 persistent launch changed"



# ---------------------------------------------------------------------------------------------------------------------
# tile phases: the long levels inside persistent launches (JEN1_TILE_PHASES=1; include/jen1_deep.h JEN1_DEEP_TILE)
# ---------------------------------------------------------------------------------------------------------------------
def tile_plan(model, B, T, nrep, causal):
    """a plan whose layers over >= 200 positions are tile phases (built directly: the engine's plan cache is keyed without the knob)"""
    from jen1_amd.engine import Plan
    eng = model.engine()
    old = eng.use_tile_phases
    eng.use_tile_phases = True
    try:
        return Plan(eng, B, T, nrep, causal, None, deep=True)
    finally:
        eng.use_tile_phases = old


@pytest.mark.parametrize("B,T,nrep,causal", [(8, 1500, 1, False), (2, 1500, 2, True), (3, 1499, 1, False)])
def test_tile_phases_equal_launch_path_f32(full_f32, B, T, nrep, causal):
    """every activation of the plan with tile phases (levels of 1500 / 375 positions: two more persistent launches around the
    deep one) against the launch-per-layer plan; the statistics partials are summed in a fixed order, so two runs are bit-identical"""
    model = full_f32
    pt = tile_plan(model, B, T, nrep, causal)
    pl = model.engine().plan(B, T, nrep, causal, deep=False)
    assert len(pt.progs) == 3 and pt.progs[0].kinds.count("tile") >= 12 and pt.progs[2].kinds.count("tile") >= 10, [p.kinds for p in pt.progs]
    assert not pt.tile_errors, pt.tile_errors
    x, cond = synth.latents(B, T), synth.conditioning(B, T, "music_cont" if causal else "text_guided")
    t = np.array([(131 * i + 7) % 1000 for i in range(B)], dtype=np.int64)
    run_plan(model, pl, x, t, cond)
    run_plan(model, pt, x, t, cond)
    assert pt.take_error() == 0
    assert len(pt.acts) == len(pl.acts)
    worst = 0.0
    for i, (a, b) in enumerate(zip(pt.acts, pl.acts)):
        ra, rb = a.t[:, :, : a.C].float(), b.t[:, :, : b.C].float()
        if not torch.isfinite(rb).all() or float(rb.abs().max()) == 0.0:
            continue
        assert torch.isfinite(ra).all(), f"activation {i}: a sentinel / non-finite value survived"
        e = float((ra - rb).abs().max()) / float(rb.abs().max())
        worst = max(worst, e)
        assert e < 2e-5, f"activation {i} of {len(pt.acts)} (shape {tuple(a.t.shape)}): {e:.3e}"
    # the tile programs themselves are bit-reproducible (fixed-order partial sums; the rest of this plan -- the pack kernel's
    # statistics, the launches of the 94-position level -- still uses float atomics): each one alone, twice, on unchanged inputs
    s = torch.cuda.current_stream().cuda_stream
    for prog, out in ((pt.progs[0], pt.taps["down1"].t), (pt.progs[2], pt.net_out.t)):
        res = []
        for _ in range(2):
            prog.poison(s)
            prog.launch(s)
            torch.cuda.synchronize()
            res.append(out.clone())
        assert pt.take_error() == 0
        assert torch.isfinite(res[0].float()).all() and torch.equal(res[0], res[1]), "two runs of a tile program differ bitwise"
    print(f"tile phases B={B} T={T} nrep={nrep}: {pt.n_launch} launches (launch path {pl.n_launch}), worst activation difference {worst:.2e}")


def test_tile_phases_bf16_and_fp8_mode(full_bf16, full_fp8):
    """bf16: close to the launch path; JEN1_FP8 mode: the tile phases stay bf16 (same long-level activations as the bf16 model)"""
    B, T = 8, 1500
    x, cond = synth.latents(B, T), synth.conditioning(B, T, "text_guided")
    t = np.array([(131 * i + 7) % 1000 for i in range(B)], dtype=np.int64)
    pt = tile_plan(full_bf16, B, T, 1, False)
    pl = full_bf16.engine().plan(B, T, 1, False, deep=False)
    run_plan(full_bf16, pl, x, t, cond)
    run_plan(full_bf16, pt, x, t, cond)
    assert pt.take_error() == 0 and len(pt.progs) == 3
    for a, b in zip(pt.acts, pl.acts):
        ra, rb = a.t[:, :, : a.C].float(), b.t[:, :, : b.C].float()
        if not torch.isfinite(rb).all() or float(rb.abs().max()) == 0.0:
            continue
        assert float((ra - rb).abs().max()) / float(rb.abs().max()) < 6e-2
    p8 = tile_plan(full_fp8, B, T, 1, False)
    run_plan(full_fp8, p8, x, t, cond)
    assert p8.take_error() == 0 and len(p8.progs) == 3
    # program 0 (to_in, level 0 / 1 down) does not depend on the deep levels: the same bf16 computation in both modes (up to the
    # float atomics of the pack kernel's statistics)
    for k in ("to_in", "down0"):
        ra, rb = p8.taps[k].t.float(), pt.taps[k].t.float()
        assert float((ra - rb).abs().max()) / float(rb.abs().max()) < 2e-2, k
    ra, rb = p8.net_out.t.float(), pt.net_out.t.float()
    assert float((ra - rb).abs().max()) / float(rb.abs().max()) < 0.25


def test_column_chunked_level_equals_launch_path(full_f32, full_bf16):
    """JEN1_DEEP_MAX_LEN=96: the 94-position level of T = 1500 inside the persistent launch (units of <= 64 positions that stage the
    whole batch element; the block input of a second conv does not fit LDS there, so its 1x1 shortcut becomes a phase of its own).
    bf16: the level joins (188 phases); float32: its first conv does not fit and the plan starts one level deeper, as by default."""
    from jen1_amd.engine import Plan
    if not full_bf16.engine().lib.jen1_deep_has_chunks():
        pytest.skip("column chunks are a build option (-DJEN1_DEEP_CHUNKS; JEN1_LIB points at such a build): off in the default library")
    B, T = 8, 1500
    x, cond = synth.latents(B, T), synth.conditioning(B, T, "text_guided")
    t = np.array([(131 * i + 7) % 1000 for i in range(B)], dtype=np.int64)
    for model, tol, n_ph in ((full_bf16, 6e-2, 188), (full_f32, 2e-5, 171)):
        eng = model.engine()
        old = eng.deep_max_len
        eng.deep_max_len = 96
        try:
            pd = Plan(eng, B, T, 1, False, None, deep=True)
        finally:
            eng.deep_max_len = old
        pl = eng.plan(B, T, 1, False, deep=False)
        assert pd.deep_level is not None and len(pd.deep) == n_ph, (pd.deep_level, len(pd.deep), pd.deep_errors)
        run_plan(model, pl, x, t, cond)
        run_plan(model, pd, x, t, cond)
        assert pd.take_error() == 0
        for k in pd.taps:
            ra, rb = pd.taps[k].t[:, :, : pd.taps[k].C].float(), pl.taps[k].t[:, :, : pl.taps[k].C].float()
            assert torch.isfinite(ra).all()
            assert float((ra - rb).abs().max()) / float(rb.abs().max()) < tol, k
