"""-m gpu: the sample-resident long-level kernel (include/jen1_long.h, csrc/long_kernel.hip) against the launch-per-layer path.

to_in, the levels above the deep ones and to_out (reference jen1/model/model.py:243-262; blocks.py:98-145, :168-231, :540-650, :653-764)
run as two launches of 256 resident workgroups, sample b on workgroups b, b + B, ...; the golden / oracle parity of the whole model is
tests/test_gpu_model.py (the default plans use these launches).  Here EVERY intermediate activation of the two execution paths is
compared, the launches are replayed on unchanged inputs (fixed-order statistics: bit-reproducible), the ticket form is compared with
the static one, and configurations that do not fit fall back to one launch per layer.
"""
import numpy as np
import pytest
import torch

from jen1_amd import synth
from jen1_amd.config import full_model_config, tiny_model_config

pytestmark = pytest.mark.gpu


def dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


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


def run_plan(model, plan, x, t, cond, drop=None):
    s = torch.cuda.current_stream().cuda_stream
    model._prepare(plan, dev(x), dev(t), dev(cond["cross_attn_cond"]), dev(cond["cross_attn_masks"]), [dev(cond["input_concat_cond"])], drop)
    plan.run(s)
    torch.cuda.synchronize()


def make_plan(model, B, T, nrep, causal, long_levels: bool):
    """a plan with the persistent deep-level launch, with or without the two sample-resident long-level launches around it (built
    directly: the engine's plan cache is keyed without the knob)"""
    from jen1_amd.engine import Plan
    eng = model.engine()
    old = eng.use_long
    eng.use_long = long_levels
    try:
        return Plan(eng, B, T, nrep, causal, None, deep=True)
    finally:
        eng.use_long = old


def compare_acts(pa, pb, tol):
    """every activation of plan ``pa`` against its counterpart in ``pb`` (same creation order; ``pb`` may hold a few extra tensors -- the
    output of a 1x1 shortcut that ``pa`` fused into the block's second conv -- which are skipped: a tensor of ``pb`` that does not match is
    passed over as long as ``pb`` has tensors to spare)"""
    def err(a, b):
        ra, rb = a.t[:, :, : a.C].float(), b.t[:, :, : b.C].float()
        if not torch.isfinite(rb).all() or float(rb.abs().max()) == 0.0:
            return None
        assert torch.isfinite(ra).all(), "a sentinel / non-finite value survived"
        return float((ra - rb).abs().max()) / float(rb.abs().max())

    spare = len(pb.acts) - len(pa.acts)
    assert 0 <= spare <= 2, (len(pa.acts), len(pb.acts))
    worst, j = 0.0, 0
    for i, a in enumerate(pa.acts):
        while True:
            b = pb.acts[j]
            j += 1
            e = err(a, b) if tuple(a.t.shape) == tuple(b.t.shape) else 1.0
            if e is None or e < tol or spare == 0:
                break
            spare -= 1
        if e is not None:
            worst = max(worst, e)
            assert e < tol, f"activation {i} of {len(pa.acts)} (shape {tuple(a.t.shape)}): {e:.3e}"
    return worst


@pytest.mark.parametrize("B,T,nrep,causal", [(8, 1500, 1, False), (2, 1500, 2, True), (8, 1500, 2, False), (3, 1499, 1, False), (1, 9000, 2, True)],
                         ids=["B8", "B2-pair-causal", "B8-pair", "B3-T1499", "T9000-pair-causal"])
def test_long_levels_equal_launch_path_f32(full_f32, B, T, nrep, causal):
    """every activation of the plan with the two long-level launches against the plan that runs those levels one launch per layer
    (float32: 2e-5); the long-level programs alone, twice, on unchanged inputs: bit-identical (fixed-order statistics, no atomics)"""
    model = full_f32
    pn = make_plan(model, B, T, nrep, causal, True)
    pl = make_plan(model, B, T, nrep, causal, False)
    assert pn.long_levels >= 1 and not pn.long_errors, pn.long_errors
    kinds = [p.kinds[0] for p in pn.progs]
    assert kinds[0] == "long" and kinds[-1] == "long" and "gemm" in kinds, kinds
    assert len(pn.progs[0]) >= 19 and len(pn.progs[-1]) >= 19, [len(p) for p in pn.progs]
    assert pl.long_levels == 0 and all(p.kinds[0] != "long" for p in pl.progs)
    x, cond = synth.latents(B, T), synth.conditioning(B, T, "music_cont" if causal else "text_guided")
    t = np.array([(131 * i + 7) % 1000 for i in range(B)], dtype=np.int64)
    run_plan(model, pl, x, t, cond)
    run_plan(model, pn, x, t, cond)
    assert pn.take_error() == 0
    worst = compare_acts(pn, pl, 2e-5)
    s = torch.cuda.current_stream().cuda_stream
    for prog, out in ((pn.progs[0], pn.taps[f"down{pn.long_levels - 1}"].t), (pn.progs[-1], pn.net_out.t)):
        res = []
        for _ in range(2):
            prog.poison(s)
            prog.launch(s)
            torch.cuda.synchronize()
            res.append(out.clone())
        assert pn.take_error() == 0
        assert torch.isfinite(res[0].float()).all() and torch.equal(res[0], res[1]), "two runs of a long-level program differ bitwise"
    print(f"long levels B={B} T={T} nrep={nrep}: {pn.n_launch} launches (one launch per layer there: {pl.n_launch}), worst activation difference {worst:.2e}")


def test_long_levels_ticket_form_equals_static(full_f32):
    """the ticket form of the launch (any number of resident workgroups makes progress: what a plan uses while another program holds the
    device's static schedule) computes the same bits as the static form: each long-level program alone, on unchanged inputs"""
    model = full_f32
    B, T = 8, 1500
    pn = make_plan(model, B, T, 1, False, True)
    assert pn.long_levels >= 1
    x, cond = synth.latents(B, T), synth.conditioning(B, T, "text_guided")
    t = np.array([(131 * i + 7) % 1000 for i in range(B)], dtype=np.int64)
    run_plan(model, pn, x, t, cond)
    assert pn.take_error() == 0
    s = torch.cuda.current_stream().cuda_stream
    lead = pn.progs[0]
    was = lead.exclusive
    try:
        for prog, out in ((pn.progs[0], pn.taps[f"down{pn.long_levels - 1}"].t), (pn.progs[-1], pn.net_out.t)):
            res = []
            for form in (True, False, False):
                lead.exclusive = form
                prog.sync.zero_()                     # (the ticket counter: part of the per-step arena reset in a plan)
                prog.poison(s)
                prog.launch(s)
                torch.cuda.synchronize()
                res.append(out.clone())
            assert pn.take_error() == 0
            assert torch.isfinite(res[0].float()).all()
            assert torch.equal(res[0], res[1]) and torch.equal(res[1], res[2]), "the ticket form and the static form differ bitwise"
    finally:
        lead.exclusive = was
    # and through the whole plan (the pack kernel's statistics use float atomics: compared at the float32 tolerance)
    ref = [a.t.clone() for a in pn.acts]
    lead.exclusive = False
    try:
        run_plan(model, pn, x, t, cond)
    finally:
        lead.exclusive = was
    assert pn.take_error() == 0
    for i, (a, r) in enumerate(zip(pn.acts, ref)):
        ra, rb = a.t[:, :, : a.C].float(), r[:, :, : a.C].float()
        if torch.isfinite(rb).all() and float(rb.abs().max()) > 0:
            assert float((ra - rb).abs().max()) <= 2e-5 * float(rb.abs().max()), i


def test_long_levels_bf16(full_bf16):
    """bf16 storage: close to the launch path on every activation, same final output within bf16 rounding"""
    B, T = 8, 1500
    x, cond = synth.latents(B, T), synth.conditioning(B, T, "text_guided")
    t = np.array([(131 * i + 7) % 1000 for i in range(B)], dtype=np.int64)
    pn = make_plan(full_bf16, B, T, 1, False, True)
    pl = make_plan(full_bf16, B, T, 1, False, False)
    assert pn.long_levels >= 1 and len(pn.progs) == 3
    run_plan(full_bf16, pl, x, t, cond)
    run_plan(full_bf16, pn, x, t, cond)
    assert pn.take_error() == 0
    worst = compare_acts(pn, pl, 6e-2)
    print(f"long levels bf16: {pn.n_launch} launches (launch path {pl.n_launch}), worst activation difference {worst:.2e}")


def test_configurations_that_do_not_fit_keep_one_launch_per_layer():
    """the tiny configuration (64 channels: an M block of 128 GEMM rows does not exist) refuses the long-level launch and runs as before"""
    from jen1_amd.model import UNetCFG1d
    model = UNetCFG1d(**tiny_model_config(), init_seed=1234, compute_dtype="f32", device="cuda")
    p = model.engine().plan(2, 300, 2, False)
    assert p.long_levels == 0 and all(pr.kinds[0] != "long" for pr in p.progs)
    x, cond = synth.latents(2, 300), synth.conditioning(2, 300, "text_guided")
    run_plan(model, p, x, np.array([999, 499], dtype=np.int64), cond)
    assert torch.isfinite(p.net_out.t.float()).all()


def test_pack_input_fixed_order_statistics():
    """jen1_pack_input_parts + jen1_gn_stats_from_parts: the network input's fine-group GroupNorm sums without float atomics -- equal to a
    float64 sum of the same values (1e-6), identical bits on every run, CFG pair replicated"""
    from jen1_amd import lib as L
    lib = L.load()
    B, C, Cc, T, ld, nrep = 3, 128, 129, 1499, 288, 2
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn((B, C, T), device="cuda", generator=g)
    ctx = torch.randn((B, Cc, T), device="cuda", generator=g)
    s = torch.cuda.current_stream().cuda_stream
    res = []
    for _ in range(2):
        y = torch.zeros((nrep * B, T, ld), dtype=torch.bfloat16, device="cuda")
        parts = torch.empty((B, (T + 31) // 32, ld, 2), dtype=torch.float32, device="cuda")
        st = torch.full((nrep * B, 32, 2), float("nan"), dtype=torch.float32, device="cuda")
        L.check(lib.jen1_pack_input_parts(x.data_ptr(), ctx.data_ptr(), y.data_ptr(), parts.data_ptr(), B, C, Cc, T, ld, nrep, L.BF16, s), "pack")
        L.check(lib.jen1_gn_stats_from_parts(parts.data_ptr(), st.data_ptr(), B, T, ld, nrep, s), "finish")
        torch.cuda.synchronize()
        res.append((y.clone(), st.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    y, st = res[0]
    full = torch.cat([x, ctx], 1).double()                                     # [B, 257, T]
    full = torch.nn.functional.pad(full, (0, 0, 0, ld - C - Cc))
    assert torch.equal(y[:B].float(), full.transpose(1, 2).to(torch.bfloat16).float()) and torch.equal(y[:B], y[B:])
    fg = full.view(B, 32, ld // 32, T)
    want = torch.stack([fg.sum(dim=(2, 3)), (fg * fg).sum(dim=(2, 3))], -1)    # [B, 32, 2]
    got = st.double()
    assert float((got[:B] - want).abs().max()) <= 1e-6 * float(want.abs().max())
    assert torch.equal(st[:B], st[B:])


def test_default_sampling_is_bit_reproducible(full_bf16):
    """with fixed-order statistics in the pack kernel and in all three persistent launches the DEFAULT plan of
    the bench workload has no float atomics left: a 4-step DDIM trajectory from the same start ends on the same bits, eagerly and as a
    replayed graph"""
    from jen1_amd.diffusion import GaussianDiffusion, get_beta_schedule
    m = full_bf16
    B, T, S = 8, 1500, 4
    betas, _ = get_beta_schedule("linear", 1000)
    gd = GaussianDiffusion(steps=1000, betas=betas, objective="noise", loss_type="l2", device="cuda", cfg_dropout_proba=0.0,
                           embedding_scale=1.0, batch_cfg=True, scale_cfg=True, sampling_timesteps=S)
    cond = {k: dev(v) for k, v in synth.conditioning(B, T).items()}
    init = dev(synth.noise_list(1, (B, 128, T), seed=7)[0])
    noises = [dev(n) for n in synth.noise_list(S, (B, 128, T), seed=11)]
    outs = [gd.sample(m, (B, 128, T), cond, init_noise=init, step_noises=noises, use_graph=ug) for ug in (True, True, False)]
    torch.cuda.synchronize()
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]), "two replayed trajectories differ"
    assert torch.equal(outs[0], outs[2]), "the replayed and the eager trajectory differ"


@pytest.mark.parametrize("scale,eta", [(1.0, 1.0), (3.0, 1.0), (1.0, 0.0)])
def test_fused_step_pack_matches_separate_pack(full_bf16, monkeypatch, scale, eta):
    """the sampler's step kernel writes the next step's network input (jen1_cfg_ddim_step_pack: rows in the compute dtype + statistics
    partials in pack_input's order) instead of a pack launch at the head of every step, and -- jen1_step_tail -- the same launch sets the
    next step's sentinels and zeroes its statistics arena: same bits as the separate launches, with and without the CFG pair, with and
    without per-step noise, eagerly and as a replayed graph; a foreign host-side run of the plan between two steps (which consumes the
    sentinels) is noticed and repaired"""
    from jen1_amd.diffusion import DDIMStepper, GaussianDiffusion, get_beta_schedule
    m = full_bf16
    B, T, S = 2, 1500, 3
    betas, _ = get_beta_schedule("linear", 1000)
    gd = GaussianDiffusion(steps=1000, betas=betas, objective="v", loss_type="l2", device="cuda", cfg_dropout_proba=0.0,
                           embedding_scale=scale, batch_cfg=True, scale_cfg=True, sampling_timesteps=S, ddim_sampling_eta=eta)
    cond = {k: dev(v) for k, v in synth.conditioning(B, T).items()}
    init = dev(synth.noise_list(1, (B, 128, T), seed=3)[0])
    noises = [dev(n) for n in synth.noise_list(S, (B, 128, T), seed=5)]
    outs = {}
    for pack, tail, ug, foreign in (("0", "0", False, False), ("1", "0", False, False), ("1", "1", False, False), ("1", "1", True, False),
                                    ("1", "1", True, True)):
        monkeypatch.setenv("JEN1_STEP_PACK", pack)
        monkeypatch.setenv("JEN1_STEP_TAIL", tail)
        st = DDIMStepper(gd, m, (B, 128, T), cond, use_graph=ug)
        assert st.fused_pack == (pack == "1") and st.fused_tail == (tail == "1")
        st.reset(init, fresh_noise=False)
        for i in range(S):
            st.step(i, noise=noises[i])
            if foreign and i == 0:
                keep = st.x.clone()
                st.plan.run()                  # (a full pass of the same plan from the host: packs, poisons, consumes)
                assert torch.equal(keep, st.x)
        st.check()
        outs[(pack, tail, ug, foreign)] = st.x.clone()
    torch.cuda.synchronize()
    ref = outs[("0", "0", False, False)]
    assert torch.isfinite(ref).all()
    for k, v in outs.items():
        assert torch.equal(ref, v), f"JEN1_STEP_PACK={k[0]} JEN1_STEP_TAIL={k[1]} graph={k[2]} foreign run={k[3]}: differs from the separate launches"
