"""Bodies of the fault-injection / shared-device cases.  NOT collected by pytest: tests/test_gpu_zz_fault_injection.py starts

    python tests/fault_cases.py <case> [argument]

in a process of its own with its own model and judges the exit code -- a case that takes the interpreter down (a HIP runtime abort, a GPU
memory fault, a hang that the wrapper's timeout ends) fails ONE test instead of ending the run, and nothing these cases do to
class-level state (`DeepProgram._static_owner`), to the device's error words or to the caching allocator is seen by a parity test."""
import copy
import gc
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jen-1-pytorch_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

from helpers import rel_err  # noqa: E402
from jen1_amd import synth  # noqa: E402
from jen1_amd.config import full_model_config, tiny_model_config  # noqa: E402


def dev(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()


def full(mode):
    from jen1_amd.model import UNetCFG1d
    return UNetCFG1d(**full_model_config(), init_seed=1234, compute_dtype=mode, device="cuda")


def run_plan(model, plan, x, t, cond, drop=None):
    s = torch.cuda.current_stream().cuda_stream
    model._prepare(plan, dev(x), dev(t), dev(cond["cross_attn_cond"]), dev(cond["cross_attn_masks"]), [dev(cond["input_concat_cond"])], drop)
    plan.run(s)
    torch.cuda.synchronize()


def broken_copy(prog):
    """the persistent program re-linked WITHOUT its first phase: the first remaining phase polls a tensor nobody produces"""
    broken = copy.copy(prog)
    broken.leader = broken
    broken._exclusive = False
    broken._err_shared = prog.err
    for name in ("bufs", "labels", "outs", "kinds"):
        setattr(broken, name, list(getattr(prog, name))[1:])
    broken.finalize(prog.sync)
    return broken


# ---------------------------------------------------------------------------------------------------------------------
def concurrent(n):
    """several FULL-MODEL samplers in flight on one GPU, every one with its own persistent launch (own plan buffers, own replayed
    graph, own stream): units are handed to workgroups by ticket, so a launch whose workgroups are only partly resident still
    makes progress -- no deadlock, no time-out (error word 0) -- and every trajectory ends bit for bit where it ends alone
    (fixed-order statistics on the launch-per-layer levels make the runs reproducible)."""
    import weakref
    from jen1_amd.diffusion import GaussianDiffusion, get_beta_schedule
    from jen1_amd.engine import DeepProgram
    n = int(n)
    B, T, S = 2, 1500, 4
    betas, _ = get_beta_schedule("linear", 1000)
    m = full("f32")
    m.deterministic = True
    m.engine().deep_all_slots = True          # (by default only slot 0 gets the persistent launch: see Engine.plan)
    gd = GaussianDiffusion(steps=1000, betas=betas, objective="noise", loss_type="l2", device="cuda", cfg_dropout_proba=0.0,
                           embedding_scale=0.8, batch_cfg=True, scale_cfg=True, sampling_timesteps=S)
    tasks = ("text_guided", "music_inpaint", "music_cont", "text_guided")
    conds = [{k: dev(v) for k, v in synth.conditioning(B, T, tasks[i]).items()} for i in range(n)]
    inits = [dev(x) for x in synth.noise_list(n, (B, 128, T), seed=31)]
    noises = [dev(x) for x in synth.noise_list(S, (B, 128, T), seed=32)]
    sts, want = [], []
    for i in range(n):
        st = gd.stepper(m, (B, 128, T), conds[i], causal=False, use_graph=True, plan_slot=20 + i)
        assert st.plan.deep_level is not None
        st.reset(inits[i], fresh_noise=False)
        for k in range(S):
            st.step(k, noise=noises[k])
        torch.cuda.synchronize()
        st.check()
        sts.append(st)
        want.append(st.x.clone())
    assert len({st.plan.deep.dev.data_ptr() for st in sts}) == n           # n different programs / buffer sets
    # scheduling forms: at most one program per device uses the static unit -> workgroup map (needs all its workgroups
    # resident), everybody else goes by ticket.  Hand the static form to the first sampler here, so the runs below mix one
    # static launch with n - 1 ticket launches on the same GPU.
    assert sum(1 for st in sts if st.plan.deep.exclusive) <= 1
    old = DeepProgram._static_owner.get(str(m.engine().device))
    if old is not None and old() is not None:
        old().exclusive = False
    DeepProgram._static_owner[str(m.engine().device)] = weakref.ref(sts[0].plan.deep.leader)      # (the plan's first persistent program holds the claim)
    sts[0].plan.deep.exclusive = True
    streams = [torch.cuda.Stream() for _ in range(n)]
    for rep in range(3):
        for i, st in enumerate(sts):
            st.reset(inits[i], fresh_noise=False)
        torch.cuda.synchronize()
        for k in range(S):
            for st, s_ in zip(sts, streams):
                with torch.cuda.stream(s_):
                    st.step(k, noise=noises[k])
        torch.cuda.synchronize()
        for i, st in enumerate(sts):
            st.check()                                                       # raises on a time-out of any dependency wait
            assert torch.equal(st.x, want[i]), (n, rep, i, rel_err(st.x.cpu().numpy(), want[i].cpu().numpy()))


def nan_flow(mode):
    """A producer whose results ARE the reserved pattern (include/jen1_deep.h "Reserved word"): the bias of a level-3 convolution is set
    to float32 NaNs with the sign and every mantissa bit set (0xFFFFFFFF), so its epilogue computes acc + bias = that NaN for every
    channel -- stored as it is, every 8-byte word of the layer's output would be the 'not stored yet' sentinel and its consumers would
    spin to the time-out.  The stores canonicalise the pattern: the launch completes at its usual speed with NO error word, NaNs come
    out (as the reference produces them for NaN weights), and the next launch with the bias restored is clean and equal to the one before."""
    mode, _, where = str(mode).partition(":")       # "bf16" (a deep level: downsamples.3) or "bf16:long" (a long level: downsamples.1)
    level = "downsamples.1" if where == "long" else "downsamples.3"
    model = full(mode)
    eng = model.engine()
    B, T = 2, 1500
    plan = eng.plan(B, T, 1, False, deep=True)
    assert plan.deep_level is not None and (where != "long" or plan.long_levels >= 2)
    x, cond = synth.latents(B, T), synth.conditioning(B, T)
    t = np.array([999, 3], dtype=np.int64)
    run_plan(model, plan, x, t, cond)
    assert plan.take_error() == 0
    want = plan.net_out.t.clone()
    assert torch.isfinite(want.float()).all()
    keys = [k for k in eng.W.v if k.endswith("blocks.0.conv1.bias") and k.startswith(level)]
    assert keys, [k for k in eng.W.v if level in k][:8]
    bias = eng.W.v[keys[0]]
    saved = bias.clone()
    bias.view(torch.int32).fill_(-1)                    # 0xFFFFFFFF: -NaN, all mantissa bits
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_plan(model, plan, x, t, cond)
    dt = time.perf_counter() - t0
    assert plan.take_error() == 0, "a NaN activation was taken for the sentinel: a consumer waited for data that had arrived"
    assert dt < 0.1, f"the launch took {dt * 1e3:.0f} ms: a consumer spun on a NaN word"
    assert torch.isnan(plan.net_out.t.float()).any()
    bias.copy_(saved)
    run_plan(model, plan, x, t, cond)
    assert plan.take_error() == 0
    # (the launch-per-layer levels around the persistent launch sum their statistics with float atomics: equal to rounding)
    assert rel_err(plan.net_out.t.float().cpu().numpy(), want.float().cpu().numpy()) < (1e-4 if mode == "f32" else 3e-2)


def time_out(which=None):
    """The persistent program re-linked WITHOUT its first phase: the first remaining phase polls a tensor nobody produces (it starts the
    launch poisoned).  The bounded spin gives up after JEN1_DEEP_POLL_LIMIT polls, the error word says which phase, every other waiter
    is released (the launch ends in a fraction of a second instead of hanging the GPU), ``take_error`` reports and clears the word, and
    the intact program runs clean right after."""
    model = full("bf16")
    eng = model.engine()
    B, T = 2, 1500
    plan = eng.plan(B, T, 1, False, deep=True)
    # ("long": the same injection into a sample-resident long-level launch, include/jen1_long.h -- the up half: its phase 1 then polls the
    # output of a phase 0 that never runs)
    prog = plan.deep if which != "long" else plan.progs[-1]
    assert which != "long" or prog.kinds[0] == "long"
    x, cond = synth.latents(B, T), synth.conditioning(B, T)
    t = np.array([999, 3], dtype=np.int64)
    run_plan(model, plan, x, t, cond)
    assert plan.take_error() == 0
    want = plan.net_out.t.clone()
    broken = broken_copy(prog)
    s = torch.cuda.current_stream().cuda_stream
    prog.poison(s)                                          # (the full program's table: phase 0's output starts as the sentinel)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    broken.launch(s)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    e = prog.take_error()
    assert e >= 1, "the wait for a tensor nobody produced did not time out"
    assert dt < 5.0, f"{dt:.1f} s: the other waiters were not released"
    assert prog.error() == 0                                # cleared when reported
    run_plan(model, plan, x, t, cond)                       # the intact program, right after
    assert plan.take_error() == 0
    assert rel_err(plan.net_out.t.float().cpu().numpy(), want.float().cpu().numpy()) < 3e-2


def forward_time_out(_=None):
    """The same injection seen through the PUBLIC call: ``UNetCFG1d.forward`` whose persistent launch timed out.  Asynchronous mode (the
    default): the call itself returns; ``check_errors()`` raises ``Jen1HipError`` for it, so does the next ``forward``, and the literal
    sampler loops end with ``check_errors()`` (their last call is checked too).  ``strict_errors``: the call itself raises.  An optimiser
    step's ``_invalidate_engine`` reports a pending error instead of dropping it.  After the report the word is clear and a call is clean."""
    from jen1_amd.lib import Jen1HipError
    model = full("bf16")
    eng = model.engine()
    B, T = 2, 1500
    x, cond = synth.latents(B, T), synth.conditioning(B, T)
    t = np.array([999, 3], dtype=np.int64)
    kw = dict(embedding=dev(cond["cross_attn_cond"]), embedding_mask=dev(cond["cross_attn_masks"]), channels_list=[dev(cond["input_concat_cond"])])
    want = model(dev(x), dev(t), **kw).clone()
    model.check_errors()
    plan = eng.plan(B, T, 1, False)
    assert plan.deep_level is not None
    idx = [i for i, op in enumerate(plan.ops) if getattr(op, "kind", "") == "deep"]
    assert len(idx) == 1
    good = plan.ops[idx[0]]
    broken = broken_copy(plan.deep)
    bad = lambda s: broken.launch(s)
    bad.kind, bad.prog, bad.label = "deep", broken, "deep[fault injection: first phase removed]"

    def expect_raise(fn, what):
        try:
            fn()
        except Jen1HipError as e:
            assert "timed out" in str(e), e
            return
        raise AssertionError(f"{what}: no Jen1HipError")

    # 1. asynchronous (default): the call returns, check_errors() raises, and the report clears the word
    plan.ops[idx[0]] = bad
    model(dev(x), dev(t), **kw)
    expect_raise(model.check_errors, "check_errors() after a timed-out forward")
    model.check_errors()                                     # reported once
    assert plan.progs[0].error() == 0
    # 2. the NEXT forward raises for the previous one
    model(dev(x), dev(t), **kw)
    torch.cuda.synchronize()
    plan.ops[idx[0]] = good
    expect_raise(lambda: model(dev(x), dev(t), **kw), "the forward after a timed-out forward")
    y = model(dev(x), dev(t), **kw)
    model.check_errors()
    assert rel_err(y.cpu().numpy(), want.cpu().numpy()) < 3e-2
    # 3. strict_errors: the call itself raises
    model.strict_errors = True
    plan.ops[idx[0]] = bad
    expect_raise(lambda: model(dev(x), dev(t), **kw), "strict_errors forward")
    plan.ops[idx[0]] = good
    y = model(dev(x), dev(t), **kw)
    assert rel_err(y.cpu().numpy(), want.cpu().numpy()) < 3e-2
    model.strict_errors = False
    # 4. a pending error is not dropped by an engine invalidation (an optimiser step)
    plan.ops[idx[0]] = bad
    model(dev(x), dev(t), **kw)
    expect_raise(model._invalidate_engine, "_invalidate_engine with a pending error")
    plan.ops[idx[0]] = good
    # 5. the literal sampler loop (a model the fused stepper refuses: here, forced) checks its LAST call
    from jen1_amd.diffusion import GaussianDiffusion, get_beta_schedule
    model2 = full("bf16")
    betas, _ = get_beta_schedule("linear", 1000)
    gd = GaussianDiffusion(steps=1000, betas=betas, objective="noise", loss_type="l2", device="cuda", cfg_dropout_proba=0.0,
                           embedding_scale=1.0, batch_cfg=False, scale_cfg=False, sampling_timesteps=1)
    condd = {k: dev(v) for k, v in cond.items()}
    gd._ddim_generic(model2, (B, 128, T), condd, False, False, None, None, None, None)      # builds the plan; clean
    plan2 = model2.engine().plan(B, T, 1, False)
    i2 = [i for i, op in enumerate(plan2.ops) if getattr(op, "kind", "") == "deep"][0]
    broken2 = broken_copy(plan2.deep)
    bad2 = lambda s: broken2.launch(s)
    bad2.kind, bad2.prog, bad2.label = "deep", broken2, "deep[fault injection]"
    plan2.ops[i2] = bad2
    expect_raise(lambda: gd._ddim_generic(model2, (B, 128, T), condd, False, False, None, None, None, None),
                 "one-step literal DDIM loop whose only call timed out")


def gc_in_capture(_=None):
    """Regression of the round-4 SIGABRT (jen1_amd/graphs.py): a dead stepper -- its replayed hipGraph sits in a reference cycle -- is
    pending collection while new graphs are captured, and the cyclic collector is set to run at EVERY allocation.  torch's
    ``~CUDAGraph`` inside a capture window aborts the process on ROCm; the package's capture helper collects before the window and keeps the
    collector out of it.  Sampling and training captures both."""
    from jen1_amd.diffusion import GaussianDiffusion, get_beta_schedule
    from jen1_amd.model import UNetCFG1d
    betas, _ = get_beta_schedule("linear", 1000)
    cond = {k: dev(v) for k, v in synth.conditioning(2, 300).items()}

    def sample_once(S):
        m = UNetCFG1d(**tiny_model_config(), init_seed=1234, compute_dtype="f32", device="cuda")
        gd = GaussianDiffusion(steps=1000, betas=betas, objective="noise", loss_type="l2", device="cuda", cfg_dropout_proba=0.0,
                               embedding_scale=0.8, batch_cfg=True, scale_cfg=True, sampling_timesteps=S)
        y = gd.sample(m, (2, 128, 300), cond)
        torch.cuda.synchronize()
        return y

    gc.collect()
    gc.disable()
    a = sample_once(4)                       # its stepper, plan and graph are garbage now, uncollected
    gc.enable()
    old = gc.get_threshold()
    gc.set_threshold(1, 1, 1)                # the collector runs at every container allocation from here on
    try:
        b = sample_once(3)                   # warm-up + capture + replay with the dead graph pending
        m = UNetCFG1d(**tiny_model_config(), init_seed=1234, compute_dtype="f32", device="cuda")
        m.train()
        gd = GaussianDiffusion(steps=1000, betas=betas, objective="noise", loss_type="l2", device="cuda", cfg_dropout_proba=0.0,
                               embedding_scale=0.8, batch_cfg=True, scale_cfg=True)
        for _ in range(2):                   # capture, then replay
            loss = gd.training_loosses(m, dev(synth.latents(2, 300, key="clip")), torch.tensor([17, 801], device="cuda"), cond, causal=False)
            loss.backward()
        torch.cuda.synchronize()
        assert torch.isfinite(loss)
    finally:
        gc.set_threshold(*old)
    assert torch.isfinite(a).all() and torch.isfinite(b).all()


CASES = {f.__name__: f for f in (concurrent, nan_flow, time_out, forward_time_out, gc_in_capture)}

if __name__ == "__main__":
    name = sys.argv[1]
    if name == "abort":                      # the wrapper's own self-test: what a dying case looks like from outside
        os.abort()
    CASES[name](sys.argv[2] if len(sys.argv) > 2 else None)
    torch.cuda.synchronize()
    print(f"fault case {name} {sys.argv[2:]} OK", flush=True)
