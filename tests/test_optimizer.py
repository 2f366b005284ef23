"""Optimiser step and gradient exchange (e).
CPU: the oracle restatement against torch.optim.AdamW / clip_grad_norm_ / LinearLR (the reference uses exactly these:
train.py:56-60, :84, trainer.py:144-149), the flat-buffer host logic, the bucketed mean all-reduce on gloo world-2.
GPU: jen1_grad_sqnorm + jen1_adamw_step through the C ABI against the oracle and torch."""
import os
import sys

import numpy as np
import pytest
import torch

from helpers import ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import jen1_oracle as O  # noqa: E402  (the checker, tests only)
from jen1_amd.optim import FusedAdamW, GradExchange, LinearLR, allreduce_gradients  # noqa: E402

SHAPES = [(37, 5, 3), (128,), (64, 33), (1,), (19, 7)]


def _torch_run(params0, grads, steps, lr0=3e-5, max_norm=0.7):
    ps = [torch.nn.Parameter(torch.from_numpy(p.copy())) for p in params0]
    opt = torch.optim.AdamW(ps, lr=lr0, betas=(0.9, 0.95), weight_decay=0.1)
    sched = torch.optim.lr_scheduler.LinearLR(opt)
    norms = []
    for k in range(steps):
        for p, g in zip(ps, grads[k]):
            p.grad = torch.from_numpy(g.copy())
        norms.append(float(torch.nn.utils.clip_grad_norm_(ps, max_norm)))
        opt.step()
        sched.step()
    return [p.detach().numpy() for p in ps], norms


def _data(steps, seed=0, scale=1.0):
    rng = np.random.default_rng(seed)
    params0 = [rng.standard_normal(s).astype(np.float32) for s in SHAPES]
    grads = [[(rng.standard_normal(s) * scale).astype(np.float32) for s in SHAPES] for _ in range(steps)]
    return params0, grads


def test_oracle_adamw_clip_linearlr_match_torch():
    steps = 7
    params0, grads = _data(steps, scale=0.3)
    want, norms = _torch_run(params0, grads, steps)
    ps = [p.copy() for p in params0]
    ms = [np.zeros_like(p) for p in ps]
    vs = [np.zeros_like(p) for p in ps]
    for k in range(steps):
        g, total = O.clip_grad_norm(grads[k], 0.7)
        assert abs(total - norms[k]) < 1e-4 * norms[k]
        lr = 3e-5 * O.linear_lr_factor(k)
        for i in range(len(ps)):
            ps[i], ms[i], vs[i] = O.adamw_step(ps[i], g[i], ms[i], vs[i], k + 1, lr=lr)
    for a, b in zip(ps, want):
        assert np.abs(a - b).max() <= 2e-7 * max(1.0, np.abs(b).max())
    sched = LinearLR(3e-5)
    for k in range(9):
        assert abs(sched.get_last_lr() - 3e-5 * O.linear_lr_factor(k)) < 1e-12
        sched.step()


def test_flat_buffers_are_views_and_aligned():
    params0, _ = _data(1)
    ps = [torch.nn.Parameter(torch.from_numpy(p.copy())) for p in params0]
    opt = FusedAdamW(ps)
    assert all(o % 4 == 0 for o in opt.offsets) and opt.numel % 4 == 0
    for p, p0, o in zip(ps, params0, opt.offsets):
        assert np.array_equal(p.detach().numpy(), p0)
        assert p.data_ptr() == opt.flat_param.data_ptr() + 4 * o and p.grad.data_ptr() == opt.flat_grad.data_ptr() + 4 * o
    ps[2].grad.fill_(2.0)
    assert float(opt.flat_grad.sum()) == 2.0 * ps[2].numel()
    opt.zero_grad()
    assert float(ps[2].grad.abs().sum()) == 0.0
    if not torch.cuda.is_available():
        from jen1_amd.lib import Jen1HipError
        with pytest.raises(Jen1HipError):
            opt.step()


def _allreduce_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    allreduce_gradients(g, bucket_bytes=1024)            # 4 buckets of 256 floats
    q.put((rank, g.numpy().copy()))
    dist.destroy_process_group()


def test_gradient_allreduce_mean_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_allreduce_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    want = np.arange(1000, dtype=np.float32) * 1.5        # mean of 1x and 2x
    assert np.allclose(outs[0], want) and np.allclose(outs[1], want)


@pytest.mark.gpu
@pytest.mark.parametrize("max_norm,scale", [(0.7, 0.3), (0.7, 1e-3), (None, 0.3)])
def test_fused_adamw_matches_oracle_and_torch(max_norm, scale):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    steps = 5
    params0, grads = _data(steps, seed=3, scale=scale)
    ps = [torch.nn.Parameter(torch.from_numpy(p.copy()).cuda()) for p in params0]
    opt = FusedAdamW(ps, lr=3e-5, betas=(0.9, 0.95), weight_decay=0.1, max_norm=max_norm)
    sched = LinearLR(3e-5)
    ref, ms, vs = [p.copy() for p in params0], [np.zeros_like(p) for p in params0], [np.zeros_like(p) for p in params0]
    for k in range(steps):
        for p, g in zip(ps, grads[k]):
            p.grad.copy_(torch.from_numpy(g))
        opt.step(lr=sched.get_last_lr())
        g = grads[k]
        if max_norm is not None:
            g, total = O.clip_grad_norm(g, max_norm)
            assert abs(float(opt.grad_norm()) - total) < 1e-5 * total
        for i in range(len(ref)):
            ref[i], ms[i], vs[i] = O.adamw_step(ref[i], g[i], ms[i], vs[i], k + 1, lr=sched.get_last_lr())
        sched.step()
    # float32 throughout; the kernel's a*b+c are fused multiply-adds while numpy / torch round every operation:
    # a few ulp after 5 steps (1 ulp of a value in [2, 4) is 2.4e-7)
    tol = 1e-6
    for p, r in zip(ps, ref):
        assert np.abs(p.detach().cpu().numpy() - r).max() <= tol * max(1.0, np.abs(r).max())
    if max_norm is not None:
        want, _ = _torch_run(params0, grads, steps, max_norm=max_norm)
        for p, r in zip(ps, want):
            assert np.abs(p.detach().cpu().numpy() - r).max() <= tol * max(1.0, np.abs(r).max())


@pytest.mark.gpu
def test_fused_adamw_skips_nonfinite_and_large_buffer():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    n = (1 << 22) + 3
    p = torch.nn.Parameter(torch.randn(n, device="cuda"))
    opt = FusedAdamW([p], max_norm=0.7, skip_nonfinite=True)
    before = p.detach().clone()
    p.grad.normal_()
    p.grad[12345] = float("inf")
    opt.step()
    torch.cuda.synchronize()
    assert torch.equal(p.detach(), before) and float(opt.exp_avg.abs().sum()) == 0.0
    assert opt.step_count == 0          # a skipped step is not counted (GradScaler.step + torch AdamW: trainer.py:146)
    p.grad.normal_()
    g = p.grad.clone()
    opt.step()
    total = float(g.double().norm())
    assert abs(float(opt.grad_norm()) - total) < 1e-4 * total
    coef = min(1.0, 0.7 / (total + 1e-6))
    m = 0.1 * g * coef
    v = 0.05 * (g * coef) ** 2
    # ... so the first step TAKEN uses the bias corrections of step 1
    want = before * (1 - 3e-5 * 0.1) - (3e-5 / (1 - 0.9)) * m / (v.sqrt() / (1 - 0.95) ** 0.5 + 1e-8)
    assert float((p.detach() - want).abs().max()) < 1e-6
    assert opt.step_count == 1


@pytest.mark.gpu
def test_gradient_norm_is_bit_reproducible():
    """clip_grad_norm_'s norm (jen1_grad_sqnorm) must not depend on the arrival order of the blocks: data-parallel
    replicas compute it independently and have to clip identically"""
    from jen1_amd.optim import FusedAdamW
    p = torch.nn.Parameter(torch.zeros(5_000_003, device="cuda"))
    opt = FusedAdamW([p], lr=0.0, weight_decay=0.0, max_norm=0.7)
    p.grad.normal_(0, 1e-2)
    vals = []
    for _ in range(6):
        opt.step()
        torch.cuda.synchronize()
        vals.append(opt._gnorm_sq.clone())
    assert all(torch.equal(vals[0], v) for v in vals)
    ref = float((p.grad.double() ** 2).sum())
    assert abs(float(vals[0]) - ref) <= 1e-5 * ref


class _FlatOpt:
    """the part of FusedAdamW that GradExchange reads (flat buffer + per-parameter offsets), on CPU"""

    def __init__(self, shapes):
        self.params, self.offsets, n = [], [], 0
        for sh in shapes:
            p = torch.nn.Parameter(torch.zeros(sh))
            self.params.append(p)
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4
        self.flat_grad = torch.zeros(n)


_EX_NAMES = ["to_mapping.0.weight", "to_time.0.0.weights", "to_in.block.a", "downsamples.0.x", "downsamples.0.y", "downsamples.1.x",
             "bottleneck.a", "upsamples.0.a", "upsamples.1.a", "upsamples.1.b", "to_out.block.a", "to_time_embedding.0.1.weight",
             "fixed_embedding.embedding.weight"]
_EX_SHAPES = [(7, 5), (6,), (33,), (100, 3), (17,), (64, 9), (300,), (1000,), (13, 13), (5,), (77,), (4, 129), (129, 16)]


def _exchange_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    opt = _FlatOpt(_EX_SHAPES)
    g0 = torch.randn(opt.flat_grad.numel(), generator=torch.Generator().manual_seed(100 + rank))
    ex = GradExchange(opt, _EX_NAMES, bucket_bytes=512)                # several chunks per region
    # overlapped: the backward pass releases the regions from the back, two sub-batches share the pass (two hooks per region)
    opt.flat_grad.copy_(g0)
    ex.begin()
    order = ["to_out", "upsamples.1", "upsamples.0", "bottleneck", "downsamples.1", "downsamples.0"]
    for r in order:
        ex.expect(r)
        ex.expect(r)
    for r in order:
        ex.region_ready(r)
        assert r not in ex._sent
        ex.region_ready(r)
        assert r in ex._sent
    ex.finish()                                                          # head (to_mapping, to_time, to_in) and tail regions
    overlapped = opt.flat_grad.clone()
    opt.flat_grad.copy_(g0)
    ex.blocking()
    blocking = opt.flat_grad.clone()
    # the ordering check (debug_check): a region whose gradient changes after its hook fired is reported, a quiet pass is not
    opt.flat_grad.copy_(g0)
    ex.debug_check = True
    ex.begin()
    ex.region_ready("to_out")
    ex.finish()
    assert torch.equal(opt.flat_grad, blocking)
    opt.flat_grad.copy_(g0)
    ex.begin()
    ex.region_ready("to_out")
    lo, hi = ex.regions["to_out"]
    opt.flat_grad[lo] += 1.0
    try:
        ex.finish()
        raised = False
    except RuntimeError:
        raised = True
    assert raised
    ex.debug_check = False
    # bf16 on the wire: same chunks, the mean is bf16-accurate
    exb = GradExchange(opt, _EX_NAMES, bucket_bytes=512, grad_dtype="bf16")
    opt.flat_grad.copy_(g0)
    exb.blocking()
    assert float((opt.flat_grad - blocking).abs().max()) <= 2e-2 * float(blocking.abs().max())
    assert not exb.capturable                  # gloo / CPU buffers cannot be recorded into a graph
    q.put((rank, overlapped.numpy().copy(), blocking.numpy().copy(), g0.numpy().copy(), dict(ex.regions)))
    dist.destroy_process_group()


def test_grad_exchange_overlapped_equals_blocking_gloo_world2():
    """DDP's bucketed exchange (optim.GradExchange): regions released during the backward pass give bit-for-bit what the
    blocking exchange gives, both equal the mean over ranks, and the regions tile the flat buffer in forward order"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + os.getpid() % 2000
    procs = [ctx.Process(target=_exchange_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = {r: rest for r, *rest in (q.get(timeout=120) for _ in range(2))}
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    mean = (outs[0][2] + outs[1][2]) / 2
    for r in range(2):
        ov, bl, _, regions = outs[r]
        assert np.array_equal(ov, bl)
        assert np.array_equal(ov, mean.astype(np.float32))
    regions = outs[0][3]
    names = list(regions)
    assert names == ["to_mapping", "to_time", "to_in", "downsamples.0", "downsamples.1", "bottleneck", "upsamples.0", "upsamples.1",
                     "to_out", "to_time_embedding", "fixed_embedding"]
    assert regions[names[0]][0] == 0 and all(regions[a][1] == regions[b][0] for a, b in zip(names[:-1], names[1:]))


def test_optimizer_state_is_torch_adamw_schema_both_ways():
    """FusedAdamW.state_dict() / load_state_dict() speak torch.optim.AdamW's format (script_util.py:79-124 saves and loads the
    optimiser with it): a state written here loads into torch's AdamW over the same parameters, a state written by torch's AdamW
    (the reference's optimiser, train.py:56-60) loads here, shapes / counts are validated, the earlier flat format still loads"""
    shapes = [(5, 3), (7,), (2, 3, 4), (1,)]
    g = torch.Generator().manual_seed(3)
    mine = [torch.nn.Parameter(torch.randn(sh, generator=g)) for sh in shapes]
    theirs = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    ref = torch.optim.AdamW(theirs, lr=3e-5, betas=(0.9, 0.95), weight_decay=0.1)
    for it in range(3):
        for p in theirs:
            p.grad = torch.randn(p.shape, generator=g)
        ref.step()
    opt = FusedAdamW(mine, lr=1e-3)
    opt.load_state_dict(ref.state_dict())                                    # reference -> here
    assert opt.step_count == 3 and opt.lr == 3e-5 and tuple(opt.betas) == (0.9, 0.95) and opt.weight_decay == 0.1
    for i, (p, o) in enumerate(zip(opt.params, opt.offsets)):
        st = ref.state_dict()["state"][i]
        assert torch.equal(opt.exp_avg[o:o + p.numel()].view_as(p), st["exp_avg"])
        assert torch.equal(opt.exp_avg_sq[o:o + p.numel()].view_as(p), st["exp_avg_sq"])
    sd = opt.state_dict()                                                    # here -> reference
    assert set(sd) == {"state", "param_groups"} and sd["param_groups"][0]["params"] == [0, 1, 2, 3]
    assert set(sd["param_groups"][0]) == set(ref.state_dict()["param_groups"][0])
    other = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(sh)) for sh in shapes], lr=1.0)
    other.load_state_dict(sd)
    back = other.state_dict()
    for i in range(4):
        assert float(back["state"][i]["step"]) == 3.0
        assert torch.equal(back["state"][i]["exp_avg"], ref.state_dict()["state"][i]["exp_avg"])
        assert torch.equal(back["state"][i]["exp_avg_sq"], ref.state_dict()["state"][i]["exp_avg_sq"])
    assert back["param_groups"][0]["lr"] == 3e-5 and back["param_groups"][0]["weight_decay"] == 0.1
    # a fresh optimiser (no step yet) has an empty state, like torch's
    assert FusedAdamW([torch.nn.Parameter(torch.zeros(3))]).state_dict()["state"] == {}
    # refusals: wrong parameter count, wrong shape
    with pytest.raises(ValueError):
        FusedAdamW([torch.nn.Parameter(torch.zeros(3))]).load_state_dict(sd)
    bad = ref.state_dict()
    bad["state"][1]["exp_avg"] = torch.zeros(8)
    with pytest.raises(ValueError):
        opt.load_state_dict(bad)
    # the flat format of round-1 checkpoints
    flat = {"step": 7, "exp_avg": torch.arange(opt.numel, dtype=torch.float32), "exp_avg_sq": torch.ones(opt.numel)}
    opt.load_state_dict(flat)
    assert opt.step_count == 7 and float(opt.exp_avg[5]) == 5.0


def test_adopt_torch_optimizer_and_scheduler_keep_the_configured_rate_and_state():
    """The reference's call site (train.py:56-60, :84, :110-125) hands the trainer a torch AdamW whose LinearLR has ALREADY scaled
    param_groups[0]['lr'] by start_factor.  The takeover must schedule from the configured rate (lr sequence == torch's) and carry
    the moments / step count of a resumed optimiser into the flat buffers."""
    from jen1_amd.optim import adopt_optimizer
    params0, grads = _data(3, scale=0.3)
    ps = [torch.nn.Parameter(torch.from_numpy(p.copy())) for p in params0]
    opt = torch.optim.AdamW(ps, lr=3e-5, betas=(0.9, 0.95), weight_decay=0.1)
    sched = torch.optim.lr_scheduler.LinearLR(opt)                     # train.py:84: built before the trainer
    assert abs(opt.param_groups[0]["lr"] - 1e-5) < 1e-12               # ... which is why the group lr is not the base rate
    want = []
    ref_opt = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=3e-5)
    ref_sched = torch.optim.lr_scheduler.LinearLR(ref_opt)
    for _ in range(8):
        want.append(ref_sched.get_last_lr()[0])
        ref_opt.step()
        ref_sched.step()
    fused, lin = adopt_optimizer(ps, opt, sched, grad_clip=0.7)
    assert isinstance(fused, FusedAdamW) and isinstance(lin, LinearLR)
    assert abs(fused.lr - 3e-5) < 1e-12 and fused.max_norm == 0.7 and fused.step_count == 0
    got = []
    for _ in range(8):
        got.append(lin.get_last_lr())
        lin.step()
    assert np.allclose(got, want, rtol=1e-12, atol=0), (got, want)
    # resumed run: two torch steps first, then the takeover (moments, step count and the scheduler's position are kept)
    ps = [torch.nn.Parameter(torch.from_numpy(p.copy())) for p in params0]
    opt = torch.optim.AdamW(ps, lr=3e-5, betas=(0.9, 0.95), weight_decay=0.1)
    sched = torch.optim.lr_scheduler.LinearLR(opt)
    for k in range(2):
        for p, g in zip(ps, grads[k]):
            p.grad = torch.from_numpy(g.copy())
        opt.step()
        sched.step()
    m_ref = [opt.state[p]["exp_avg"].clone() for p in ps]
    v_ref = [opt.state[p]["exp_avg_sq"].clone() for p in ps]
    fused, lin = adopt_optimizer(ps, opt, sched, grad_clip=0.7)
    assert fused.step_count == 2 and abs(fused.lr - 3e-5) < 1e-12
    assert abs(lin.get_last_lr() - want[2]) < 1e-15
    for p, o, m, v in zip(fused.params, fused.offsets, m_ref, v_ref):
        assert torch.equal(fused.exp_avg[o:o + p.numel()].view_as(p), m) and torch.equal(fused.exp_avg_sq[o:o + p.numel()].view_as(p), v)
    # a FusedAdamW / LinearLR pair passes through (only clip norm and skip flag are applied)
    f2, l2 = adopt_optimizer(ps, fused, lin, grad_clip=0.5, skip_nonfinite=True)
    assert f2 is fused and l2 is lin and fused.max_norm == 0.5 and fused.skip_nonfinite
