#!/usr/bin/env python3
"""bench.py -- denoiser steps/sec of the JEN-1 hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one denoiser step of the DDIM sampler: one UNetCFG1d forward over the batch plus the
fused x0/eps prediction and DDIM update (reference gdm.py:202-222).  Workload at N=1 is
BASELINE.json configs[1]: full JEN-1 1D-UNet (296.5 M parameters, random init), B=8 synthetic
Encodec latents 128x1500, 100-step DDIM schedule, bf16 storage / fp32 accumulate.  With N>1 every
rank runs its own B=8 batch (independent samples, no data-path collective): weak scaling.

Inside the timed region: every launch of the step -- five since round 6: the sample-resident long-level launch of the down path
(csrc/long_kernel.hip), the persistent deep-level launch (csrc/deep_kernel.hip), the long-level launch of the up path, the step's
tail (jen1_step_tail: CFG / DDIM update, the next step's network input, the next step's sentinels) and the sum of the input's
GroupNorm partials.  Computed once per sampling run, outside it: the text
K/V projection and the time-embedding / FiLM / time-token K/V tables of the whole schedule; the DDIM
noise (eta = 1, the reference's default) is a table drawn before the region and read inside it; CFG
dropout is off (sampling).  ``extra.end_to_end`` times whole ``sample()`` calls including all of that.
The timed region of --steps steps is repeated (>= 3 regions) and the median region is the headline.

The JSON line also carries
  roofline      -- the dominant kernel of the step, ``deep_kernel`` (the persistent launch that
                   runs every level below T'=64 as one phase list): algorithmic HBM bytes of its
                   phases / its measured launch duration (HIP events on the launch stream, the
                   launch replayed alone in a HIP graph) against the 8 TB/s HBM3E peak of
                   MI355X_MICROARCH.md; ``long_levels`` holds the same figure for the two
                   sample-resident launches of the levels above it (``long_kernel``);
  cpu_baseline  -- the torch-CPU oracle (oracle/jen1_oracle_torch.py, a restatement of the
                   reference's CPU path, all physical host cores) on a bounded sample of the same
                   workload (rank 0, N=1 only);
  extra         -- configs[2] (CFG pair, effective batch 16) measured the same way, and the other
                   rows of SURVEY.md section 8.

    python bench.py --mode train --gpus N     (torchrun as above)
measures BASELINE configs[3]'s per-GPU shape instead: one optimiser step of the multi-task trainer
(8 clips per GPU as 3/3/2 sub-batches through the CFG pair, gradient exchange over RCCL overlapped
with the backward pass, clip + AdamW), reported in clips/s.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "jen-1-pytorch_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from jen1_amd.graphs import capture as capture_graph  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=0, help="timed regions of --steps steps each (0: as many as make >= 600 steps, at least 3)")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--length", type=int, default=1500)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "fp8"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--tiny", action="store_true", help="configs[0] tiny UNet (plumbing check)")
    ap.add_argument("--mode", default="sample", choices=["sample", "train"])
    ap.add_argument("--eager-train", action="store_true", help="--mode train: eager backward (overlapped exchange) only")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="sampling with --gpus N: weak = every rank its own --batch samples (replicas); strong = --batch is the GLOBAL batch, "
                         "split over the ranks (SURVEY.md section 8e: both forms)")
    ap.add_argument("--accum", type=int, default=1, help="--mode train: micro-batches per optimiser step (the reference's window is 10, "
                                                         "trainer.py:139-143, utils/config.py:96): the gradient exchange runs once per window")
    ap.add_argument("--grad-bf16", action="store_true", help="--mode train: gradient exchange in bf16 buckets (float32 master accumulate)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend (nccl = RCCL on ROCm; gloo: CPU plumbing check)")
    ap.add_argument("--dry-run", action="store_true", help="launcher / rendezvous / collective plumbing only: no GPU work, value null")
    return ap.parse_args()


def relaunch_under_torchrun(args) -> int:
    """``python bench.py --gpus N`` with N > 1 and no torchrun around it: start the N ranks ourselves (one process per GPU, the
    reference's mp.spawn + init_process_group, train.py:14-31) through ``python -m torch.distributed.run`` on 127.0.0.1 and hand
    the child's exit code back.  The children see WORLD_SIZE and take the normal path."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def dev(a, device):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(device)


def make_diffusion(device, cfg_pair):
    from jen1_amd.diffusion import GaussianDiffusion, get_beta_schedule
    betas, _ = get_beta_schedule("linear", 1000)
    return GaussianDiffusion(steps=1000, betas=betas, objective="noise", loss_type="l2", device=device,
                             cfg_dropout_proba=0.0, embedding_scale=0.8 if cfg_pair else 1.0, batch_cfg=True,
                             scale_cfg=True, sampling_timesteps=100)


def build_stepper(model, B, T, device, cfg_pair, use_graph, plan_slot=0):
    from jen1_amd import synth
    gd = make_diffusion(device, cfg_pair)
    cond = {k: dev(v, device) for k, v in synth.conditioning(B, T).items()}
    st = gd.stepper(model, (B, 128, T), cond, causal=False, use_graph=use_graph, plan_slot=plan_slot)
    st.reset(dev(synth.latents(B, T), device))
    return st


def timed_steps(st, steps, warmup, barrier, repeats=1):
    """``warmup`` untimed steps, then ``repeats`` timed regions of EXACTLY ``steps`` steps, each bracketed by barrier +
    synchronize on both sides.  Returns the list of region durations (seconds)."""
    for i in range(warmup):
        st.step(i % st.num_steps)
    out = []
    k = warmup
    for _ in range(repeats):
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            st.step((k + i) % st.num_steps)
        torch.cuda.synchronize()
        barrier()
        out.append(time.perf_counter() - t0)
        k += steps
    return out if repeats > 1 else out[0]


def end_to_end_bench(model, st, B, T, device):
    """What the timed region leaves out, measured: (a) the per-run set-up of a sampling run -- text K/V projection
    (``Plan.set_context``) and the schedule tables (``Plan.run_time``: time MLP, FiLM GEMM and time-token K/V GEMM for all 100
    steps) -- and (b) the wall time of real ``GaussianDiffusion.sample()`` calls of 100 DDIM steps on the bench shape (start noise,
    set-up, 100 replays, final copy, error check, one host sync), first call (builds the plan's graph) and repeated call."""
    from jen1_amd import synth
    plan = st.plan
    cond = {k: dev(v, device) for k, v in synth.conditioning(B, T).items()}
    s = torch.cuda.current_stream(device).cuda_stream
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    torch.cuda.synchronize()
    e[0].record()
    plan.set_context(cond["cross_attn_cond"], cond["cross_attn_masks"], s)
    e[1].record()
    plan.run_time(s)
    e[2].record()
    torch.cuda.synchronize()
    gd = make_diffusion(device, False)
    walls = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x = gd.sample(model, (B, 128, T), cond)
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
    assert bool(torch.isfinite(x).all())
    return {"set_context_ms": round(e[0].elapsed_time(e[1]), 3), "run_time_tables_ms": round(e[1].elapsed_time(e[2]), 3),
            "setup_ms": round(e[0].elapsed_time(e[2]), 3),
            "sample_100_steps_wall_ms": {"first_call": round(walls[0] * 1e3, 1), "repeated_call": round(min(walls[1:]) * 1e3, 1)},
            "steps_per_s_end_to_end": round(100.0 / min(walls[1:]), 1)}


def concurrent_batches_bench(model, B, T, device, n, steps, warmup):
    """Serving view of the same workload: n independent B=8 sample batches in flight on one GPU, each with its own
    plan buffers and its own replayed graph on its own stream, sharing one copy of the packed weights.  A single batch
    is a chain of ~300 dependent launches that leaves most of the chip idle between them; independent chains overlap.
    Reported beside the headline number, never instead of it."""
    sts = [build_stepper(model, B, T, device, cfg_pair=False, use_graph=True, plan_slot=1 + i) for i in range(n)]
    streams = [torch.cuda.Stream(device) for _ in range(n)]
    cur = torch.cuda.current_stream(device)

    def run(k0, k):
        for s in streams:
            s.wait_stream(cur)
        for i in range(k):
            for st, s in zip(sts, streams):
                with torch.cuda.stream(s):
                    st.step((k0 + i) % st.num_steps)
        for s in streams:
            cur.wait_stream(s)
    run(0, warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(warmup, steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for st in sts:
        st.check()                 # the error word of every persistent launch (a dependency wait that timed out raises here)
    return {"batches_in_flight": n, "steps_per_s_aggregate": round(n * steps / dt, 1), "ms_per_step_per_batch": round(dt / steps * 1e3, 3)}


def latest_profile(suffix):
    """profiles/rNN_<suffix> of the highest round present"""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    if not c:
        raise FileNotFoundError(suffix)
    return c[-1]


def pmc_entry(name):
    """HBM bytes / MFMA-busy of the committed rocprofv3 --pmc passes of this command (profiles/rNN_pmc.json of the latest
    round, produced by tools/profile_round.sh -> tools/pmc_summary.py from the per-pass databases; FETCH_SIZE / WRITE_SIZE
    corrected as MI355X_MICROARCH.md prescribes)"""
    try:
        path = latest_profile("pmc.json")
        d = json.load(open(path))
        e = d["kernels"].get(name)
        if e is not None:
            e = dict(e, collected_at=d.get("collected_at"), source="profiles/" + os.path.basename(path))
        return e
    except Exception:
        return None


def graph_time_ms(fn, R=20):
    """average duration of ``fn`` (launches on the current stream) replayed R times as a HIP graph between two HIP events"""
    stream = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    side.wait_stream(stream)
    with torch.cuda.stream(side):
        fn(side.cuda_stream)
    stream.wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with capture_graph(g):
        fn(torch.cuda.current_stream().cuda_stream)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(R):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / R


def deep_roofline(st, dtype):
    """The persistent deep-level launch on its own: [poison its tensors, launch] replayed as a HIP graph minus the
    poisoning replayed alone.  Algorithmic bytes = every phase's weights once + its input and output activations once
    (DESIGN.md section 5); inputs are whatever the last step left in the plan's buffers (no data-dependent control flow)."""
    plan = st.plan
    prog = plan.deep
    deep = [op for op in plan.ops if getattr(op, "kind", "") == "deep" and getattr(op, "prog", prog) is prog]
    if not deep:
        return None
    op = deep[0]

    def zero(s):
        prog.poison(s)

    def both(s):
        prog.poison(s)
        op(s)
    t_zero = graph_time_ms(zero)
    t_both = graph_time_ms(both)
    ms = t_both - t_zero
    e_ = prog.error()
    assert e_ == 0, f"deep kernel: dependency wait timed out (error word {e_:#x})"
    alg = op.w_bytes + op.act_bytes
    achieved = alg / (ms * 1e-3) / 1e9
    pm = pmc_entry("deep_kernel") or {}
    n_ph = len(prog)
    return {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pm.get("hbm_bytes_per_launch"),
        "kernel": "deep_kernel<%s> (persistent: %d dependent phases in one launch, levels %d..bottom)" % (
            dtype, n_ph, plan.deep_level),
        "launches_per_step": 1, "avg_launch_us": round(ms * 1e3, 1), "phases": n_ph, "us_per_phase": round(ms * 1e3 / n_ph, 2),
        "alg_bytes_per_launch": int(alg), "alg_weight_bytes": int(op.w_bytes), "alg_act_bytes": int(op.act_bytes),
        "executed_gflop_per_launch": round(op.flops / 1e9, 2), "mfma_busy_pct": pm.get("mfma_busy_pct"),
        "pmc_source": pm.get("source"), "pmc_collected_at": pm.get("collected_at"),
    }


def conv_roofline(st, reps=3):
    """Average duration of the fused conv-GEMM launches of one denoiser step: the 190 launches of the step's plan are
    captured alone into a HIP graph (same arguments, same order, the other kernels of the step left out) and the graph is
    replayed between two HIP events on its stream, so the figure is launch-to-launch time in the same replayed regime as
    the timed step (kernel duration + launch boundary, no host time, no event packets between the launches).  The
    per-launch list (``slowest_launches_us``) still comes from an eager pass with an event pair around every launch and
    therefore carries ~3 us of event overhead per entry."""
    plan = st.plan
    stream = torch.cuda.current_stream()
    s = stream.cuda_stream
    FAM = ("conv_gemm", "long")        # launch-per-layer convs and the sample-resident launches that replace them (include/jen1_long.h)
    convs = [op for op in plan.ops if getattr(op, "kind", "") in FAM]
    n = len(convs)
    n_long = sum(1 for op in convs if op.kind == "long")
    long_phases = sum(len(op.prog) for op in convs if op.kind == "long")
    tot_ms = 0.0
    per_op = np.zeros(n)
    for rep in range(reps + 1):
        evs = []
        for op in plan.ops:
            if getattr(op, "kind", "") in FAM:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                op(s)
                e1.record(stream)
                evs.append((e0, e1))
            else:
                op(s)
        torch.cuda.synchronize()
        if rep == 0:
            continue            # first pass warms caches / clocks
        d = np.array([a.elapsed_time(b) for a, b in evs])
        per_op += d
        tot_ms += float(d.sum())
    per_op /= reps
    # the figure that feeds `achieved`: conv-family launches only, as one replayed graph
    # (a sample-resident launch polls tensors that start the step poisoned: the step's poisoning node rides along in the replayed family)
    fam_ops = ([op for op in plan.ops if getattr(op, "kind", "") == "deep_poison"] if n_long else []) + convs
    side = torch.cuda.Stream()
    side.wait_stream(stream)
    with torch.cuda.stream(side):
        for op in fam_ops:
            op(side.cuda_stream)
    stream.wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with capture_graph(g):
        cs = torch.cuda.current_stream().cuda_stream
        for op in fam_ops:
            op(cs)
    R = 20
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(R):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    eager_ms = tot_ms / reps
    tot_ms = e0.elapsed_time(e1) / R * reps
    w_bytes = sum(op.w_bytes for op in convs)
    a_bytes = sum(op.act_bytes for op in convs)
    flops = sum(op.flops for op in convs)
    conv_ms = tot_ms / reps
    alg = w_bytes + a_bytes
    achieved = alg / (conv_ms * 1e-3) / 1e9
    top = sorted(range(n), key=lambda i: -per_op[i])[:5]
    if os.environ.get("JEN1_BENCH_OPS"):
        with open(os.environ["JEN1_BENCH_OPS"], "w") as f:
            for i in range(n):
                f.write(f"{per_op[i] * 1e3:8.1f} us  w={convs[i].w_bytes / 1e6:7.2f}MB act={convs[i].act_bytes / 1e6:6.2f}MB  {convs[i].label}\n")
    pm = pmc_entry("conv_family") or {}
    traffic = pm.get("hbm_bytes_per_launch") if pm.get("launches_per_step") == n else None
    mfma_busy = pm.get("mfma_busy_pct") if pm.get("launches_per_step") == n else None
    return {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "mfma_busy_pct": mfma_busy,
        "mfma": {"bound": "mfma", "achieved": round(flops / (conv_ms * 1e-3) / 1e12, 2), "peak": 2500.0, "unit": "TFLOP/s",
                 "frac": round(flops / (conv_ms * 1e-3) / 1e12 / 2500.0, 5)},
        "kernel": (f"long_kernel<*> (include/jen1_long.h): {n_long} sample-resident launches, {long_phases} phases (one convolution each: fused "
                   "GroupNorm+FiLM+SiLU prologue from fixed-order partial statistics + conv implicit GEMM), sample b on workgroups b, b + B, ...; "
                   "timed together with the step's poisoning node" if n_long else
                   "jen1_conv_gemm launches of the levels above the persistent launch: tile_gemm_kernel<*> (fused GroupNorm+FiLM+SiLU "
                   "prologue + conv implicit GEMM) and the few stream_gemm / conv_gemm launches among them"),
        "sample_resident_launches": n_long, "phases": long_phases,
        "us_per_layer": round(conv_ms * 1e3 / max(1, long_phases + (n - n_long)), 2),
        "launches_per_step": n, "avg_launch_us": round(conv_ms * 1e3 / n, 2), "conv_ms_per_step": round(conv_ms, 4),
        "conv_ms_per_step_eager_with_event_pairs": round(eager_ms, 4),
        "alg_bytes_per_step": int(alg), "alg_weight_bytes": int(w_bytes), "alg_act_bytes": int(a_bytes),
        "alg_bytes_per_launch": int(alg / n), "executed_gflop_per_step": round(flops / 1e9, 2),
        "slowest_launches_us": [[convs[i].label.split(" pro=")[0], round(per_op[i] * 1e3, 1)] for i in top],
    }


def kv_gemm_roofline(st):
    """The one large-M matrix-core GEMM of the path (MFMA roofline only where M is large): the text-token K/V
    projection of all cross-attention layers, one grouped launch per conditioning (csrc/big_gemm.hip; blocks.py:402-407, :427-434).
    Executed FLOPs / launch duration (replayed as a HIP graph between two events) against the dense bf16 MFMA peak."""
    ops = [op for op in st.plan.ctx_ops if getattr(op, "kind", "") == "big_gemm"]
    if not ops:
        return None

    def run(s):
        for op in ops:
            op(s)
    ms = graph_time_ms(run)
    fl = sum(op.flops for op in ops)
    tf = fl / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4),
            "kernel": "big_gemm_nt_s4_kernel<272> (four LDS-DMA stages of 256x272 tiles x 32, v_mfma_f32_16x16x32_bf16, row-wise coalesced epilogue; 17408 = 64 x 272 columns: one round of 256 tiles): " + "; ".join(op.label for op in ops),
            "launches": len(ops), "us": round(ms * 1e3, 1), "executed_gflop": round(fl / 1e9, 2),
            "alg_bytes": int(sum(getattr(op, "bytes", 0) for op in ops)),
            "when": "once per conditioning (Plan.set_context), outside the timed region; extra.end_to_end includes it"}


def optimizer_step_bench(n_params, device, reps=5):
    """SURVEY.md section 8 row a16 measured on its own: clip_grad_norm_ + AdamW over the full model's parameter count
    (jen1_grad_sqnorm + jen1_adamw_step on flat float32 buffers).  Algorithmic bytes: the norm reads g (4 B), the update
    reads p, g, m, v and writes p, m, v (28 B) = 32 B per parameter."""
    from jen1_amd.optim import FusedAdamW
    p = torch.nn.Parameter(torch.randn(n_params, device=device) * 0.02)
    opt = FusedAdamW([p], lr=3e-5, betas=(0.9, 0.95), weight_decay=0.1, max_norm=0.7)
    p.grad.normal_(0, 1e-3)
    for _ in range(2):
        opt.step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        opt.step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gbs = 32.0 * n_params / (ms * 1e-3) / 1e9
    return {"what": "clip_grad_norm_(0.7) + AdamW step over %d float32 parameters (row a16)" % n_params, "ms": round(ms, 3),
            "achieved_GBps": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4)}


def train_step_bench(cfg, B, T, dtype, device, reps=5, fwd_flops=None):      # (fwd_flops: unused, kept for callers)
    """BASELINE configs[3] on one GPU (SURVEY.md section 8 rows a14 / a16 / e): one micro-batch of the trainer --
    ``training_loosses`` forward + backward of B clips through the CFG pair (2B rows, gdm.py:245-272, model.py:332-369)
    on the HIP training path, replayed as a HIP graph -- and one clip + AdamW step.  Not the headline metric."""
    from jen1_amd import synth
    from jen1_amd.diffusion import GaussianDiffusion, get_beta_schedule
    from jen1_amd.model import UNetCFG1d
    from jen1_amd.optim import FusedAdamW
    from jen1_amd.train import GraphedLossStep
    model = UNetCFG1d(**cfg, init_seed=1234, compute_dtype=dtype, device=device)
    model.train()
    opt = FusedAdamW(model.parameters())
    graph = model.train_graph(dtype)
    graph.attach_optimizer(opt)
    betas, _ = get_beta_schedule("linear", 1000)
    gd = GaussianDiffusion(steps=1000, betas=betas, objective="noise", loss_type="l2", device=device, cfg_dropout_proba=0.2,
                           embedding_scale=0.8, batch_cfg=True, scale_cfg=True)
    x0 = dev(synth.latents(B, T, key="clip"), device)
    cond = {k: dev(v, device) for k, v in synth.conditioning(B, T, "music_inpaint").items()}
    t = torch.randint(0, 1000, (B,), device=device)
    step = GraphedLossStep(graph, gd, scale=0.1)
    opt.zero_grad()
    step(x0, t, cond, False)
    opt.step()
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    opt.zero_grad()
    e[0].record()
    for _ in range(reps):
        loss = step(x0, t, cond, False)
    e[1].record()
    opt.step()
    e[2].record()
    torch.cuda.synchronize()
    fb = e[0].elapsed_time(e[1]) / reps
    # the pass's OWN launch list, counted in one eager pass outside the timed region: executed FLOPs and algorithmic bytes of every GEMM
    # launch (forward, data gradient, weight gradient; the text-context to_kv products run here, they are not hoisted in training), split
    # by kernel family.  There is no vendor GEMM on the path (profiles/r04_train_step_kernel_stats.txt: no Cijk_* kernels).
    graph.rt.stats = {}
    opt.zero_grad()
    lc = gd.training_loosses(graph, x0, t, cond, causal=False)
    (lc * 0.1).backward()
    torch.cuda.synchronize()
    counted, graph.rt.stats = graph.rt.stats, None
    fl = sum(v[1] for v in counted.values())
    by = sum(v[2] for v in counted.values())
    mfma = None
    if dtype == "bf16":
        tf = fl / (fb * 1e-3) / 1e12
        busy = tp = None
        try:
            tp = latest_profile("train_pmc.json")
            busy = json.load(open(tp))["kernels"]["train_gemm"]["mfma_busy_pct"]
        except Exception:
            pass
        mfma = {"bound": "mfma", "achieved": round(tf, 2), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 5),
                "kernel": "the pass's GEMM launches: train_gemm_* (conv / linear forward, data and weight gradients) and big_gemm_nt / big_gemm_tn "
                          "(text-context to_kv); own kernels only, vendor GEMMs: 0",
                "executed_gflop_per_pass": round(fl / 1e9, 1),
                "by_family": {k: {"launches": v[0], "gflop": round(v[1] / 1e9, 1), "alg_MB": round(v[2] / 1e6, 1)} for k, v in counted.items()},
                "hbm": {"bound": "hbm", "alg_bytes_per_pass": int(by), "achieved": round(by / (fb * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(by / (fb * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                "mfma_busy_pct": busy, "pmc_source": None if busy is None else "profiles/" + os.path.basename(tp)}
    return {"what": f"configs[3] per-GPU shape: forward + backward of {B} clips x 128x{T} through the CFG pair (2B rows), hipGraph replay",
            "fwd_bwd_ms": round(fb, 2), "roofline_mfma": mfma, "clips_per_s": round(B / fb * 1e3, 1), "optimizer_ms": round(e[1].elapsed_time(e[2]), 2),
            "loss": round(float(loss), 4), "grad_allreduce_bytes": 4 * opt.numel,
            "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}


def encodec_decode_bench(B, T, dtype, device):
    """SURVEY.md section 8 f1, the step after the sampler (generation.py:130): SEANet decoder of Encodec 48 kHz on the HIP
    kernels, B latents 128xT -> B x 2 x 320 T samples, synthetic weights (the checkpoint is not available offline)."""
    from jen1_amd.encodec import SEANetDecoderHIP
    from jen1_amd.init_fill import fill
    import json as _json
    sch = _json.loads(str(np.load(os.path.join(ROOT, "tests", "golden", "encodec.npz"))["schema"]))
    dec = SEANetDecoderHIP({k: torch.from_numpy(fill("encodec.decoder." + k, tuple(sh), 1234)) for k, sh in sch}, compute_dtype=dtype, device=device)
    emb = torch.randn((B, 128, T), device=device)
    dec(emb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    y = dec(emb)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"what": f"SEANet decoder, {B} x 128x{T} latents -> {tuple(y.shape)} samples", "ms": round(dt * 1e3, 1),
           "audio_seconds_per_second": round(B * y.shape[-1] / 48000 / dt, 1)}
    # the way in (generation.py:146, dataloader.py:106-114): EncodecModel.encode's segment loop + get_emb's RVQ decode
    from jen1_amd.encodec import EncodecHIP, ResidualVectorQuantizerHIP, SEANetEncoderHIP
    from jen1_amd.init_fill import fill_normal
    esch = _json.loads(str(np.load(os.path.join(ROOT, "tests", "golden", "encodec.npz"))["enc_schema"]))
    enc = SEANetEncoderHIP({k: torch.from_numpy(fill("encodec.encoder." + k, tuple(sh), 1234)) for k, sh in esch}, compute_dtype=dtype, device=device)
    quant = ResidualVectorQuantizerHIP(torch.from_numpy(np.stack([fill_normal(f"encodec.quantizer.layers.{i}.codebook.embed", (1024, 128), 1234)
                                                                   for i in range(16)])), device=device)
    model = EncodecHIP(dec, quant, encoder=enc)
    audio = torch.randn((B, 2, 320 * T), device=device) * 0.1

    def get_emb():
        frames = model.encode(audio)
        return quant.decode(torch.cat([f[0] for f in frames], dim=-1).transpose(0, 1))
    get_emb()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e = get_emb()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["encode"] = {"what": f"encode + quantizer.decode (get_emb), {tuple(audio.shape)} samples -> {tuple(e.shape)} latents, 16 codebooks",
                     "ms": round(dt * 1e3, 1), "audio_seconds_per_second": round(B * audio.shape[-1] / 48000 / dt, 1)}
    return out


def golden_parity(cfg, device, model_bf16=None):
    """max-abs / max-ref of ONE forward of the bench workload (B = 8, 128 x 1500, full model) against the reference's own output on the same
    inputs and weights (tests/golden/full_bench.npz, written by tests/golden/make_golden.py from the unmodified reference; every 16th frame):
    the float32 mode is the mode of BASELINE's 1e-3 gate, bf16 is the benched dtype.  A fixture is data: nothing under oracle/ runs here."""
    from jen1_amd import synth
    from jen1_amd.model import UNetCFG1d
    path = os.path.join(ROOT, "tests", "golden", "full_bench.npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    B, T = 8, 1500
    x, c = synth.latents(B, T), synth.conditioning(B, T)
    want = g["B8.y.nocfg"]
    out = {"what": "UNetCFG1d.forward, B=8, 128x1500, against the reference's output (tests/golden/full_bench.npz); max-abs/max-ref", "tol_f32": 1e-3}
    want_cfg = g["B8.y.cfg"]
    for mode in ("f32", "bf16"):
        m = model_bf16 if (mode == "bf16" and model_bf16 is not None) else UNetCFG1d(**cfg, init_seed=1234, compute_dtype=mode, device=device)
        kw = dict(embedding=dev(c["cross_attn_cond"], device), embedding_mask=dev(c["cross_attn_masks"], device),
                  channels_list=[dev(c["input_concat_cond"], device)], causal=False)
        y = m(dev(x, device), dev(g["B8.t"], device), embedding_scale=1.0, **kw)
        torch.cuda.synchronize()
        m.check_errors()
        out[mode] = float(f"{float(np.abs(y.cpu().numpy()[:, :, ::16] - want).max() / np.abs(want).max()):.3e}")
        # configs[2]: the CFG pair (2B = 16) + rescale of the same inputs
        y = m(dev(x, device), dev(g["B8.t"], device), embedding_scale=0.8, batch_cfg=True, scale_cfg=True, **kw)
        torch.cuda.synchronize()
        m.check_errors()
        out[mode + " CFG pair"] = float(f"{float(np.abs(y.cpu().numpy()[:, :, ::16] - want_cfg).max() / np.abs(want_cfg).max()):.3e}")
        del m, y
    out["ok"] = bool(out["f32"] < 1e-3 and out["f32 CFG pair"] < 1e-3)
    return out


def physical_cores():
    """physical cores this process may run on (SMT siblings counted once): (physical id, core id) pairs of /proc/cpuinfo
    restricted to the affinity mask"""
    try:
        allowed = os.sched_getaffinity(0)
    except Exception:
        allowed = set(range(os.cpu_count() or 1))
    cores, cur = set(), {}
    try:
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = [x.strip() for x in line.split(":", 1)]
                cur[k] = v
            elif cur:
                if int(cur.get("processor", -1)) in allowed:
                    cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                cur = {}
    except Exception:
        pass
    return max(1, len(cores)) if cores else max(1, len(allowed))


def cpu_baseline(B, T, tiny):
    """The CPU oracle on the host cores: oracle/jen1_oracle_torch.py (restatement of the reference's CPU path on torch's CPU
    tensor library, pinned against the reference's outputs by tests/test_oracle_golden.py), one thread per physical core,
    3 warm-up forwards + at least 5 timed ones on the bench shape."""
    from jen1_amd import synth
    from jen1_amd.config import UNetSpec, full_model_config, tiny_model_config
    from jen1_amd.init_fill import fill
    from oracle import jen1_oracle_torch as OT
    cores = physical_cores()
    cfg = tiny_model_config() if tiny else full_model_config()
    spec = UNetSpec(**cfg)
    net = OT.TorchOracleUNetCFG1d({k: fill(k, s, 1234) for k, s in spec.param_shapes()}, **cfg)
    x, cond = synth.latents(B, T), synth.conditioning(B, T)
    t = np.full((B,), 999, dtype=np.int64)
    kw = dict(embedding=cond["cross_attn_cond"], embedding_mask=cond["cross_attn_masks"], embedding_scale=1.0,
              channels_list=[cond["input_concat_cond"]], causal=False)
    # thread count: one per physical core is the starting point; these layers are small enough that a 128-thread pool spends
    # its time in fork / join, so the count is halved while that is faster (2 forwards each) and the best one is timed
    tried = {}
    n = cores
    while n >= 1:
        torch.set_num_threads(n)
        net(x, t, **kw)
        t0 = time.perf_counter()
        net(x, t, **kw)
        tried[n] = time.perf_counter() - t0
        if len(tried) >= 2 and tried[n] > 1.15 * min(tried.values()) or n <= 4:
            break
        n //= 2
    threads = min(tried, key=tried.get)
    torch.set_num_threads(threads)
    for _ in range(3):
        net(x, t, **kw)                                # warm-up (allocator, thread pool, oneDNN primitive cache)
    times = []
    t_start = time.perf_counter()
    while len(times) < 5 or (time.perf_counter() - t_start < 12.0 and len(times) < 40):
        t0 = time.perf_counter()
        net(x, t, **kw)
        times.append(time.perf_counter() - t0)
    return {"value": round(1.0 / float(np.median(times)), 4), "unit": "denoiser steps/s", "cores": threads, "kind": "port",
            "sample": f"3 warm-up + {len(times)} timed UNetCFG1d forwards (no CFG) at B={B}, T={T}, float32, torch CPU oracle "
                      f"(oracle/jen1_oracle_torch.py) with torch.set_num_threads({threads}), the fastest of "
                      f"{ {k: round(v, 3) for k, v in tried.items()} } s per forward tried from the {cores} physical cores down "
                      f"({os.cpu_count()} logical); median; min {min(times):.3f}s max {max(times):.3f}s"}


def train_mode(args, world, rank, device, dist, barrier):
    """BASELINE configs[3]'s per-GPU shape: one optimiser step of the multi-task trainer (trainer.py:126-213 + train.py:88-89)
    = 8 clips of 128x1500 latents per GPU as 3 / 3 / 2 task sub-batches through the CFG pair (forward + backward on the HIP
    training path), the gradient exchange (mean all-reduce of the 296.5 M float32 gradients over RCCL), clip + AdamW + LinearLR.
    Conditioner outputs are synthetic T5-shaped embeddings resident in HBM; masks, timesteps, noise and CFG-dropout rows are
    drawn inside the timed region as the reference draws them."""
    import random
    from jen1_amd import synth
    from jen1_amd.config import full_model_config, tiny_model_config
    from jen1_amd.diffusion import GaussianDiffusion, get_beta_schedule
    from jen1_amd.model import UNetCFG1d
    from jen1_amd.optim import FusedAdamW, LinearLR
    from jen1_amd.trainer import UnifiedMultiTaskTrainer
    cfg = tiny_model_config() if args.tiny else full_model_config()
    B, T = args.batch, args.length
    model = UNetCFG1d(**cfg, init_seed=1234, compute_dtype=args.dtype, device=device)
    model.train()
    betas, _ = get_beta_schedule("linear", 1000)
    gd = GaussianDiffusion(steps=1000, betas=betas, objective="noise", loss_type="l2", device=device, cfg_dropout_proba=0.2,
                           embedding_scale=0.8, batch_cfg=True, scale_cfg=True)
    opt = FusedAdamW(model.parameters(), lr=3e-5, betas=(0.9, 0.95), weight_decay=0.1, max_norm=0.7)
    sched = LinearLR(3e-5)
    c = synth.conditioning(B, T, "text_guided", seed=rank)
    emb, msk = dev(c["cross_attn_cond"], device), dev(c["cross_attn_masks"], device)

    def conditioner(metadata, device_):
        idx = torch.tensor(metadata, device=device_)
        return {"prompt": (emb[idx], msk[idx])}
    accum = max(1, args.accum)
    tr = UnifiedMultiTaskTrainer.build(model, gd, conditioner, opt, sched, grad_accum_every=accum, rng=random.Random(rank), device=device,
                                 use_graph=not args.eager_train, allow_uneven_tasks=True, grad_dtype="bf16" if args.grad_bf16 else "f32")
    audio = dev(synth.latents(B, T, key="clip", seed=rank), device)
    meta = list(range(B))
    torch.manual_seed(rank)
    # a "step" of this mode is one MICRO-batch (forward + backward of B clips); with --accum A every A-th one also carries the gradient
    # exchange and ends with clip + AdamW + LinearLR (trainer.py:139-149: A = 10 in the reference).  Warm-up and the timed region are whole
    # windows, so the region holds exactly steps / A optimiser steps.
    n_warm = -(-max(args.warmup, 8) // accum) * accum
    n_steps = -(-args.steps // accum) * accum
    for _ in range(n_warm):                   # the merged passes come in four (size, causal) shapes (text_guided flips a coin for
                                              # causal): all of them are captured before the timed region
        loss, _, _ = tr.train_step(audio, meta)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        loss, _, _ = tr.train_step(audio, meta)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    args.steps = n_steps
    dt, steps_per_s = aggregate(dist, dt, args.steps, world, device)
    out = {
        "metric": "training clips/sec (multi-task DDP step, 8 clips x 128x1500 latents per GPU)", "value": round(steps_per_s * B, 2),
        "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": n_warm,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic", "grad_accum_every": accum, "optimizer_steps": n_steps // accum,
        "grad_exchange_dtype": "bf16" if args.grad_bf16 else "f32",
        "config": {"workload": ("configs[0] tiny 1D-UNet" if args.tiny else "configs[3] full JEN-1 1D-UNet (296.5M params)")
                   + f", {B} clips per GPU (3/3/2 over text_guided / music_inpaint / music_cont; all sub-batches share ONE pass per "
                   + f"micro-batch: the causal flag travels per clip), latents 128x{T}, CFG pair, "
                   + ("eager backward with the exchange overlapped" if args.eager_train else
                      "hipGraph-replayed forward+backward; with N > 1 the replayed pass that completes the gradients carries the RCCL all-reduces "
                      "of the finished regions on a forked communication stream")
                   + ", clip 0.7 + AdamW + LinearLR", "global_batch": B * world, "seq_len": T, "parallelism": f"ddp x{world}"},
        "loss": round(float(loss), 4),
    }
    if world > 1:
        # the exchange on its own (blocking form: every region, then wait), the same steps with the exchange switched off, and from the
        # two how much of it the overlapped / recorded form hides behind the backward pass
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(3):
            tr.exchange.blocking()
        torch.cuda.synchronize()
        ex_ms = (time.perf_counter() - t0) / 3 * 1e3
        tr.exchange.disabled = True
        for _ in range(-(-4 // accum) * accum):
            tr.train_step(audio, meta)
        torch.cuda.synchronize()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            tr.train_step(audio, meta)
        torch.cuda.synchronize()
        barrier()
        dt0, _ = aggregate(dist, time.perf_counter() - t0, args.steps, world, device)
        tr.exchange.disabled = False
        ms0 = dt0 / args.steps * 1e3
        exposed = max(0.0, out["ms_per_step"] - ms0)
        seen = torch.zeros(world, dtype=torch.int32, device=device)
        seen[rank] = 1
        dist.all_reduce(seen)
        out["exchange_ms"] = round(ex_ms, 3)
        out["exchange_bytes"] = (2 if args.grad_bf16 else 4) * opt.numel
        out["ms_per_step_without_exchange"] = round(ms0, 3)
        out["exchange_exposed_ms"] = round(exposed, 3)
        out["exchange_overlapped_fraction"] = round(max(0.0, 1.0 - exposed / ex_ms), 3) if ex_ms > 0 else None
        out["exchange_recorded_in_graph"] = bool(tr.exchange.capturable and not args.eager_train)
        out["ranks_seen"] = int(seen.sum().item())
    out["peak_mem_GiB"] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)
    if rank == 0:
        print(json.dumps(out))


def init_dist(world, rank, device, backend="nccl"):
    """one process per GPU; RCCL ("nccl" on ROCm) is used only for the barrier and the max-over-ranks time:
    sampling shards by sample, there is no data-path collective."""
    if world <= 1:
        return None
    import torch.distributed as dist
    kw = {"device_id": torch.device(device)} if backend == "nccl" else {}
    dist.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank, **kw)
    return dist


def aggregate(dist, dt, steps, world, device):
    """whole-job throughput: all ranks' steps / the slowest rank's time"""
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt, world * steps / dt


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (python bench.py --gpus N starts them itself; "
                         "or torchrun --nproc-per-node N bench.py --gpus N)")
    on_gpu = not args.dry_run
    if on_gpu:
        torch.cuda.set_device(local)
    device = f"cuda:{local}" if on_gpu else "cpu"
    dist = init_dist(world, rank, device, backend=args.backend)

    def barrier():
        if dist is not None:
            dist.barrier()

    # every rank that takes part in the run adds 1: the line shows how many processes were actually there
    ranks_seen = 1
    if dist is not None:
        one = torch.ones((1,), dtype=torch.int32, device=device)
        dist.all_reduce(one)
        ranks_seen = int(one.item())
    if args.dry_run:
        if rank == 0:
            print(json.dumps({"metric": "denoiser steps/sec (B=8, 128x1500 latents)", "value": None, "unit": "denoiser steps/s", "n_gpus": world,
                              "ranks_seen": ranks_seen, "scaling": args.scaling, "mode": args.mode, "dry_run": True, "backend": args.backend}))
        barrier()
        if dist is not None:
            dist.destroy_process_group()
        return

    if args.mode == "train":
        train_mode(args, world, rank, device, dist, barrier)
        barrier()
        if dist is not None:
            dist.destroy_process_group()
        return

    from jen1_amd.config import full_model_config, tiny_model_config
    from jen1_amd.model import UNetCFG1d
    cfg = tiny_model_config() if args.tiny else full_model_config()
    B, T = args.batch, args.length
    strong = args.scaling == "strong" and world > 1
    if strong:
        # strong scaling: --batch is the GLOBAL batch (configs[1]: 8 samples), rank r denoises samples [r B / N, (r + 1) B / N) -- still no
        # data-path collective (samples are independent); a step of the job = one step of every sample, so value = steps / slowest rank
        if B % world:
            raise SystemExit(f"bench.py: --scaling strong needs --batch {B} divisible by --gpus {world}")
        B = B // world
    model = UNetCFG1d(**cfg, init_seed=1234, compute_dtype=args.dtype, device=device)
    st = build_stepper(model, B, T, device, cfg_pair=False, use_graph=not args.no_graph)
    # the timed region is EXACTLY --steps steps; it is repeated (>= 3 regions, >= 600 steps in total) and the MEDIAN region is the
    # headline, so one clock hiccup of a 26 ms region cannot move it; every region's ms/step is in the line ("regions")
    reps = args.repeats if args.repeats > 0 else max(3, -(-600 // max(1, args.steps)))
    dts = timed_steps(st, args.steps, args.warmup, barrier, repeats=reps)
    per_region = []
    for d in dts:
        d_, _ = aggregate(dist, d, args.steps, world, device)
        per_region.append(d_)
    dt = float(np.median(per_region))
    value = (1 if strong else world) * args.steps / dt

    out = {
        "metric": "denoiser steps/sec (B=8, 128x1500 latents)", "value": round(value, 2), "unit": "denoiser steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "ranks_seen": ranks_seen, "vs_baseline": None, "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": ("configs[0] tiny 1D-UNet" if args.tiny else "configs[1] full JEN-1 1D-UNet (296.5M params)")
                   + f", B={B} per GPU, latents 128x{T}, 100-step DDIM schedule, eta=1 (the reference's default, gdm.py:28: every step adds "
                     "sigma * noise; the per-step noise is a table drawn before the timed region and READ inside it), no CFG (CFG dropout off: "
                     "sampling), hipGraph-replayed step; per sampling run and outside the timed region: the text K/V projection and the "
                     "time-embedding / FiLM / time-token K/V tables of the schedule (extra.end_to_end times a whole sample() call with them)",
                   "global_batch": B * world, "seq_len": T,
                   "parallelism": (f"global batch {B * world} split over {world} ranks (independent samples, no collective)" if strong
                                   else f"replicas x{world} (independent samples)")},
        "regions": {"count": len(per_region), "steps_each": args.steps, "ms_per_step": [round(d / args.steps * 1e3, 4) for d in per_region],
                    "min": round(min(per_region) / args.steps * 1e3, 4), "median": round(dt / args.steps * 1e3, 4),
                    "max": round(max(per_region) / args.steps * 1e3, 4), "headline": "median"},
    }
    if rank == 0:
        long_levels = conv_roofline(st)
        deep = deep_roofline(st, args.dtype)
        if deep is not None:
            out["roofline"] = deep                       # the dominant kernel of the step
            out["roofline"]["long_levels"] = long_levels  # the fused conv-GEMM launches of the levels above the deep launch
        else:
            out["roofline"] = long_levels
        # the whole step against the HBM roofline: SURVEY.md 8(d)'s algorithmic bytes of one denoiser step (every layer's weights once +
        # its input and output activations once, summed over the plan's launches and phases) / the headline time per step
        es_ = 4 if args.dtype == "f32" else 2
        pl_ = st.plan
        Cx_, Cc_ = model.spec.in_channels, model.spec.ctx_ch0
        # the two kernels at the step's boundary: pack_input reads x and the concat context as float32 [B, C, T] and writes the channel-last
        # network input; the fused CFG / DDIM step reads the network output, x and the noise and writes x
        # (fused sampler step, DDIMStepper.fused_pack: the step kernel writes the latents' channels of the next network input itself -- the
        # concat context is packed once per trajectory -- so the step's own packing traffic is those rows)
        fused_ = bool(getattr(st, "fused_pack", False))
        pack_bytes = (pl_.Beff * T * Cx_ * es_) if fused_ else (B * (Cx_ + Cc_) * T * 4 + pl_.Beff * T * (-(-(Cx_ + Cc_) // 32) * 32) * es_)
        cfg_bytes = pl_.Beff * T * model.spec.out_channels * es_ + B * model.spec.out_channels * T * 12
        step_alg = sum(getattr(op, "w_bytes", 0) + getattr(op, "act_bytes", 0) for op in st.plan.ops) + pack_bytes + cfg_bytes
        step_gbs = step_alg / (dt / args.steps) / 1e9
        out["roofline"]["whole_step"] = {"bound": "hbm", "alg_bytes_per_step": int(step_alg), "of_which_pack_input": int(pack_bytes),
                                         "of_which_cfg_ddim_step": int(cfg_bytes), "ms_per_step": round(dt / args.steps * 1e3, 4),
                                         "achieved": round(step_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(step_gbs / HBM_PEAK_GBS, 4),
                                         "executed_gflop_per_step": round(sum(getattr(op, "flops", 0) for op in st.plan.ops) / 1e9, 2)}
        out["launches_per_step"] = getattr(st, "launches_per_step", st.plan.n_launch + 1)
        out["step_launches"] = ("long (down levels), deep, long (up levels), step_tail (CFG / DDIM update + next input rows + next step's sentinels "
                                "and arena reset), gn_stats_from_parts" if getattr(st, "fused_tail", False) else None)
        if args.dtype == "bf16":
            kvr = kv_gemm_roofline(st)
            if kvr is not None:
                out["roofline"]["to_kv_gemm"] = kvr
        extra = None
        if not args.no_extra:
            st2 = build_stepper(model, B, T, device, cfg_pair=True, use_graph=not args.no_graph)
            dt2 = timed_steps(st2, max(10, args.steps // 2), max(3, args.warmup // 2), lambda: None)
            n2 = max(10, args.steps // 2)
            r2 = conv_roofline(st2)
            extra = {"configs[2] CFG pair (2B=16) + rescale, steps/s": round(n2 / dt2, 2),
                            "ms_per_step": round(dt2 / n2 * 1e3, 4),
                            "roofline_frac": r2["frac"], "alg_bytes_per_step": r2["alg_bytes_per_step"],
                            "conv_ms_per_step": r2["conv_ms_per_step"]}
            # fixed-order statistics (Plan(deterministic=True)): what bit-reproducibility costs on the same workload
            model.deterministic = True
            st_d = build_stepper(model, B, T, device, cfg_pair=False, use_graph=not args.no_graph)
            n_d = max(10, args.steps // 2)
            dt_d = timed_steps(st_d, n_d, max(3, args.warmup // 2), lambda: None)
            extra["deterministic statistics mode, steps/s"] = round(n_d / dt_d, 2)
            extra["deterministic statistics mode, launches_per_step"] = getattr(st_d, "launches_per_step", st_d.plan.n_launch + 1)
            del st_d
            model.deterministic = False
            if not args.tiny and not args.no_graph:
                # first of the extras: HIP maps streams onto four hardware queues in creation order, and every stepper built below
                # creates streams -- measured later, two of the four chains can land on one queue (1 409 -> 819 steps/s)
                extra["concurrent_batches"] = [concurrent_batches_bench(model, B, T, device, n, max(20, args.steps // 2), 5)
                                                      for n in (2, 4)]
            extra["end_to_end"] = end_to_end_bench(model, st, B, T, device)
            if args.dtype != "f32":
                # the parity dtype (tests gate f32 at 1e-3 against the reference) timed on the same workload
                m32 = UNetCFG1d(**cfg, init_seed=1234, compute_dtype="f32", device=device)
                st32 = build_stepper(m32, B, T, device, cfg_pair=False, use_graph=not args.no_graph)
                n32 = max(10, args.steps // 2)
                dt32 = timed_steps(st32, n32, max(3, args.warmup // 2), lambda: None)
                extra["f32 mode (the dtype of the 1e-3 parity gate), steps/s"] = round(n32 / dt32, 2)
                del st32, m32
            if not args.tiny and not args.no_graph:
                # the long levels as tile phases of two more persistent launches (JEN1_TILE_PHASES=1; off by default: DESIGN.md 4b)
                mt = UNetCFG1d(**cfg, init_seed=1234, compute_dtype=args.dtype, device=device)
                mt.engine().use_tile_phases = True
                stt = build_stepper(mt, B, T, device, cfg_pair=False, use_graph=True)
                ntl = max(10, args.steps // 2)
                dtt = timed_steps(stt, ntl, max(3, args.warmup // 2), lambda: None)
                stt.check()
                extra["long levels as tile phases (JEN1_TILE_PHASES=1)"] = {
                    "steps_per_s": round(ntl / dtt, 2), "launches_per_step": getattr(stt, "launches_per_step", stt.plan.n_launch + 1),
                    "programs": [{"phases": len(p_), "tile_phases": p_.kinds.count("tile")} for p_ in stt.plan.progs]}
                del stt, mt
            if not args.tiny:
                extra["optimizer_step"] = optimizer_step_bench(sum(p.numel() for p in model.parameters()), device)
                fwd_flops = sum(getattr(op, "flops", 0) for op in st.plan.ops)
                extra["train_step"] = train_step_bench(cfg, B, T, args.dtype, device, fwd_flops=fwd_flops if args.dtype == "bf16" else None)
                extra["encodec_decode"] = encodec_decode_bench(B, T, args.dtype, device)
                if not args.no_graph:
                    # BASELINE configs[4] shape (long-form continuation / inpaint, T ~ 9000), bf16, no fp8 path yet
                    st5 = build_stepper(model, 1, 9000, device, cfg_pair=True, use_graph=True)
                    n5 = max(10, args.steps // 5)
                    dt5 = timed_steps(st5, n5, 3, lambda: None)
                    extra["configs[4] long-form B=1 T=9000 CFG pair, steps/s"] = round(n5 / dt5, 2)
                    # per-kernel figures of that plan: the tiled conv launches of its long levels (T' = 9000 ... 71) and its
                    # persistent launch (levels 5..8)
                    r5 = conv_roofline(st5)
                    d5 = deep_roofline(st5, args.dtype)
                    extra["configs[4] kernels"] = {
                        "launches_per_step": getattr(st5, "launches_per_step", st5.plan.n_launch + 1),
                        "long_levels": {k: r5[k] for k in ("launches_per_step", "avg_launch_us", "conv_ms_per_step", "alg_bytes_per_step", "achieved",
                                                           "frac", "executed_gflop_per_step", "slowest_launches_us")},
                        "deep_kernel": None if d5 is None else {k: d5[k] for k in ("phases", "avg_launch_us", "us_per_phase", "alg_bytes_per_launch",
                                                                                   "achieved", "frac")}}
                    del st5
                    # configs[4] as named: the JEN1_FP8 mode (e4m3 weights + fp8 matrix-core GEMM / attention units of the persistent
                    # launch, fp8 attention on the launch-per-layer levels; activations and the long levels stay bf16)
                    if args.dtype == "bf16":
                        m8 = UNetCFG1d(**cfg, init_seed=1234, compute_dtype="fp8", device=device)
                        st8 = build_stepper(m8, 1, 9000, device, cfg_pair=True, use_graph=True)
                        dt8 = timed_steps(st8, n5, 3, lambda: None)
                        d8 = deep_roofline(st8, "fp8")
                        e8 = {"what": "configs[4]: B=1, T=9000, CFG pair, JEN1_FP8 mode (e4m3 operands in the persistent launch and in attention)",
                              "steps_per_s": round(n5 / dt8, 2), "ms_per_step": round(dt8 / n5 * 1e3, 4),
                              "roofline": None if d8 is None else {k: d8[k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "phases", "avg_launch_us",
                                                                                      "us_per_phase", "alg_bytes_per_launch", "alg_weight_bytes", "alg_act_bytes")}}
                        del st8
                        st8 = build_stepper(m8, B, T, device, cfg_pair=False, use_graph=True)
                        dt8 = timed_steps(st8, max(10, args.steps // 2), 3, lambda: None)
                        e8["configs[1] shape (B=8, T=1500) in JEN1_FP8 mode, steps/s"] = round(max(10, args.steps // 2) / dt8, 2)
                        d8 = deep_roofline(st8, "fp8")
                        if d8 is not None:
                            e8["configs[1] shape deep_kernel<fp8>"] = {k: d8[k] for k in ("avg_launch_us", "us_per_phase", "alg_bytes_per_launch", "achieved", "frac")}
                        extra["configs[4] fp8"] = e8
                        del st8, m8
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(B, T, args.tiny)
        if extra is not None:
            # the numbers a reader wants first, as short top-level keys (the full records follow in "extra", LAST in the line: a log tail that
            # truncates the line loses detail, not these)
            ts = extra.get("train_step")
            if ts is not None:
                rm = ts.get("roofline_mfma") or {}
                out["train"] = {"what": "configs[3] per-GPU shape, 8 clips, fwd+bwd through the CFG pair, hipGraph replay (1 GPU)", "fwd_bwd_ms": ts["fwd_bwd_ms"],
                                "clips_per_s": ts["clips_per_s"], "optimizer_ms": ts["optimizer_ms"], "mfma_frac": rm.get("frac"),
                                "hbm_frac": (rm.get("hbm") or {}).get("frac")}
            cf = {"configs[2] CFG pair 2B=16, steps/s": extra.get("configs[2] CFG pair (2B=16) + rescale, steps/s"),
                  "configs[4] B=1 T=9000 CFG pair bf16, steps/s": extra.get("configs[4] long-form B=1 T=9000 CFG pair, steps/s")}
            f8 = extra.get("configs[4] fp8")
            if f8 is not None:
                cf["configs[4] B=1 T=9000 CFG pair JEN1_FP8, steps/s"] = f8["steps_per_s"]
                cf["fp8 note"] = "latency-bound chain: fp8 halves weight bytes of a launch that sits at ~8 % of the HBM roofline; no gain expected, none measured"
            e2e = extra.get("end_to_end")
            if isinstance(e2e, dict):
                cf["configs[1] end to end through sample() incl. set-up, steps/s"] = e2e.get("steps_per_s_end_to_end")
            out["configs"] = cf
            if not args.tiny:
                out["parity"] = golden_parity(cfg, device, model if args.dtype == "bf16" else None)
            out["extra"] = extra
        print(json.dumps(out))
    barrier()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
