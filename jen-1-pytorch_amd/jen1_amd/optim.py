"""Optimiser step and data-parallel gradient exchange of the trainer (SURVEY.md section 8 rows a16 and e).

Host-side mirror of /root/reference/train.py:56-60, :84 and /root/reference/trainer.py:139-150:
``torch.optim.AdamW(lr=3e-5, betas=(0.9, 0.95), weight_decay=0.1)`` over ONE parameter group,
``nn.utils.clip_grad_norm_(parameters, 0.7)``, ``LinearLR`` (factor 1/3 -> 1 over 5 optimiser steps) and, under DDP
(train.py:88-89), the mean all-reduce of the gradients.  Parameters, gradients and both moments live in flat float32
buffers (the model's parameters become views of the flat buffer), so clip + AdamW is two HIP launches over 296.5 M
elements (jen1_grad_sqnorm, jen1_adamw_step) with no host synchronisation, and the gradient exchange is a few large
RCCL all-reduces over xGMI instead of 979 small ones.  The gradients come from jen1_amd/train.py, whose kernels
accumulate straight into ``flat_grad`` through the ``p.grad`` views.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch

from . import lib as L


class LinearLR:
    """torch.optim.lr_scheduler.LinearLR defaults (train.py:84): factor start_factor -> 1 over total_iters steps."""

    def __init__(self, base_lr: float, start_factor: float = 1.0 / 3, end_factor: float = 1.0, total_iters: int = 5, last_epoch: int = -1):
        self.base_lr, self.start_factor, self.end_factor, self.total_iters = base_lr, start_factor, end_factor, total_iters
        self.last_epoch = last_epoch + 1

    def factor(self) -> float:
        t = min(self.last_epoch, self.total_iters)
        return self.start_factor + (self.end_factor - self.start_factor) * t / self.total_iters

    def get_last_lr(self) -> float:
        return self.base_lr * self.factor()

    def step(self) -> None:
        self.last_epoch += 1


class FusedAdamW:
    """AdamW + global-norm clipping over flat buffers.  ``params``: the model's parameters (any device for construction;
    ``step`` needs the GPU).  After construction every ``p.data`` is a view of ``flat_param`` and every ``p.grad`` a view
    of ``flat_grad`` (so autograd, or a test, writes gradients straight into the flat buffer)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 3e-5, betas=(0.9, 0.95), eps: float = 1e-8,
                 weight_decay: float = 0.1, max_norm: Optional[float] = 0.7, skip_nonfinite: bool = False):
        self.params: List[torch.nn.Parameter] = [p for p in params]
        assert self.params and all(p.dtype == torch.float32 for p in self.params), "float32 master parameters (train.py:56)"
        dev = self.params[0].device
        # every tensor starts on a 16-byte boundary of the flat buffer (float4 access in the kernels)
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4
        self.numel = n
        self.flat_param = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, self.offsets):
            self.flat_param[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.flat_param[o:o + p.numel()].view_as(p)
            p.grad = self.flat_grad[o:o + p.numel()].view_as(p)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_norm, self.skip_nonfinite = max_norm, skip_nonfinite
        # the number of steps TAKEN lives on the device (jen1_adamw_step_counted): the kernel derives the bias corrections from it
        # and a step dropped by ``skip_nonfinite`` does not advance it -- GradScaler.step semantics (trainer.py:146)
        self._steps = torch.zeros(1, dtype=torch.int32, device=dev)
        self._gnorm_sq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._sq_scratch: Optional[torch.Tensor] = None       # block partials + arrival ticket of the norm, owned by this optimiser
        self.post_step_hooks = []      # callables run after every step (TrainGraph.invalidate: re-pack the compute weights)

    @property
    def step_count(self) -> int:
        """optimiser steps taken so far (one host sync; checkpoints and tests read it, the training loop does not)"""
        return int(self._steps.item())

    @step_count.setter
    def step_count(self, n: int) -> None:
        self._steps.fill_(int(n))

    def zero_grad(self) -> None:
        g = self.flat_grad
        if g.is_cuda:
            L.check(L.load().jen1_memset_zero(g.data_ptr(), g.numel() * 4, torch.cuda.current_stream(g.device).cuda_stream), "jen1_memset_zero")
        else:
            g.zero_()

    def grad_norm(self) -> torch.Tensor:
        """total L2 norm of the gradient as a device scalar (what clip_grad_norm_ returns)"""
        return self._gnorm_sq.sqrt()

    def step(self, lr: Optional[float] = None) -> None:
        """clip_grad_norm_ + AdamW.step() (trainer.py:145-147) on the current stream"""
        lib = L.load()
        if self.flat_param.device.type != "cuda":
            raise L.Jen1HipError("FusedAdamW.step needs a ROCm GPU; no CPU path exists in this package")
        s = torch.cuda.current_stream(self.flat_param.device).cuda_stream
        gn = None
        if self.max_norm is not None or self.skip_nonfinite:
            self._gnorm_sq.zero_()
            if self._sq_scratch is None:
                self._sq_scratch = torch.zeros(int(lib.jen1_grad_sqnorm_scratch_bytes()) // 4, dtype=torch.float32, device=self.flat_param.device)
            L.check(lib.jen1_grad_sqnorm_ws(self.flat_grad.data_ptr(), self.numel, self._gnorm_sq.data_ptr(), self._sq_scratch.data_ptr(), s),
                    "jen1_grad_sqnorm_ws")
            gn = self._gnorm_sq.data_ptr()
        L.check(lib.jen1_adamw_step_counted(self.flat_param.data_ptr(), self.flat_grad.data_ptr(), self.exp_avg.data_ptr(),
                                            self.exp_avg_sq.data_ptr(), self.numel, float(self.lr if lr is None else lr), float(self.betas[0]),
                                            float(self.betas[1]), float(self.eps), float(self.weight_decay), self._steps.data_ptr(), gn,
                                            float(self.max_norm or 0.0), 1 if self.skip_nonfinite else 0, s),
                "jen1_adamw_step_counted")
        for h in self.post_step_hooks:
            h()

    def state_dict(self) -> dict:
        """``torch.optim.AdamW.state_dict()``'s schema (what script_util.py:79-90 saves with the reference's optimiser): per-parameter
        ``state[i] = {"step", "exp_avg", "exp_avg_sq"}`` cut out of the flat buffers, ONE entry of ``param_groups`` with the
        hyper-parameters and ``params = [0 .. n-1]`` in ``model.parameters()`` order.  A file written here loads into
        ``torch.optim.AdamW`` over the same parameters and the other way round."""
        proto = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=self.lr, betas=tuple(self.betas), eps=self.eps,
                                  weight_decay=self.weight_decay).state_dict()["param_groups"][0]
        group = dict(proto)
        group["params"] = list(range(len(self.params)))
        state = {}
        step_count = self.step_count
        if step_count > 0:
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                n = p.numel()
                state[i] = {"step": torch.tensor(float(step_count)), "exp_avg": self.exp_avg[o:o + n].view_as(p).clone(),
                            "exp_avg_sq": self.exp_avg_sq[o:o + n].view_as(p).clone()}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd: dict) -> None:
        """accepts the torch AdamW schema (see ``state_dict``; checkpoints of the reference) and the flat format this class
        wrote before (``{"step", "exp_avg", "exp_avg_sq", ...}`` over the padded flat layout)"""
        if "param_groups" not in sd:                               # flat format of earlier checkpoints
            if sd["exp_avg"].numel() != self.numel or sd["exp_avg_sq"].numel() != self.numel:
                raise ValueError(f"flat optimiser state of {sd['exp_avg'].numel()} elements does not fit this model ({self.numel})")
            self.step_count = int(sd["step"])
            self.exp_avg.copy_(sd["exp_avg"])
            self.exp_avg_sq.copy_(sd["exp_avg_sq"])
            return
        groups = sd["param_groups"]
        if len(groups) != 1:
            raise ValueError(f"{len(groups)} parameter groups in the checkpoint; the trainer uses one (train.py:56-60)")
        ids = list(groups[0]["params"])
        if len(ids) != len(self.params):
            raise ValueError(f"the checkpoint's optimiser covers {len(ids)} parameters, this model has {len(self.params)}")
        g = groups[0]
        self.lr = float(g.get("lr", self.lr))
        self.betas = tuple(g.get("betas", self.betas))
        self.eps = float(g.get("eps", self.eps))
        self.weight_decay = float(g.get("weight_decay", self.weight_decay))
        state = sd.get("state", {})
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        steps = set()
        for pos, (p, o) in enumerate(zip(self.params, self.offsets)):
            st = state.get(ids[pos], state.get(str(ids[pos])))
            if st is None:
                continue
            for name, flat in (("exp_avg", self.exp_avg), ("exp_avg_sq", self.exp_avg_sq)):
                v = st[name]
                if tuple(v.shape) != tuple(p.shape):
                    raise ValueError(f"optimiser state {name} of parameter {pos} has shape {tuple(v.shape)}, the parameter {tuple(p.shape)}")
                flat[o:o + p.numel()].copy_(v.reshape(-1))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"parameters at different optimiser steps {sorted(steps)}: one fused step counter cannot represent that")
        self.step_count = steps.pop() if steps else 0


def adopt_optimizer(params, optimizer, lr_scheduler, grad_clip=None, skip_nonfinite: bool = False):
    """What the reference's call site hands the trainer (train.py:56-60, :84, :110-125) -> ``(FusedAdamW, LinearLR | None)``.

    ``optimizer`` may be a ``torch.optim.AdamW`` over ONE parameter group, ``lr_scheduler`` torch's ``LinearLR`` built on it.
    torch's scheduler constructor has already multiplied ``param_groups[0]["lr"]`` by ``start_factor`` by the time the trainer
    sees the optimiser, so the configured rate is the group's ``initial_lr`` / the scheduler's ``base_lrs[0]``; moments and step
    count of a resumed optimiser (train.py:63-81) are carried over into the flat buffers."""
    if isinstance(optimizer, torch.optim.Optimizer):
        g = optimizer.param_groups
        assert len(g) == 1, "the trainer optimises one parameter group (train.py:56-60)"
        base_lr = float(g[0].get("initial_lr", g[0]["lr"]))
        if lr_scheduler is not None and getattr(lr_scheduler, "base_lrs", None):
            base_lr = float(lr_scheduler.base_lrs[0])
        torch_state = optimizer.state_dict()
        optimizer = FusedAdamW(list(params), lr=base_lr, betas=tuple(g[0]["betas"]), eps=g[0]["eps"],
                               weight_decay=g[0]["weight_decay"], max_norm=grad_clip, skip_nonfinite=skip_nonfinite)
        if torch_state.get("state"):
            optimizer.load_state_dict(torch_state)
            optimizer.lr = base_lr             # load_state_dict read the (already scheduled) group rate: keep the base rate
    else:
        optimizer.max_norm = grad_clip if grad_clip is not None else optimizer.max_norm
        optimizer.skip_nonfinite = optimizer.skip_nonfinite or skip_nonfinite
    if lr_scheduler is not None and not isinstance(lr_scheduler, LinearLR):
        ls = lr_scheduler                      # torch.optim.lr_scheduler.LinearLR (train.py:84)
        lr_scheduler = LinearLR(optimizer.lr, getattr(ls, "start_factor", 1.0 / 3), getattr(ls, "end_factor", 1.0),
                                getattr(ls, "total_iters", 5), last_epoch=getattr(ls, "last_epoch", 0) - 1)
    return optimizer, lr_scheduler


class GradExchange:
    """DDP's bucketed gradient exchange overlapped with the backward pass (train.py:88-89: ``DDP(model)`` all-reduces a
    bucket as soon as autograd has finished the gradients in it).

    The flat gradient buffer is laid out in forward order (to_mapping, to_time, to_in, downsamples.0 .. 8, bottleneck,
    upsamples.0 .. 8, to_out, to_time_embedding, fixed_embedding), the backward pass finishes it from the back.  A *region* is
    the slice of one top-level block; ``TrainGraph`` hooks the activation that enters each block, and when the gradient with
    respect to that activation exists every parameter gradient of the block is complete (``region_ready``).  A region is
    reduced in chunks of at most ``bucket_bytes`` on a communication stream (few large RCCL messages: xGMI rings are per-link
    bound) while the rest of the backward pass runs; regions that only finish with the pass (the head: time MLPs and to_in; the
    tail: the tiny global embeddings) go out in ``finish``.  Mean over ranks; the result is bit-identical to one blocking
    all-reduce of the whole buffer chunked the same way (same chunks, same reduction)."""

    def __init__(self, opt: "FusedAdamW", names: List[str], group=None, bucket_bytes: int = 128 << 20, params: Optional[dict] = None,
                 grad_dtype: str = "f32", single_rank: Optional[bool] = None):
        """``params``: ``dict(model.named_parameters())`` -- checked against the optimiser's parameter list by identity (an
        optimiser built over a re-ordered or filtered list would mis-assign regions).  ``grad_dtype="bf16"``: every chunk is
        rounded to bfloat16 for the wire and widened again (half the xGMI bytes, two extra passes over the chunk in HBM; the mean
        is then only bf16-accurate) -- off by default, like torch DDP's bf16 compression hook.  ``single_rank`` (default: env
        JEN1_EXCHANGE_SINGLE_RANK=1): run the exchange -- hooks, communication stream, RCCL collectives, recording into the replayed
        graph -- in a process group of ONE rank too, where it is the identity: the whole overlapped / recorded path can then be
        exercised on a single GPU (tests/test_gpu_train.py::test_recorded_rccl_exchange_single_rank)."""
        import torch.distributed as dist
        assert grad_dtype in ("f32", "bf16")
        self.opt, self.group, self.bucket = opt, group, max(1, bucket_bytes // 4)
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.backend = dist.get_backend(group) if dist.is_available() and dist.is_initialized() else None
        assert len(names) == len(opt.params)
        if params is not None:
            for n, p in zip(names, opt.params):
                assert params[n] is p, f"optimiser parameter order differs from the model's at {n!r}"
        self.regions: "dict[str, tuple[int, int]]" = {}
        for n, p, o in zip(names, opt.params, opt.offsets):
            r = ".".join(n.split(".")[:2]) if n.startswith(("downsamples.", "upsamples.")) else n.split(".")[0]
            lo, hi = self.regions.get(r, (o, o))
            self.regions[r] = (min(lo, o), max(hi, o + (p.numel() + 3) // 4 * 4))
        # regions are contiguous slices of the flat buffer: disjoint, and together they cover it (each is reduced exactly once)
        spans = sorted(self.regions.values())
        assert spans[0][0] == 0 and spans[-1][1] == opt.flat_grad.numel() and all(a[1] == b[0] for a, b in zip(spans, spans[1:])), \
            "parameter regions must tile the flat gradient buffer"
        self.bf16 = grad_dtype == "bf16"
        self._wire: Optional[torch.Tensor] = None
        import os
        self.single_rank = (os.environ.get("JEN1_EXCHANGE_SINGLE_RANK", "0") == "1") if single_rank is None else bool(single_rank)
        # cumulative counters (never reset): all-reduce calls issued (eager or recorded), those issued while a graph was being
        # recorded, regions that left during a backward pass (before finish())
        self.collectives = self.recorded_collectives = self.regions_in_pass = 0
        self.active = False
        # debugging aid for one step: instead of sending a region when its hook fires, keep a copy of it; ``finish`` checks that
        # no gradient landed in the region afterwards (the ordering assumption the overlap rests on), then sends everything
        self.debug_check = False
        self._snap: "dict[str, torch.Tensor]" = {}
        self._comm = None
        self._works, self._expect, self._sent = [], {}, set()
        self.sent_during_pass = 0         # regions that left before finish() (what the overlap test and the bench report)

    @property
    def capturable(self) -> bool:
        """RCCL collectives on device buffers can be recorded into the replayed backward pass; gloo / CPU cannot"""
        return self.enabled and self.backend == "nccl" and self.opt.flat_grad.is_cuda

    @property
    def enabled(self) -> bool:
        """more than one rank -- or one rank of an initialised process group when ``single_rank`` asks for the path anyway"""
        return (self.world > 1 or (self.single_rank and self.backend is not None)) and not getattr(self, "disabled", False)

    def begin(self) -> None:
        """arm the exchange for the backward pass(es) that follow (the last micro-batch of an accumulation window)"""
        self.active = self.enabled
        self._works, self._expect, self._sent = [], {}, set()
        self.sent_during_pass = 0
        g = self.opt.flat_grad
        if self.active and g.is_cuda and self._comm is None:
            self._comm = torch.cuda.Stream(g.device)
        if self.active and self.bf16 and self._wire is None:
            self._wire = torch.empty(min(self.bucket, g.numel()), dtype=torch.bfloat16, device=g.device)
        self._snap = {}

    def expect(self, region: str) -> None:
        """a forward pass registered one more hook for ``region`` (several sub-batches share one backward pass)"""
        self._expect[region] = self._expect.get(region, 0) + 1

    def region_ready(self, region: str, also: Optional["torch.cuda.Stream"] = None) -> None:
        """``also``: a second stream gradients of the region were enqueued on (train.TrainRuntime.weight_grad: the queue of weight-gradient launches)"""
        if not self.active or region not in self.regions or region in self._sent:
            return
        left = self._expect.get(region, 1) - 1
        self._expect[region] = left
        if left > 0:
            return
        self.sent_during_pass += 1
        self.regions_in_pass += 1
        if self.debug_check:
            lo, hi = self.regions[region]
            if also is not None:
                torch.cuda.current_stream(self.opt.flat_grad.device).wait_stream(also)
            self._snap[region] = self.opt.flat_grad[lo:hi].clone()
            return
        self._send(region, also)

    def _reduce(self, chunk: torch.Tensor, sync: bool) -> None:
        import torch.distributed as dist
        self.collectives += 1
        if chunk.is_cuda and torch.cuda.is_current_stream_capturing():
            self.recorded_collectives += 1
        if self.bf16:
            w = self._wire[:chunk.numel()]
            w.copy_(chunk)
            dist.all_reduce(w, op=dist.ReduceOp.SUM, group=self.group)       # (the wire buffer is reused: stream-ordered, not async)
            chunk.copy_(w)
        elif sync:
            dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group)
        else:
            self._works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _send(self, region: str, also=None) -> None:
        self._sent.add(region)
        lo, hi = self.regions[region]
        g = self.opt.flat_grad
        if g.is_cuda:
            cur = torch.cuda.current_stream(g.device)
            capturing = torch.cuda.is_current_stream_capturing()
            self._comm.wait_stream(cur)      # the gradients of the region are enqueued before this point
            if also is not None:
                self._comm.wait_stream(also)
            with torch.cuda.stream(self._comm):
                for o in range(lo, hi, self.bucket):
                    # while a graph is being recorded the collective is a node of it: issued in stream order on the communication
                    # stream (forked from the capturing stream above, joined in ``finish``), no Work handle to wait on
                    self._reduce(g[o:min(hi, o + self.bucket)], sync=capturing)
        else:
            for o in range(lo, hi, self.bucket):
                self._reduce(g[o:min(hi, o + self.bucket)], sync=False)

    def finish(self) -> None:
        """after the backward pass: send what is left (in reverse order), wait, turn the sums into means.  Called while a graph is
        being recorded (``GraphedLossStep``) it records the same: the replay then carries the whole exchange."""
        if not self.active:
            return
        for region, snap in self._snap.items():
            lo, hi = self.regions[region]
            if not torch.equal(snap, self.opt.flat_grad[lo:hi]):
                raise RuntimeError(f"GradExchange: gradients of region {region!r} changed after its hook fired")
        self._snap = {}
        for region in reversed(list(self.regions)):
            if region not in self._sent:
                self._send(region)
        for w in self._works:
            w.wait()
        self._works = []
        g = self.opt.flat_grad
        if g.is_cuda:
            torch.cuda.current_stream(g.device).wait_stream(self._comm)
        g.mul_(1.0 / self.world)
        self.active = False

    def done_in_graph(self) -> None:
        """a replayed graph that carries the exchange has been enqueued: nothing is left for ``finish``"""
        self.active = False

    def blocking(self) -> None:
        """the same exchange with no overlap: every region, reverse order, then wait"""
        self.begin()
        self.finish()


def allreduce_gradients(flat_grad: torch.Tensor, group=None, bucket_bytes: int = 256 << 20) -> None:
    """DDP gradient exchange (train.py:88-89): mean over ranks, in place, in a few large buckets (xGMI rings are
    per-link bound: big messages, not 979 small ones).  ``backend="nccl"`` is RCCL on ROCm; gloo on CPU for tests."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return
    world = dist.get_world_size(group)
    if world == 1:
        return
    per = max(1, bucket_bytes // flat_grad.element_size())
    works = []
    for o in range(0, flat_grad.numel(), per):
        works.append(dist.all_reduce(flat_grad[o:o + per], op=dist.ReduceOp.SUM, group=group, async_op=True))
    for w in works:
        w.wait()
    flat_grad.mul_(1.0 / world)
