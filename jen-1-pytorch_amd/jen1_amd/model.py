"""``UNetCFG1d``: drop-in for /root/reference/jen1/model/model.py:268-376.

Same constructor kwargs, same ``state_dict`` keys, same
``forward`` signature and keyword names (the call sites are
jen1/diffusion/gdm/gdm.py:118-125 and :251-258).  The arithmetic runs in
libjen1_hip.so through ``engine.Engine``; this module only routes tensors.

``compute_dtype``: "f32" is the parity mode (fp32 storage, exact-fp32 MFMA; the
1e-3 gate of BASELINE.json is checked in this mode), "bf16" the fast mode
(bf16 storage, fp32 accumulate).
"""
from __future__ import annotations

import weakref
from typing import Dict, Optional, Sequence

import torch
import torch.nn as nn

from . import lib as L
from .config import UNetSpec
from .engine import Engine, Plan
from .init_fill import fill


class _Node(nn.Module):
    """parameter container; nested nodes reproduce the reference's dotted key names."""


def _register(root: nn.Module, key: str, value: torch.Tensor):
    parts = key.split(".")
    m = root
    for p in parts[:-1]:
        if not hasattr(m, p):
            m.add_module(p, _Node())
        m = getattr(m, p)
    m.register_parameter(parts[-1], nn.Parameter(value, requires_grad=True))


def _torch_default_init(key: str, shape, shapes) -> torch.Tensor:
    """what the reference's freshly constructed modules hold for this parameter (torch defaults): nn.Conv1d / ConvTranspose1d / Linear
    weight kaiming_uniform_(a = sqrt 5) = U(+-1 / sqrt(fan_in)) with fan_in = shape[1] * kernel, their bias U(+-1 / sqrt(fan_in));
    GroupNorm / LayerNorm weight 1, bias 0; nn.Embedding and the learned Fourier frequencies N(0, 1)"""
    if key.endswith(".weights") or key.endswith("embedding.weight"):
        return torch.randn(shape)
    if key.endswith(".weight"):
        if len(shape) == 1:
            return torch.ones(shape)
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        bound = 1.0 / fan_in ** 0.5
        return torch.empty(shape).uniform_(-bound, bound)
    if key.endswith(".bias"):
        w = shapes.get(key[:-4] + "weight")
        if w is None or len(w) == 1:
            return torch.zeros(shape)
        fan_in = 1
        for d in w[1:]:
            fan_in *= d
        bound = 1.0 / fan_in ** 0.5
        return torch.empty(shape).uniform_(-bound, bound)
    return torch.randn(shape)


_STEPPER_CACHES = weakref.WeakSet()          # the diffusion objects that cache fused steppers (diffusion.GaussianDiffusion.stepper, vdm.VDM)


def _drop_steppers_of(model) -> None:
    """steppers hold the packed weights, plan buffers, a captured graph and the [S][B][C][T] noise table of an engine (~614 MB at
    B = 8, T = 1500 plus 0.6 GB of weights): when the engine they were built on is dropped they are dead weight, release them now instead
    of one by one when the cache wraps"""
    for gd in list(_STEPPER_CACHES):
        cache = gd.__dict__.get("_steppers")
        if cache:
            for k in [k for k, st in cache.items() if st.model is model]:
                cache.pop(k)


class UNetCFG1d(nn.Module):
    """UNet1d with classifier-free guidance on MI355X (reference model.py:268)."""

    def __init__(self, context_embedding_max_length: int, context_embedding_features: int,
                 use_xattn_time: bool = False, *, compute_dtype: str = "bf16", device="cuda",
                 init_seed="torch", **kwargs):
        """``init_seed``: "torch" (default) = the distributions torch's own modules start from in the reference (model.py builds
        nn.Conv1d / nn.Linear / nn.GroupNorm / nn.LayerNorm / nn.Embedding: kaiming-uniform weights and fan-in-bounded biases, unit
        norm weights and zero norm biases, N(0, 1) embeddings), drawn from torch's global generator; an int = the deterministic
        test filler ``init_fill.fill(key, shape, seed)`` (perturbed norm affines: what the parity tests, the oracle and bench.py
        use); None = zeros (a shell for ``load_state_dict``)."""
        super().__init__()
        self.spec = UNetSpec(context_embedding_max_length=context_embedding_max_length,
                             context_embedding_features=context_embedding_features,
                             use_xattn_time=use_xattn_time, **kwargs)
        self.compute_dtype = compute_dtype
        self._device = torch.device(device)
        shapes = dict(self.spec.param_shapes())
        for key, shape in self.spec.param_shapes():
            if init_seed is None:
                v = torch.zeros(shape, dtype=torch.float32)
            elif isinstance(init_seed, str):
                assert init_seed == "torch", init_seed
                v = _torch_default_init(key, tuple(shape), shapes)
            else:
                v = torch.from_numpy(fill(key, shape, init_seed))
            _register(self, key, v.to(self._device))
        self._engine: Optional[Engine] = None
        self._deterministic: Optional[bool] = None     # None: the engine's default (env JEN1_DETERMINISTIC)
        self._train_graph = None
        self._ctx_key = None
        self._handle: Optional[int] = None             # torch.ops.jen1.unet_cfg_forward's handle of this module (jen1_amd/ops.py)
        self._pending_err = None                       # [plan, pinned word, event, armed]: the asynchronous error check of forward
        self.strict_errors = False                     # True: every forward synchronises and raises for its own launch
        self.register_load_state_dict_post_hook(lambda m, k: m._invalidate())

    # ------------------------------------------------------------------ plumbing
    def _invalidate_engine(self):
        """the parameters changed in place (an optimiser step): the inference engine's packed weights, its plans and their
        captured graphs are stale; the next ``engine()`` call packs again.  An error word that is still on its way to the host is
        looked at first: a timed-out launch is reported (``Jen1HipError``) even when an optimiser step comes between the call and the
        next one; the engine is dropped either way"""
        try:
            self.check_errors()
        finally:
            self._pending_err = None
            self._engine = None
            self._ctx_key = None
            _drop_steppers_of(self)

    def _invalidate(self):
        self._invalidate_engine()
        if self._train_graph is not None:
            self._train_graph.invalidate()

    def attach_optimizer(self, opt) -> None:
        """for optimisers driven outside ``TrainGraph`` / the trainer: every ``FusedAdamW.step`` invalidates the packed weights"""
        opt.post_step_hooks.append(self._invalidate)

    def train_graph(self, compute_dtype: Optional[str] = None):
        """The differentiable forward of this module (jen1_amd/train.py): a callable with ``forward``'s signature
        whose backward fills ``param.grad``.  ``GaussianDiffusion.training_loosses`` picks it up by itself when
        gradients are enabled and the module is in training mode (trainer.py:194, :204-208)."""
        from .train import TrainGraph
        cd = compute_dtype or self.compute_dtype
        cd = "bf16" if cd == "fp8" else cd          # (JEN1_FP8 is a sampling mode; training computes in bf16 / float32 master weights)
        if self._train_graph is None or self._train_graph.compute_dtype != cd:
            self._train_graph = TrainGraph(self, self.spec, cd, self._device)
        return self._train_graph

    def engine(self) -> Engine:
        if self._engine is None:
            L.load()   # raises when the HIP extension is missing: there is no fallback
            if self._device.type != "cuda":
                raise L.Jen1HipError("UNetCFG1d needs a ROCm GPU (device='cuda'); no CPU path exists in this package")
            self._engine = Engine(self.spec, {k: v for k, v in self.state_dict().items()}, self.compute_dtype, self._device)
            if self._deterministic is not None:
                self._engine.deterministic = self._deterministic
        return self._engine

    @property
    def deterministic(self) -> bool:
        """fixed-order GroupNorm / LayerNorm statistics in every plan built from now on (Plan(deterministic=True)): two runs of
        the same inputs are bit-identical; the launch-per-layer levels pay one statistics launch per normalised tensor"""
        return bool(self._deterministic) if self._engine is None else self._engine.deterministic

    @deterministic.setter
    def deterministic(self, on: bool) -> None:
        self._deterministic = bool(on)
        if self._engine is not None:
            self._engine.deterministic = bool(on)

    def repack(self):
        """Re-pack weights after an optimiser step / in-place parameter change."""
        self._invalidate()

    def _stream(self) -> int:
        return torch.cuda.current_stream(self._device).cuda_stream

    def _prepare(self, plan: Plan, x, time, embedding, embedding_mask, channels_list, drop_rows, uncond_only=False):
        plan.x_in.copy_(x.to(torch.float32))
        if not plan.table_mode:
            plan.set_times(time)
        if self.spec.ctx_ch0:
            assert channels_list is not None and channels_list[0] is not None, "Missing context"   # model.py:189
            ch = channels_list[0]
            assert ch.shape[1] == self.spec.ctx_ch0, f"Expected context with {self.spec.ctx_ch0} channels at idx 0"
            plan.ctx_in.copy_(ch.to(torch.float32))
        # the text K/V cache is keyed on the identity (weakref) + in-place version of the tensors
        k = getattr(plan, "_ctx_key", None)
        hit = (k is not None and k[1]() is embedding and k[2] == embedding._version and
               ((embedding_mask is None and k[3] is None) or
                (embedding_mask is not None and k[3] is not None and k[3]() is embedding_mask and k[4] == embedding_mask._version)))
        if not hit:
            plan.set_context(embedding, embedding_mask, self._stream())
            plan._ctx_key = (plan, weakref.ref(embedding), embedding._version,
                             None if embedding_mask is None else weakref.ref(embedding_mask),
                             None if embedding_mask is None else embedding_mask._version)
        plan.set_rows(drop_rows, uncond_only)

    def _check_deep(self, plan: Plan) -> None:
        """the persistent deep-level launch replaces a hang by an error word (a dependency wait that timed out: its workgroups
        were not all resident, e.g. another persistent launch held CUs); the results are garbage then and must not be used.
        ``forward`` does NOT synchronise for it: the word is copied to pinned host memory behind the call's launches and looked at
        by the next ``forward`` / ``check_errors()`` once that copy has completed (a literal sampler loop or ``eval`` over many
        calls stays asynchronous); ``check_errors()`` is the explicit, synchronising form and what ``strict_errors`` runs per call.
        The fused sampler checks once per sampling run (DDIMStepper.check)."""
        if not getattr(plan, "progs", None) or torch.cuda.is_current_stream_capturing():
            return
        if self.strict_errors:
            self._raise_deep(plan, plan.take_error())
            return
        pend = self._pending_err
        if pend is None or pend[0] is not plan:
            self.check_errors()
            pend = self._pending_err = [plan, torch.zeros(1, dtype=torch.int32).pin_memory(), torch.cuda.Event(), False]
        pend[1].copy_(plan.progs[0].err[:1], non_blocking=True)
        pend[2].record(torch.cuda.current_stream(self._device))
        pend[3] = True

    def _raise_deep(self, plan: Plan, e: int) -> None:
        if e & 0x40000000:
            raise L.Jen1HipError(f"sample-resident long-level launch: workgroup {e & 0xffff} does not run on XCD {e & 7} (the XCD-local exchange "
                                 "assumes workgroup i on XCD i % 8); set JEN1_LONG_LOCAL=0 to write the exchange through")
        if e:
            raise L.Jen1HipError(f"persistent launch: the wait for phase {e - 1} timed out (another persistent "
                                 "launch on the same GPU?); the error word was cleared, the call can be repeated")

    def _poll_errors(self) -> None:
        """non-blocking: raise for an earlier call whose error word has arrived on the host"""
        pend = self._pending_err
        if pend is not None and pend[3] and pend[2].query():
            pend[3] = False
            e = int(pend[1][0])
            if e:
                pend[0].take_error()           # clears the device word: the next launch is clean
                self._raise_deep(pend[0], e)

    def check_errors(self) -> None:
        """synchronising form: raises ``Jen1HipError`` if any earlier ``forward`` of this module timed out inside its persistent launch"""
        pend = self._pending_err
        if pend is not None and pend[3]:
            pend[2].synchronize()
            self._poll_errors()

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, x: torch.Tensor, time: torch.Tensor, *, embedding: torch.Tensor,
                embedding_mask: Optional[torch.Tensor] = None, embedding_scale: float = 1.0,
                embedding_mask_proba: float = 0.0, batch_cfg: bool = False, scale_cfg: bool = False,
                scale_phi: float = 0.7, features=None, channels_list: Optional[Sequence[torch.Tensor]] = None,
                causal: Optional[bool] = False, dropout_rows: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Same contract as the reference forward (model.py:299-376); returns a fresh
        float32 [B, out_channels, T] tensor.  ``dropout_rows`` (bool[B]) optionally injects
        the CFG-dropout draw that the reference takes from ``rand_bool`` (model.py:325).
        The call goes through the dispatcher as ``torch.ops.jen1.unet_cfg_forward`` (jen1_amd/ops.py)."""
        assert features is None, "context_features is unused on the JEN-1 path"
        from . import ops as _ops  # noqa: F401  (registers torch.ops.jen1.*)
        if self._handle is None:
            self._handle = _ops.register_model(self)
        ctx = None
        if self.spec.ctx_ch0:
            assert channels_list is not None and channels_list[0] is not None, "Missing context"   # model.py:189
            ctx = channels_list[0]
        return torch.ops.jen1.unet_cfg_forward(self._handle, x, time, embedding, embedding_mask, ctx, float(embedding_scale),
                                               float(embedding_mask_proba), bool(batch_cfg), bool(scale_cfg), float(scale_phi), bool(causal),
                                               dropout_rows)

    @torch.no_grad()
    def _forward_impl(self, x: torch.Tensor, time: torch.Tensor, *, embedding: torch.Tensor,
                      embedding_mask: Optional[torch.Tensor] = None, embedding_scale: float = 1.0,
                      embedding_mask_proba: float = 0.0, batch_cfg: bool = False, scale_cfg: bool = False,
                      scale_phi: float = 0.7, channels_list: Optional[Sequence[torch.Tensor]] = None,
                      causal: Optional[bool] = False, dropout_rows: Optional[torch.Tensor] = None) -> torch.Tensor:
        """what ``torch.ops.jen1.unet_cfg_forward`` runs: plans + launches on the engine"""
        eng = self.engine()
        lib = eng.lib
        self._poll_errors()
        B, _, T = x.shape
        causal = bool(causal)
        drop = None
        if embedding_mask_proba > 0.0:
            if dropout_rows is not None:
                drop = dropout_rows
            elif embedding_mask_proba >= 1.0:
                drop = torch.ones(B, dtype=torch.bool, device=self._device)
            else:   # rand_bool (utils/module.py:36-42)
                drop = torch.bernoulli(torch.full((B,), float(embedding_mask_proba), device=self._device)).to(torch.bool)
        Co = self.spec.out_channels
        out = torch.empty((B, Co, T), dtype=torch.float32, device=self._device)
        s = self._stream()
        if embedding_scale != 1.0:
            if batch_cfg:
                plan = eng.plan(B, T, 2, causal)
                self._prepare(plan, x, time, embedding, embedding_mask, channels_list, drop)
                plan.run(s)
                net = plan.net_out
                self._check_deep(plan)
            else:
                plan = eng.plan(B, T, 1, causal)
                both = torch.empty((2 * B, T, plan.net_out.ld), dtype=eng.tdtype, device=self._device)
                self._prepare(plan, x, time, embedding, embedding_mask, channels_list, drop)
                plan.run(s)
                both[:B].copy_(plan.net_out.t)
                plan.set_rows(None, uncond_only=True)
                plan.run(s)
                both[B:].copy_(plan.net_out.t)
                self._check_deep(plan)
                net = type(plan.net_out)(both, 2 * B, T, Co, plan.net_out.ld)
            L.check(lib.jen1_cfg_combine(net.t.data_ptr(), out.data_ptr(), B, Co, T, net.ld, float(embedding_scale),
                                         1 if scale_cfg else 0, float(scale_phi), eng.dt, s), "jen1_cfg_combine")
            return out
        plan = eng.plan(B, T, 1, causal)
        self._prepare(plan, x, time, embedding, embedding_mask, channels_list, drop)
        plan.run(s)
        L.check(lib.jen1_unpack_output(plan.net_out.t.data_ptr(), out.data_ptr(), B, Co, T, plan.net_out.ld, eng.dt, s),
                "jen1_unpack_output")
        self._check_deep(plan)
        return out
