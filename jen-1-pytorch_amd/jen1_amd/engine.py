"""Execution engine of the denoiser: turns a ``UNetSpec`` + parameters into a static
list of HIP kernel launches (a *plan*) per (batch, length, CFG-pair, causal) shape.

Host-side counterpart of ``UNet1d.forward`` / ``UNetCFG1d.forward``
(/root/reference/jen1/model/model.py:225-265, :299-376) and of every block in
/root/reference/jen1/model/blocks.py.  PyTorch is used for device memory and
streams only; all arithmetic on the path is a jen1_* kernel from libjen1_hip.so.
Buffers and kernel arguments of a plan are allocated once, so replaying a plan
enqueues ~300 launches with fixed pointers: exactly what a hipGraph captures.

Design notes (DESIGN.md has the long form):
  * activations are channel-last [B][L][C]; the reference's pad / cat / crop /
    permute copies disappear into kernel index math;
  * GroupNorm / LayerNorm never run as kernels: producers accumulate the sums in
    their epilogue, consumers normalise in their LDS-staging prologue;
  * cross-attention K/V of the 128 text tokens (and of the learned "fixed"
    embedding used by the unconditional CFG half) are step-invariant: they are
    projected once per conditioning (``set_context``) and only the time-token
    row is projected per step (exact).
"""
from __future__ import annotations

import contextlib
import ctypes as C
import math
import weakref
import os
import time
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import lib as L
from .config import ResSpec, TransformerSpec, UNetSpec
from .packing import (conv_weight_to_gemm, convT_weight_to_gemm, fold_layernorm, pack_gemm_weight)

FG = 32  # fine groups per tensor for GroupNorm statistics


def _ceil_to(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class Act:
    """a channel-last activation [B][L][ld] plus the statistics its consumers need.
    ``ld`` is the row pitch; ``cp`` the channels a GEMM reads (C padded to 32): they differ only for a column
    window of a wider tensor (``cols``)."""
    __slots__ = ("t", "B", "L", "C", "ld", "gn", "rs", "cp", "part", "gn_valid", "lp")

    def __init__(self, t, B, L, C, ld, gn=None, rs=None, cp=None):
        self.t, self.B, self.L, self.C, self.ld, self.gn, self.rs = t, B, L, C, ld, gn, rs
        self.cp = cp if cp is not None else ld
        # GroupNorm statistics as a tile phase of the persistent launch leaves them: (tensor [B][tiles][groups][2], tiles, groups)
        self.part = None
        # ... and as a phase of the sample-resident long-level launch leaves them: (tensor [B][8][entries][2], entries, M blocks, 16-channel sub-groups)
        self.lp = None
        # ``gn`` (the fine-group totals) is written by whoever produces the tensor -- except phases of the persistent launch
        self.gn_valid = True

    def cols(self, c0: int, n: int, rs=None) -> "Act":
        """channels [c0, c0 + n) of this tensor as an activation of its own (same row pitch)"""
        assert c0 % 32 == 0 and n % 32 == 0 and c0 + n <= self.ld
        return Act(self.t[:, :, c0:], self.B, self.L, n, self.ld, None, rs, cp=n)


class Weights:
    """Packed, device-resident parameters in the compute dtype (+ float32 vectors)."""

    def __init__(self, spec: UNetSpec, params: Dict[str, torch.Tensor], dtype: torch.dtype, device):
        self.spec, self.dtype, self.device = spec, dtype, device
        p = {k: v.detach().to(device=device, dtype=torch.float32) for k, v in params.items()}
        self.w: Dict[str, torch.Tensor] = {}
        self.v: Dict[str, torch.Tensor] = {}
        self._fp8: Dict[int, tuple] = {}
        pk = lambda w_tmk: pack_gemm_weight(w_tmk, dtype)
        f32 = lambda t: t.contiguous()

        def rowsum(w_mk):
            """row sums of the weights exactly as the MFMA sees them (rounded to the compute dtype):
            the epilogue form of LayerNorm subtracts mean * rowsum(W') from the accumulator"""
            return w_mk.to(dtype).to(torch.float32).sum(dim=1).contiguous()

        def padvec(t, n):
            return torch.nn.functional.pad(t, (0, n - t.numel())).contiguous() if t.numel() != n else t.contiguous()

        # ---- time MLPs stay float32 (tiny; phases up to 2e4 rad) -------------------------------
        for k in ("to_time.0.0.weights", "to_time.0.1.weight", "to_time.0.1.bias", "to_mapping.0.weight",
                  "to_mapping.0.bias", "to_mapping.2.weight", "to_mapping.2.bias"):
            self.v[k] = f32(p[k])
        if spec.use_xattn_time:
            for k in ("to_time_embedding.0.0.weights", "to_time_embedding.0.1.weight", "to_time_embedding.0.1.bias"):
                self.v[k] = f32(p[k])

        # ---- ResnetBlock1d (blocks.py:168-231) ----------------------------------------------------
        film_w, film_b, self.film_off = [], [], {}
        off = 0
        for r in spec.res_blocks():
            n = r.name
            cin_p = _ceil_to(r.c_in, 32)
            self.w[f"{n}.conv1"] = pk(conv_weight_to_gemm(p[f"{n}.block1.project.conv.weight"]))
            self.v[f"{n}.conv1.bias"] = f32(p[f"{n}.block1.project.conv.bias"])
            self.v[f"{n}.gn1.g"] = padvec(p[f"{n}.block1.groupnorm.weight"], cin_p)
            self.v[f"{n}.gn1.b"] = padvec(p[f"{n}.block1.groupnorm.bias"], cin_p)
            self.w[f"{n}.conv2"] = pk(conv_weight_to_gemm(p[f"{n}.block2.project.conv.weight"]))
            self.v[f"{n}.conv2.bias"] = f32(p[f"{n}.block2.project.conv.bias"])
            self.v[f"{n}.gn2.g"] = f32(p[f"{n}.block2.groupnorm.weight"])
            self.v[f"{n}.gn2.b"] = f32(p[f"{n}.block2.groupnorm.bias"])
            if r.has_shortcut:
                wsc = conv_weight_to_gemm(p[f"{n}.to_out.conv.weight"]).clone()
                if r.c_in == 2 * r.c_out and spec.use_skip_scale:
                    # up-path shortcut reads cat([x, skip * 2^-1/2]) (blocks.py:732-734): the scale is folded
                    # into the skip half of the 1x1 weights so the GEMM can stream both sources unscaled
                    wsc[:, :, r.c_out:] *= 2 ** -0.5
                self.w[f"{n}.short"] = pk(wsc)
                self.v[f"{n}.short.bias"] = f32(p[f"{n}.to_out.conv.bias"])
                # streaming levels: the 1x1 shortcut rides along as extra K segments of the second conv
                # (y = conv2(h) + to_out(x), blocks.py:229-231): one launch, no residual operand
                self.w[f"{n}.conv2s"] = torch.cat([self.w[f"{n}.conv2"].flatten(0, 1), self.w[f"{n}.short"].flatten(0, 1)], 0).contiguous()
                self.v[f"{n}.conv2s.bias"] = (self.v[f"{n}.conv2.bias"] + self.v[f"{n}.short.bias"]).contiguous()
            film_w.append(p[f"{n}.to_scale_shift.to_scale_shift.1.weight"])
            film_b.append(p[f"{n}.to_scale_shift.to_scale_shift.1.bias"])
            self.film_off[n] = off
            off += 2 * r.c_out
        self.film_ld = off
        # fused GroupNorm-FiLM table (persistent deep-level kernel): film2 = A * (film[:, partner] + 1) + Bm * film with
        #   scale positions: A = gamma2, Bm = 0, partner = itself;  shift positions: A = beta2, Bm = 1, partner = its scale
        fa = torch.zeros(off, dtype=torch.float32, device=device)
        fb = torch.zeros(off, dtype=torch.float32, device=device)
        fp = torch.arange(off, dtype=torch.int64, device=device)
        for r in spec.res_blocks():
            o, c = self.film_off[r.name], r.c_out
            fa[o: o + c] = p[f"{r.name}.block2.groupnorm.weight"]
            fa[o + c: o + 2 * c] = p[f"{r.name}.block2.groupnorm.bias"]
            fb[o + c: o + 2 * c] = 1.0
            fp[o + c: o + 2 * c] = torch.arange(o, o + c, dtype=torch.int64, device=device)
        self.film2_a, self.film2_b, self.film2_partner = fa, fb, fp
        # MappingToScaleShift of all blocks as ONE GEMM on the shared mapping (blocks.py:148-165)
        self.w["film"] = pk(torch.cat(film_w, 0)[None])
        self.v["film.bias"] = torch.cat(film_b, 0).contiguous()

        # ---- down / up sampling convs (blocks.py:55-95) -------------------------------------------
        for d in spec.downs:
            self.w[f"{d.name}.down"] = pk(conv_weight_to_gemm(p[f"{d.name}.downsample.conv.weight"]))
            self.v[f"{d.name}.down.bias"] = f32(p[f"{d.name}.downsample.conv.bias"])
        for u in spec.ups:
            wt = p[f"{u.name}.upsample.weight"]
            self.w[f"{u.name}.up"] = pk(conv_weight_to_gemm(wt) if u.factor == 1 else convT_weight_to_gemm(wt, u.factor))
            self.v[f"{u.name}.up.bias"] = f32(p[f"{u.name}.upsample.bias"])

        # ---- Transformer1d (blocks.py:497-537), LayerNorm folded into the projections -------------
        kvx_w, kvx_b, self.kvx_off = [], [], {}
        off = 0
        self.kv_ctx: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
        for t in spec.transformers():
            n = t.name
            assert t.num_layers == 1, "num_transformer_blocks > 1 is not used by JEN-1 configs"
            b = f"{n}.blocks.0"
            self.w[f"{n}.proj"] = pk(conv_weight_to_gemm(p[f"{n}.conv1d.conv.weight"]))
            self.v[f"{n}.proj.bias"] = f32(p[f"{n}.conv1d.conv.bias"])
            self.v[f"{n}.gn.g"] = f32(p[f"{n}.group_norm.weight"])
            self.v[f"{n}.gn.b"] = f32(p[f"{n}.group_norm.bias"])
            a = f"{b}.attention"
            wq, bq = fold_layernorm(p[f"{a}.to_q.weight"], p[f"{a}.norm.weight"], p[f"{a}.norm.bias"])
            wkv, bkv = fold_layernorm(p[f"{a}.to_kv.weight"], p[f"{a}.norm_context.weight"], p[f"{a}.norm_context.bias"])
            self.w[f"{n}.qkv"] = pk(torch.cat([wq, wkv], 0)[None])
            self.v[f"{n}.qkv.bias"] = torch.cat([bq, bkv], 0).contiguous()
            self.v[f"{n}.qkv.u"] = rowsum(torch.cat([wq, wkv], 0))
            self.w[f"{n}.o1"] = pk(p[f"{a}.attention.to_out.weight"][None])
            self.v[f"{n}.o1.bias"] = f32(p[f"{a}.attention.to_out.bias"])
            x = f"{b}.cross_attention"
            wq2, bq2 = fold_layernorm(p[f"{x}.to_q.weight"], p[f"{x}.norm.weight"], p[f"{x}.norm.bias"])
            self.w[f"{n}.q2"] = pk(wq2[None])
            self.v[f"{n}.q2.bias"] = bq2.contiguous()
            self.v[f"{n}.q2.u"] = rowsum(wq2)
            wkv2, bkv2 = fold_layernorm(p[f"{x}.to_kv.weight"], p[f"{x}.norm_context.weight"], p[f"{x}.norm_context.bias"])
            self.v[f"{n}.kv2.bias"] = bkv2.contiguous()
            kvx_w.append(wkv2)
            kvx_b.append(bkv2)
            self.kvx_off[n] = off
            off += wkv2.shape[0]
            self.w[f"{n}.o2"] = pk(p[f"{x}.attention.to_out.weight"][None])
            self.v[f"{n}.o2.bias"] = f32(p[f"{x}.attention.to_out.bias"])
            self.w[f"{n}.ff1"] = pk(p[f"{b}.feed_forward.0.weight"][None])
            self.v[f"{n}.ff1.bias"] = f32(p[f"{b}.feed_forward.0.bias"])
            self.w[f"{n}.ff2"] = pk(p[f"{b}.feed_forward.2.weight"][None])
            self.v[f"{n}.ff2.bias"] = f32(p[f"{b}.feed_forward.2.bias"])
            # streaming levels: a projection and the LayerNorm-folded projection that consumes its output as ONE
            # dual-range GEMM; the LayerNorm finish rstd (raw - mean u) + b moves into the attention kernel
            #   [x1 | qkv_raw] = [P ; Wqkv' P] xh + [p_b ; Wqkv' p_b]                        (blocks.py:530-531, :427-429)
            #   [x2 | q2_raw]  = [W_o | 0 ; Wq2' W_o | Wq2'] [a1 | x1] + [b_o ; Wq2' b_o]     (+ x1 on the x2 rows)
            P64, pb64 = p[f"{n}.conv1d.conv.weight"][:, :, 0].double(), p[f"{n}.conv1d.conv.bias"].double()
            wqkv64, wq264 = torch.cat([wq, wkv], 0).double(), wq2.double()
            self.w[f"{n}.pqkv"] = pk(torch.cat([P64, wqkv64 @ P64], 0).float()[None]).flatten(0, 1).contiguous()
            self.v[f"{n}.pqkv.bias"] = torch.cat([pb64, wqkv64 @ pb64]).float().contiguous()
            Cc_, mid_ = P64.shape[0], wq264.shape[0]
            zc = lambda m_: torch.zeros(m_, dtype=torch.float32, device=device)
            self.v[f"{n}.pqkv.u"] = torch.cat([zc(Cc_), self.v[f"{n}.qkv.u"]]).contiguous()
            self.v[f"{n}.pqkv.b"] = torch.cat([zc(Cc_), self.v[f"{n}.qkv.bias"]]).contiguous()
            wo164, bo164 = p[f"{a}.attention.to_out.weight"].double(), p[f"{a}.attention.to_out.bias"].double()
            top1 = torch.cat([wo164, torch.zeros(Cc_, Cc_, dtype=torch.float64, device=device)], 1)
            bot1 = torch.cat([wq264 @ wo164, wq264], 1)
            self.w[f"{n}.o1q2"] = pk(torch.cat([top1, bot1], 0).float()[None]).flatten(0, 1).contiguous()
            self.v[f"{n}.o1q2.bias"] = torch.cat([bo164, wq264 @ bo164]).float().contiguous()
            self.v[f"{n}.o1q2.u"] = torch.cat([zc(Cc_), self.v[f"{n}.q2.u"]]).contiguous()
            self.v[f"{n}.o1q2.b"] = torch.cat([zc(Cc_), self.v[f"{n}.q2.bias"]]).contiguous()
            # ONE position (the levels with T' = 1): softmax over a single key is 1, so self-attention is to_out(v) with
            # v = to_v(norm_context(x1)) (blocks.py:355-380, :427-437), and a LayerNorm over the channels of a single position is a
            # one-group GroupNorm -- the staging prologue of the GEMM that follows.  The whole sub-block is then
            #   [x2 | q2_raw] = [W_o W_v | 0 ; Wq2' W_o W_v | Wq2'] [LN_ctx(x1) | x1] + [b_o ; Wq2' b_o]     (+ x1 on the x2 rows)
            # (same biases and finish vectors as o1q2; gamma / beta of norm_context are applied by the prologue): no q / k projection, no
            # attention phase.
            wv64 = p[f"{a}.to_kv.weight"].double()[wq.shape[0]:]
            wc64 = wo164 @ wv64
            tops = torch.cat([wc64, torch.zeros(Cc_, Cc_, dtype=torch.float64, device=device)], 1)
            bots = torch.cat([wq264 @ wc64, wq264], 1)
            self.w[f"{n}.s1q2"] = pk(torch.cat([tops, bots], 0).float()[None]).flatten(0, 1).contiguous()
            self.v[f"{n}.nc.g"] = f32(p[f"{a}.norm_context.weight"])
            self.v[f"{n}.nc.b"] = f32(p[f"{a}.norm_context.bias"])
            # streaming levels: cross-attention output projection and the first FeedForward layer as ONE
            # dual-range GEMM over the K concat [a | x2]:  x3 = x2 + W_o a + b_o  (rows < C, K = a only) and
            # f = gelu(W_1 x3 + b_1) = gelu((W_1 W_o) a + W_1 x2 + W_1 b_o + b_1)   (blocks.py:485-488, :440-446)
            wo64, w164 = p[f"{x}.attention.to_out.weight"].double(), p[f"{b}.feed_forward.0.weight"].double()
            top = torch.cat([wo64, torch.zeros(wo64.shape[0], w164.shape[1], dtype=torch.float64, device=device)], 1)
            bot = torch.cat([w164 @ wo64, w164], 1)
            self.w[f"{n}.o2f1"] = pk(torch.cat([top, bot], 0).float()[None]).flatten(0, 1).contiguous()
            self.v[f"{n}.o2f1.bias"] = torch.cat([p[f"{x}.attention.to_out.bias"].double(),
                                                  p[f"{b}.feed_forward.0.bias"].double() + w164 @ p[f"{x}.attention.to_out.bias"].double()]).float().contiguous()
            # streaming levels: the output 1x1 conv applied to x + ff2(f) is ONE GEMM over the K concat
            # [x | f] with weights [P | P @ W_ff2] (blocks.py:446, :488, :536): no x4 round trip
            wp64 = p[f"{n}.conv1d.conv.weight"][:, :, 0].double()
            wm = torch.cat([wp64, wp64 @ p[f"{b}.feed_forward.2.weight"].double()], 1).float()
            self.w[f"{n}.ffp"] = pk(wm[None]).flatten(0, 1).contiguous()
            self.v[f"{n}.ffp.bias"] = (p[f"{n}.conv1d.conv.bias"].double() + wp64 @ p[f"{b}.feed_forward.2.bias"].double()).float().contiguous()
        self.kvx_ld = off
        if kvx_w:
            # time-token K/V rows of every cross-attention layer as ONE GEMM per step
            # ... and, row-major [sum 2C][F], the B operand of jen1_big_gemm: the text tokens' K/V of ALL layers in one launch per
            # conditioning (Plan.set_context) and the fixed embedding's K/V once per weight load; ``kvx_off`` is each layer's n0
            self.w["kv2_all"] = torch.cat(kvx_w, 0).to(dtype).contiguous()
            self.w["kvx"] = pk(torch.cat(kvx_w, 0)[None])
            self.v["kvx.bias"] = torch.cat(kvx_b, 0).contiguous()
            self.v["kvx.u"] = rowsum(torch.cat(kvx_w, 0))
        self.fixed = p["fixed_embedding.embedding.weight"].contiguous()       # [ctx_len][F] float32

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.w.values())

    def fp8(self, w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """JEN1_FP8 form of a packed weight ([..][M/16][64 lanes][8], any leading block dims): OCP e4m3 bytes in the same
        fragment order + one float32 scale per output row m = 16 mt + (lane & 15), chosen so that the row's largest magnitude
        maps to 448 (the e4m3 maximum).  Cached per packed tensor; built when a plan first puts the layer into the persistent
        launch (the launch-per-layer levels keep the bf16 copy)."""
        hit = self._fp8.get(id(w))
        if hit is not None and hit[0] is w:
            return hit[1], hit[2]
        MT = w.shape[-3]
        wf = w.reshape(-1, MT, 4, 16, 8).to(torch.float32)                   # [chunk, mt, g, i, j]
        amax = wf.abs().amax(dim=(0, 2, 4))                                  # [mt, i]
        scale = (amax / 448.0).clamp_min(1e-20)
        q = (wf / scale[None, :, None, :, None]).to(torch.float8_e4m3fn).view(torch.uint8).reshape(-1, MT, 64, 8).contiguous()
        scale = scale.reshape(-1).contiguous()
        self._fp8[id(w)] = (w, q, scale)
        return q, scale


def _bgemm_parts(groups):
    """column groups (c, bias, n0, N, ldc) of a stacked jen1_big_gemm operand -> launches: all groups in one launch when every
    n0 and N is a multiple of the kernel's 128-column tile, else one launch per group (a single group may have any width)"""
    if all(n0 % 128 == 0 and N % 128 == 0 for _, _, n0, N, _ in groups):
        return [groups]
    return [[gr] for gr in groups]


def _group_align(part) -> int:
    """``jen1_bgemm_args.group_align`` of one launch's groups (c, bias, n0, N, ldc): 256 when every boundary RELATIVE to the launch's
    first column is a 256-column multiple (the kernel may then use its 256 x 256 tile form), else 128 (the minimum)"""
    n_lo = part[0][2]
    return 256 if all((n0 - n_lo) % 256 == 0 and N % 256 == 0 for _, _, n0, N, _ in part) else 128


class DeepIneligible(Exception):
    """a layer does not fit the persistent deep-level kernel (LDS / staging registers / unsupported option)"""


class DeepProgram:
    """Host side of the persistent deep-level launch (include/jen1_deep.h): collects one phase descriptor per layer of the
    levels with few positions, links them (dependency chain, unit -> workgroup rotation) and copies the array to the device.
    The descriptors are opaque bytes here; libjen1_hip.so fills and validates them."""

    _static_owner: Dict[str, "weakref.ref"] = {}       # per device: the program that may use the static schedule

    def __init__(self, eng, leader: Optional["DeepProgram"] = None, err: Optional[torch.Tensor] = None):
        """leader: the first program of the same plan (a plan may run several persistent launches per step, one behind the other on one
        stream: they share the scheduling form and the error word)"""
        self.eng = eng
        self.leader = leader if leader is not None else self
        self._err_shared = err
        self.w_bytes = self.act_bytes = self.flops = 0      # algorithmic traffic / work of the recorded phases
        self.kinds: List[str] = []
        self.lib = eng.lib
        self.psize = self.lib.jen1_deep_phase_size()
        self.bufs: List[C.Array] = []
        self.labels: List[str] = []
        self.outs: List[Optional["Act"]] = []
        self.nwg = self.lib.jen1_deep_num_workgroups()
        self.dev = None
        self.lds = 0
        self.sync = None
        self._produced: Dict[int, torch.Tensor] = {}     # storage pointer -> tensor, of everything a recorded phase writes
        # scheduling form of the launch (jen1_deep_run_mode): the static unit -> workgroup map is ~9 % faster but needs every
        # workgroup resident, so only ONE program per device and process may use it at a time -- the first one alive claims it, every
        # other persistent launch that could share the GPU with it (concurrent samplers, other shapes on other streams) goes by ticket
        # and cannot deadlock.  JEN1_DEEP_SHARED=1 (the GPU is shared with other PROCESSES that run this library) forces tickets.
        self._exclusive = False

    @property
    def exclusive(self) -> bool:
        return self.leader._exclusive

    @exclusive.setter
    def exclusive(self, v: bool):
        self.leader._exclusive = bool(v)

    STATIC_IDLE_S = 1.0          # an owner that has not launched for this long gives the static schedule to whoever asks

    # host-side launches / sentinel resets of ANY persistent program since the process started: a fused sampler, whose replayed step
    # relies on the sentinels its own previous step left (DDIMStepper, jen1_step_tail), re-packs and re-poisons when somebody else ran
    host_serial = [0]

    def touch(self) -> None:
        self.leader._last_use = time.monotonic()

    def claim_static(self) -> bool:
        """Ask for the device's static unit -> workgroup schedule (~9 % faster than tickets, correct only while every workgroup of
        the launch is resident, hence ONE program per device).  Granted when nobody holds it, or when the holder has been idle for
        STATIC_IDLE_S (a warm-up shape built first must not keep the main shape on tickets for the life of the process); a holder
        that is in use keeps it, so samplers running side by side never take it from each other.  Taking it over synchronises the
        device first (launches of the old holder that are still queued were recorded as static) and flips the old holder to
        tickets; a stepper whose captured graph recorded the other form re-captures (DDIMStepper.reset).  JEN1_DEEP_SHARED=1
        (other PROCESSES run persistent launches on this GPU) forces tickets everywhere.  Returns whether this program holds it."""
        me = self.leader
        self.touch()                     # asking = using: a program that is being reset for a trajectory is not idle
        if me._exclusive:
            return True
        if os.environ.get("JEN1_DEEP_SHARED", "0") == "1":
            return False
        key = str(self.eng.device)
        ref = DeepProgram._static_owner.get(key)
        owner = None if ref is None else ref()
        owner = None if owner is None else owner.leader        # (whoever registered a program of a plan meant the plan's leader)
        if owner is not None and owner is not me and owner._exclusive:
            if time.monotonic() - getattr(owner, "_last_use", 0.0) < self.STATIC_IDLE_S:
                return False
            torch.cuda.synchronize(self.eng.device)
            owner._exclusive = False
        DeepProgram._static_owner[key] = weakref.ref(me)
        me._exclusive = True
        return True

    def _note_output(self, out: Optional["Act"]):
        if out is not None:
            self._produced.setdefault(out.t.untyped_storage().data_ptr(), out.t)

    def is_live(self, t: Optional[torch.Tensor]) -> bool:
        """was ``t`` (or the tensor it is a view of) written by an earlier phase of this program?  Such operands start every
        launch poisoned and are polled by their consumers (include/jen1_deep.h)"""
        return t is not None and t.untyped_storage().data_ptr() in self._produced

    def _new(self):
        return (C.c_char * self.psize)()

    def _add(self, buf, rc, label, out, kind="gemm"):
        if rc != 0:
            msg = self.lib.jen1_last_error()
            raise DeepIneligible(f"{label}: {msg.decode() if msg else 'does not fit'}")
        self.bufs.append(buf)
        self.labels.append(label)
        self.outs.append(out)
        self.kinds.append(kind)
        self._note_output(out)

    def add_conv(self, a: "L.ConvArgs", label: str, out, nb_max: int = 0):
        buf = self._new()
        self._add(buf, self.lib.jen1_deep_phase_conv(C.byref(a), nb_max, C.cast(buf, C.c_void_p)), label, out)

    def add_tile(self, a: "L.ConvArgs", tb: int, bm: int, st, out_part, out_nfg: int, label: str, out):
        """a layer of a long level (jen1_deep_phase_tile).  st: per normalised source (pointer, tiles, groups, live) or None"""
        buf = self._new()
        st = list(st) + [None] * (2 - len(st))
        p = [(0, 0, 0) if e is None else (e[0], e[1], e[2]) for e in st]
        live = sum(1 << k for k, e in enumerate(st) if e is not None and e[3])
        rc = self.lib.jen1_deep_phase_tile(C.byref(a), tb, bm, p[0][0] or None, p[0][1], p[0][2], p[1][0] or None, p[1][1], p[1][2], live,
                                           None if out_part is None else out_part.data_ptr(), out_nfg, C.cast(buf, C.c_void_p))
        self._add(buf, rc, label, out, kind="tile")
        if out_part is not None:
            self._produced.setdefault(out_part.untyped_storage().data_ptr(), out_part)

    def add_attention(self, args, label: str, out):
        buf = self._new()
        self._add(buf, self.lib.jen1_deep_phase_attention(*args, C.cast(buf, C.c_void_p)), label, out, kind="attn")

    def add_stats(self, x: "Act", stats: torch.Tensor, label: str):
        buf = self._new()
        self._add(buf, self.lib.jen1_deep_phase_stats(x.t.data_ptr(), stats.data_ptr(), x.B, x.L, x.ld, self.eng.deep_dt,
                                                      C.cast(buf, C.c_void_p)), label, None, kind="stats")

    def __len__(self):
        return len(self.bufs)

    def sync_words(self) -> int:
        return int(self.lib.jen1_deep_sync_bytes(len(self.bufs))) // 4

    def finalize(self, sync: torch.Tensor):
        """link the phases, build the device image (per-phase blobs + headers) and move it to the device; ``sync``: int32 view
        of a zeroed-per-step area of sync_words()"""
        n = len(self.bufs)
        host = (C.c_char * (self.psize * n))()
        for i, b in enumerate(self.bufs):
            C.memmove(C.addressof(host) + i * self.psize, b, self.psize)
        bb = self.lib.jen1_deep_blob_bytes()
        blobs = (C.c_char * (bb * n))()
        hdrs = (C.c_char * (16 * n))()
        self.lds = self.lib.jen1_deep_link(C.cast(host, C.c_void_p), n, self.nwg, C.cast(blobs, C.c_void_p), C.cast(hdrs, C.c_void_p))
        if self.lds <= 0 or self.nwg < 1:
            msg = self.lib.jen1_last_error()
            raise DeepIneligible(f"jen1_deep_link: {msg.decode() if msg else 'failed'}")
        self.dev = torch.frombuffer(bytearray(bytes(blobs)), dtype=torch.uint8).to(self.eng.device)
        self.hdr = torch.frombuffer(bytearray(bytes(hdrs)), dtype=torch.uint8).to(self.eng.device)
        self.sync = sync
        # the error word lives outside the per-step zeroed area: the first time-out of any replay stays visible (error())
        self.err = self._err_shared if self._err_shared is not None else torch.zeros((16,), dtype=torch.int32, device=self.eng.device)
        # every tensor a phase writes starts each launch as the sentinel (jen1_deep_poison): {pointer, bytes} per storage
        ent = []
        for ptr, t in self._produced.items():
            nb = t.untyped_storage().nbytes()
            assert ptr % 16 == 0 and nb % 16 == 0, (ptr, nb)
            ent += [ptr, nb]
        if self.leader is self:
            self.claim_static()
        self.poison_tab = torch.tensor(ent, dtype=torch.int64).view(-1, 2).to(self.eng.device)
        self.poison_bytes = int(sum(ent[1::2]))

    def poison(self, stream: int, zero: Optional[Tuple[int, int]] = None):
        """before every launch, after the last reader of the previous one: the step's first node (Plan) does this; ``zero`` =
        (pointer, bytes) of the plan's per-step statistics arena, reset by the same launch"""
        DeepProgram.host_serial[0] += 1
        if zero is not None:
            L.check(self.lib.jen1_deep_poison_zero(self.poison_tab.data_ptr(), self.poison_tab.shape[0], self.sync.data_ptr(), zero[0], zero[1],
                                                   stream), "jen1_deep_poison_zero")
            return
        L.check(self.lib.jen1_deep_poison(self.poison_tab.data_ptr(), self.poison_tab.shape[0], self.sync.data_ptr(), stream), "jen1_deep_poison")

    def launch(self, stream: int):
        self.touch()
        DeepProgram.host_serial[0] += 1
        n = len(self.bufs)
        if os.environ.get("JEN1_DEEP_RUN_PHASES"):          # debugging: run only the first phases of the program
            n = min(n, int(os.environ["JEN1_DEEP_RUN_PHASES"]))
        km = sum(1 << k for k, name in enumerate(("gemm", "attn", "stats", "tile")) if name in self.kinds)
        L.check(self.lib.jen1_deep_run_kinds(self.dev.data_ptr(), self.hdr.data_ptr(), n, self.sync.data_ptr(), self.err.data_ptr(), self.nwg,
                                             self.lds, self.eng.deep_dt, 0 if self.exclusive else 1, km, stream), "jen1_deep_run_kinds")

    def error(self) -> int:
        """non-zero after a launch whose dependency wait timed out (1 + phase index); synchronises with the device"""
        return int(self.err[0].item())

    def take_error(self) -> int:
        """``error()``, and the word is cleared when it was set: one transient time-out must not leave the cached plan 'dead'
        (every later launch would fall through its waits on the stale word)"""
        e = self.error()
        if e:
            self.err.zero_()
        return e


class LongIneligible(DeepIneligible):
    """a layer of a long level does not fit the sample-resident launch (include/jen1_long.h): the plan keeps one launch per layer there"""


class LongProgram(DeepProgram):
    """Host side of one sample-resident long-level launch (include/jen1_long.h): to_in + the down path above the deep levels, or the
    up path above them + to_out.  One 512-byte descriptor per convolution; sample b runs on workgroups b, b + B, ... (G = nwg // B per
    sample).  Shares the plan's leader (static schedule claim, error word) with the deep program(s)."""

    POISON_CHUNK = 1 << 18          # a poisoned tensor is cut into table rows of this many bytes (8 workgroups per row)

    def __init__(self, eng, Bs: int, leader: Optional["DeepProgram"] = None, err: Optional[torch.Tensor] = None):
        super().__init__(eng, leader=leader, err=err)
        self.Bs = Bs
        self.G = self.nwg // Bs
        self.dsize = 512
        if self.G < 1:
            raise LongIneligible(f"{Bs} samples on {self.nwg} workgroups")

    def geometry(self, M: int, L_out: int):
        mb, tl, tb = C.c_int(0), C.c_int(0), C.c_int(0)
        if self.lib.jen1_long_geometry(M, L_out, self.G, C.byref(mb), C.byref(tl), C.byref(tb)) != 0:
            msg = self.lib.jen1_last_error()
            raise LongIneligible(msg.decode() if msg else "geometry")
        return mb.value, tl.value, tb.value

    def add_long(self, a: "L.ConvArgs", st, st_live: int, out_part, label: str, out):
        """st: per normalised source None or (pointer, entries, mblocks, nsub) -- entries = 0: fine-group totals"""
        buf = (C.c_char * self.dsize)()
        st = list(st) + [None] * (2 - len(st))
        q = [(None, 0, 1, 8) if e is None else e for e in st]
        rc = self.lib.jen1_long_phase_conv(C.byref(a), self.G, q[0][0], q[0][1], q[0][2], q[0][3], q[1][0], q[1][1], q[1][2], q[1][3], st_live,
                                           None if out_part is None else out_part.data_ptr(), C.cast(buf, C.c_void_p))
        if rc != 0:
            msg = self.lib.jen1_last_error()
            raise LongIneligible(f"{label}: {msg.decode() if msg else 'does not fit'}")
        self.bufs.append(buf)
        self.labels.append(label)
        self.outs.append(out)
        self.kinds.append("long")
        self._note_output(out)
        if out_part is not None:
            self._produced.setdefault(out_part.untyped_storage().data_ptr(), out_part)

    def sync_words(self) -> int:
        return 64                   # word 0: the ticket counter of the ticket form (zero when the launch starts)

    def finalize(self, sync: torch.Tensor):
        n = len(self.bufs)
        host = (C.c_char * (self.dsize * n))()
        lds = 0
        for i, b in enumerate(self.bufs):
            C.memmove(C.addressof(host) + i * self.dsize, b, self.dsize)
            lds = max(lds, int(self.lib.jen1_long_phase_lds(C.cast(b, C.c_void_p))))
        self.lds = lds
        self.dev = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(self.eng.device)
        self.sync = sync
        self.err = self._err_shared if self._err_shared is not None else torch.zeros((16,), dtype=torch.int32, device=self.eng.device)
        ent = []
        for ptr, t in self._produced.items():
            nb = t.untyped_storage().nbytes()
            assert ptr % 16 == 0 and nb % 16 == 0, (ptr, nb)
            for o in range(0, nb, self.POISON_CHUNK):
                ent += [ptr + o, min(self.POISON_CHUNK, nb - o)]
        if self.leader is self:
            self.claim_static()
        self.poison_tab = torch.tensor(ent, dtype=torch.int64).view(-1, 2).to(self.eng.device)
        self.poison_bytes = int(sum(ent[1::2]))
        self.local = self.xcd_local()

    def xcd_local(self) -> bool:
        """may this program keep its outputs in the L2 of the XCD that wrote them (jen1_long_run's ``local`` form)?  The groups are then
        formed inside the kernel from the XCD every workgroup actually runs on; needs a multiple of 8 samples and one workgroup per CU
        on 8 XCDs.  OFF by default (JEN1_LONG_LOCAL=1 switches it on): measured at B = 8, T = 1500 the step is not faster with it (836.3 against
        835.5 steps/s) -- a phase is bound by the consumer's own instruction stream, not by the hand-off."""
        return os.environ.get("JEN1_LONG_LOCAL", "0") == "1" and self.Bs % 8 == 0 and self.nwg % 8 == 0 and self.nwg % self.Bs == 0

    def poison(self, stream: int, zero=None):
        # (stand-alone use -- tests, tools: the synchronisation words are part of the plan's per-step arena reset otherwise)
        L.check(self.lib.jen1_memset_zero(self.sync.data_ptr(), self.sync.numel() * 4, stream), "memset")
        super().poison(stream, zero)

    def launch(self, stream: int):
        self.touch()
        DeepProgram.host_serial[0] += 1
        n = len(self.bufs)
        if os.environ.get("JEN1_LONG_RUN_PHASES"):          # debugging: run only the first phases of the program
            n = min(n, int(os.environ["JEN1_LONG_RUN_PHASES"]))
        static = self.exclusive
        loc = static and self.local
        L.check(self.lib.jen1_long_run(self.dev.data_ptr(), n, self.Bs, self.err.data_ptr(), None if (static and not loc) else self.sync.data_ptr(),
                                       self.nwg, self.lds, self.eng.dt, 1 if loc else 0, stream), "jen1_long_run")


class KernelCtx:
    """what an OpBuilder needs to know about the device / dtype (Engine provides the same fields)."""

    def __init__(self, dtype: str = "f32", device="cuda", target_wgs: int = 256):
        self.lib = L.load()
        self.device = torch.device(device)
        self.dt = L.F32 if dtype == "f32" else L.BF16
        self.tdtype = torch.float32 if dtype == "f32" else torch.bfloat16
        self.deep_dt = self.dt
        self.target_wgs = target_wgs
        self.splitk_target_wgs = 512
        self.splitk_min_bytes = 1 << 20
        self.fuse_shortcut = True
        self.fuse_shortcut_tiles = True
        self.fuse_ff_out = True
        self.fuse_o2_ff1 = True
        self.use_tile_kernel = True
        self.fuse_ln_proj = True
        self.fold_single_position = True
        self.tile_min_rows = 512
        self.tile_target_wgs = 256
        self.tile_one_round = False
        self.stream_bn = 64
        self.stream_max_wgs = 100000


class OpBuilder:
    """Prepares jen1_* launches with fixed pointers; ``Plan`` builds the whole network with it,
    the unit tests build single ops."""

    def __init__(self, eng):
        self.eng = eng
        self.ops: List[Callable[[int], None]] = []
        self._splitk_args: List[L.ConvArgs] = []
        self._slab_floats = 0
        self._max_tiles = 0
        self._keep: List[object] = []
        self.slab = None
        self.counters = None
        self.det = getattr(self, "det", False)      # fixed-order statistics launches instead of epilogue atomics (Plan)
        self.deep: Optional[DeepProgram] = None     # the persistent program being recorded (Plan._prog_open); after the build: the one with the deep levels
        self._deep_on = False                       # layers are GEMM / attention phases (the levels with few positions)
        self._prog: Optional[DeepProgram] = None    # the open program, if any
        self.tile_lens = getattr(self, "tile_lens", frozenset())      # input lengths whose layers are tile phases (Plan)
        self.tile_errors: List[str] = []
        self._long_on = False                       # layers are phases of a sample-resident long-level launch (Plan._long_begin)
        self._lprog: Optional[LongProgram] = None   # the open one

    def _empty(self, shape, dtype=None):
        return torch.empty(shape, dtype=dtype or self.eng.tdtype, device=self.eng.device)

    def finalize_workspace(self):
        """split-K workspace shared by all launches of the builder (stream-ordered reuse)."""
        if self._splitk_args:
            dev = self.eng.device
            self.slab = torch.empty((self._slab_floats,), dtype=torch.float32, device=dev)
            self.counters = torch.zeros((self._max_tiles,), dtype=torch.int32, device=dev)
            for a in self._splitk_args:
                a.slab, a.counters = self.slab.data_ptr(), self.counters.data_ptr()

    def run(self, stream: Optional[int] = None, pack: bool = True, poison: bool = True):
        """pack=False leaves out the ops that write the network input (kind "pack"), poison=False the head-of-step sentinel / arena
        reset node (kind "deep_poison"): the fused sampler's previous step did both (jen1_cfg_ddim_step_pack / jen1_step_tail)"""
        if stream is None:
            stream = torch.cuda.current_stream(self.eng.device).cuda_stream
        skip = (() if pack else ("pack",)) + (() if poison else ("deep_poison",))
        ops = self.ops if not skip else [op for op in self.ops if getattr(op, "kind", "") not in skip]
        if os.environ.get("JEN1_DEBUG_SYNC"):
            for i, op in enumerate(ops):
                print(f"[jen1] op {i}: {getattr(op, 'label', '?')}", flush=True)
                op(stream)
                torch.cuda.synchronize(self.eng.device)
            return
        for op in ops:
            op(stream)

    # ---------------------------------------------------------------- the fused conv / linear op
    def conv(self, ops, *, src0: Act, w: torch.Tensor, bias, out: Act, taps=1, stride=1, pad_left=0, L_out=None,
             src1: Optional[Act] = None, src1_scale=1.0, ps_f=1, ps_off=0, L_y=None, y_row0=0, pro=L.PRO_NONE,
             gn=None, film=None, ln=None, act=L.ACT_NONE, residual: Optional[Act] = None, row_scale=None,
             y_f32=False, out_C=None, force=None, label="", extra_segs=None, m_split=0, k_split=0, flat_w=False):
        """extra_segs: [(Act, row_shift)] raw sources appended to the K axis after the (tap, source) pairs of
        src0/src1 (streaming / direct mode only); ``w`` is then the flat packed weight [chunks][M/16][64][8]."""
        eng = self.eng
        a = L.ConvArgs()
        a.x0, a.c0, a.ld0 = src0.t.data_ptr(), src0.cp, src0.ld
        if src1 is not None:
            a.x1, a.c1, a.ld1 = src1.t.data_ptr(), src1.cp, src1.ld
            assert src1.L == src0.L and src1.B == src0.B
        a.w, a.bias = w.data_ptr(), _ptr(bias)
        a.dtype = eng.dt
        a.B, a.L_in = src0.B, src0.L
        a.L_out = L_out if L_out is not None else src0.L
        a.taps, a.stride, a.pad_left = taps, stride, pad_left
        out_C = out_C if out_C is not None else out.C
        a.out_C, a.ps_f, a.ps_off = out_C, ps_f, ps_off
        a.M = out_C * ps_f
        if extra_segs or flat_w:
            k_extra = sum(e.cp for e, _ in extra_segs) if extra_segs else 0
            assert w.dim() == 4 and w.shape[1] * 16 == a.M and w.shape[0] * 32 == taps * (a.c0 + a.c1) + k_extra, \
                (tuple(w.shape), taps, a.M, a.c0, a.c1, k_extra)
        else:
            assert w.shape[0] == taps and w.shape[2] * 16 == a.M and w.shape[1] * 32 == a.c0 + a.c1, \
                (tuple(w.shape), taps, a.M, a.c0, a.c1)
        a.y, a.ld_y = out.t.data_ptr(), out.ld
        a.L_y = L_y if L_y is not None else (out.L - y_row0)
        a.y_brows, a.y_row0 = out.L, y_row0
        a.y_f32 = 1 if y_f32 else 0
        if residual is not None:
            a.residual, a.ld_res = residual.t.data_ptr(), residual.ld
            assert residual.L == out.L and residual.B == out.B and y_row0 == 0
        a.pro_mode = pro
        a.src1_scale = float(src1_scale)
        if pro in (L.PRO_GN, L.PRO_GN_SILU):
            groups, creal, gamma, beta, eps = gn
            a.gn_groups = groups
            a.gn_cpg = creal // groups if groups > 1 else max(creal, a.c0 + a.c1)
            a.gn_count = (creal // groups) * src0.L
            a.gn_eps = eps
            a.gn_gamma, a.gn_beta = gamma.data_ptr(), beta.data_ptr()
            assert gamma.numel() == a.c0 + a.c1, (gamma.numel(), a.c0, a.c1)
            a.gn_stats0 = src0.gn.data_ptr()
            if src1 is not None:
                a.gn_stats1 = src1.gn.data_ptr()
            if film is not None:
                ftab, frow, foff, fC = film[:4]
                a.film, a.film_row = ftab.data_ptr(), _ptr(frow)
                a.film_off, a.film_C, a.film_ld = foff, fC, ftab.shape[-1]
                a.film_step = _ptr(film[4]) if len(film) > 4 else None
        ln_u = None
        if pro == L.PRO_LN:
            lnC, g_, b_ = ln[:3]
            ln_u = ln[3] if len(ln) > 3 else None
            a.ln_rowstats, a.ln_C, a.ln_eps = src0.rs.data_ptr(), lnC, 1e-5
            a.ln_gamma, a.ln_beta = _ptr(g_), _ptr(b_)
        a.act = act
        a.row_scale = _ptr(row_scale)
        if out.gn is not None and not y_f32:
            a.out_gn_stats, a.out_cpf = out.gn.data_ptr(), out.ld // FG
        if out.rs is not None:
            a.out_rowstats = out.rs.data_ptr()
        if self._long_on:
            return self._conv_long(a, src0=src0, src1=src1, w=w, bias=bias, out=out, pro=pro, gn=gn, film=film, act=act, residual=residual,
                                   row_scale=row_scale, y_f32=y_f32, extra_segs=extra_segs, m_split=m_split, label=label, taps=taps, out_C=out_C)
        if self._deep_on:
            # persistent deep-level kernel: one phase descriptor instead of a launch (norm_apply + streaming GEMM); the consumer
            # computes the GroupNorm statistics itself, FiLM comes from the fused GroupNorm-FiLM table
            if pro == L.PRO_LN or y_f32 or row_scale is not None:
                raise DeepIneligible(f"{label}: prologue / epilogue option outside the persistent kernel")
            if out.ld != out.C:
                raise DeepIneligible(f"{label}: {out.C} output channels are not a multiple of 32 (padding columns would stay poisoned)")
            a.out_gn_stats = a.out_rowstats = None
            w8 = None
            if eng.deep_dt == L.FP8:
                # JEN1_FP8: e4m3 weights + per-row scales feed the fp8 matrix-core path of the unit; activations stay bf16 in HBM
                w8 = eng.W.fp8(w)
                a.dtype, a.w, a.w_scale = L.FP8, w8[0].data_ptr(), w8[1].data_ptr()
            if pro in (L.PRO_GN, L.PRO_GN_SILU) and film is not None:
                a.film = self.film2.data_ptr()
            a.nseg = 0
            if extra_segs:
                for i, (e, sh) in enumerate(extra_segs):
                    assert e.B == a.B and e.L == a.L_in and e.t.dtype == eng.tdtype and e.cp == e.C
                    a.seg[i].x, a.seg[i].ld, a.seg[i].shift, a.seg[i].kch = e.t.data_ptr(), e.ld, sh, e.cp // 32
                a.nseg = len(extra_segs)
            # operands produced by earlier phases of the launch are polled (they start poisoned); source order as in
            # jen1_deep_phase_conv: x0, x1 (when present), then the extra segments
            srcs_ = [src0] + ([src1] if src1 is not None else []) + [e for e, _ in (extra_segs or [])]
            a.live_mask = sum(1 << i for i, s_ in enumerate(srcs_) if self.deep.is_live(s_.t)) | \
                (256 if residual is not None and self.deep.is_live(residual.t) else 0)
            if m_split:
                a.m_split, a.k_split = m_split, k_split
            self._keep.append((a, src0, src1, w, w8, bias, out, residual, gn, film, extra_segs))
            # batch elements per unit: fewer per unit = smaller tiles to normalise and stage (the critical path of a phase) on more
            # workgroups, but every batch group streams the layer's weights again -- so only where the weights are small
            nb_cap = eng.deep_nb_max
            if eng.deep_unit_target > 0:
                wbytes_ = (taps * (a.c0 + a.c1) + sum(e.cp for e, _ in (extra_segs or []))) * a.M * (1 if w8 is not None else (4 if eng.dt == L.F32 else 2))
                nb_ = 1
                while nb_ < a.B and ((a.M // 16) * -(-a.B // nb_) > eng.deep_unit_target or -(-a.B // nb_) * wbytes_ > eng.deep_reread_cap):
                    nb_ *= 2
                nb_cap = nb_ if nb_cap == 0 else min(nb_cap, nb_)
            self.deep.add_conv(a, f"conv[{label}] B={a.B} Lin={a.L_in} Lout={a.L_out} c0={a.c0} c1={a.c1} taps={a.taps} M={a.M}", out,
                               nb_max=nb_cap)
            es_ = 4 if eng.dt == L.F32 else 2
            c_real_ = src0.C + (src1.C if src1 is not None else 0)
            c_extra_ = sum(e.C for e, _ in extra_segs) if extra_segs else 0
            self.deep.w_bytes += (taps * c_real_ + c_extra_) * a.M * (1 if w8 is not None else es_) + (4 * a.M if w8 is not None else 0)
            self.deep.act_bytes += a.B * a.L_in * (c_real_ + c_extra_) * es_ + a.B * a.L_y * out_C * es_
            self.deep.flops += 2 * (taps * c_real_ + c_extra_) * a.M * a.B * a.L_out
            out.gn_valid = False
            return out
        if src0.L in self.tile_lens and not self.det:
            # a long level inside the persistent launch: a tile phase (jen1_deep_phase_tile); a layer that does not fit runs as a launch
            try:
                return self._conv_tile(L.ConvArgs.from_buffer_copy(a), src0=src0, src1=src1, w=w, bias=bias, out=out, pro=pro, gn=gn, film=film, ln=ln, act=act,
                                       residual=residual, row_scale=row_scale, y_f32=y_f32, extra_segs=extra_segs, m_split=m_split,
                                       flat_w=flat_w, label=label, taps=taps, out_C=out_C)
            except DeepIneligible as e:
                self.tile_errors.append(str(e))
        if self._prog is not None:
            # a launch behind phases of the open program: the program ends here; what this launch normalises gets its totals first
            need = [s_ for s_ in (src0, src1) if s_ is not None and pro in (L.PRO_GN, L.PRO_GN_SILU) and not s_.gn_valid]
            self._prog_close(need)
        if pro in (L.PRO_GN, L.PRO_GN_SILU):
            assert src0.gn_valid and (src1 is None or src1.gn_valid), f"{label}: GroupNorm totals of a tensor produced inside a persistent launch were never written"
        # the tile kernel takes up to two raw extra K segments at row shift 0 (a 1x1 shortcut riding on the block's second conv)
        extras_tile = not extra_segs or (len(extra_segs) <= 2 and all(sh == 0 and e.cp == e.C and e.cp % 32 == 0 for e, sh in extra_segs))
        det_gn = det_rs = False
        if self.det:
            det_gn, det_rs = bool(a.out_gn_stats), bool(a.out_rowstats)
            a.out_gn_stats = a.out_rowstats = None
        tile_ok = (pro in (L.PRO_NONE, L.PRO_GN, L.PRO_GN_SILU, L.PRO_SILU) and act == L.ACT_NONE and row_scale is None and (out.rs is None or det_rs)
                   and extras_tile and not m_split and (pro not in (L.PRO_GN, L.PRO_GN_SILU) or gn[0] > 1 or src1 is None)
                   and (pro not in (L.PRO_GN, L.PRO_GN_SILU) or gn[0] == 1 or
                        (a.gn_cpg % (a.c0 // FG) == 0 and (a.c1 == 0 or a.gn_cpg % (a.c1 // FG) == 0))))
        self._choose_tiles(a, force, k_extra=(sum(e.cp for e, _ in extra_segs) // 32 if extra_segs else 0), tile_ok=tile_ok)
        lib = eng.lib
        streaming = a.cfg in (L.CFG_S16x64, L.CFG_S16x32, L.CFG_S16x16)
        tiled = a.cfg in self.TILE_CFGS
        if extra_segs and tiled and not eng.fuse_shortcut_tiles:
            return None
        if extra_segs and not streaming and not tiled:
            return None           # (a wide W* tile: the caller falls back to separate launches)
        if extra_segs and tiled:
            for i, (e, sh) in enumerate(extra_segs):
                assert e.B == a.B and e.L == a.L_in and e.t.dtype == eng.tdtype
                a.seg[i].x, a.seg[i].ld, a.seg[i].shift, a.seg[i].kch = e.t.data_ptr(), e.ld, 0, e.cp // 32
            a.nseg = len(extra_segs)
        want_direct = streaming and (force is None or force.get("direct", True))
        if want_direct and pro == L.PRO_LN and ln_u is not None and a.ln_gamma is None and taps == 1 and stride == 1 \
                and src1 is None and (force is None or force.get("ln_fold", True)):
            # LayerNorm folded into the epilogue: no pre-pass, the GEMM streams the raw rows
            a.ln_fold, a.ln_u = 1, ln_u.data_ptr()
            a.pro_mode = L.PRO_NONE
            a.direct = 1
            self._keep.append(ln_u)
        elif want_direct and pro in (L.PRO_GN, L.PRO_GN_SILU, L.PRO_LN):
            # deep level: up to M/16 workgroups would each redo the prologue of the same tiny tile --
            # normalise / activate it ONCE (jen1_norm_apply), then stream the GEMM with no LDS staging
            na = L.NormArgs()
            ctot = a.c0 + a.c1
            xh = self._empty((a.B, a.L_in, ctot))
            na.x0, na.x1, na.y = a.x0, a.x1, xh.data_ptr()
            na.dtype, na.mode, na.B, na.L = eng.dt, pro, a.B, a.L_in
            na.c0, na.c1, na.ld0, na.ld1, na.ld_y = a.c0, a.c1, a.ld0, a.ld1, ctot
            na.src1_scale = a.src1_scale
            if pro == L.PRO_LN:
                na.ln_rowstats, na.count, na.eps = a.ln_rowstats, a.ln_C, a.ln_eps
                na.gamma, na.beta = a.ln_gamma, a.ln_beta
            else:
                na.gn_stats0, na.gn_stats1 = a.gn_stats0, a.gn_stats1
                na.gamma, na.beta = a.gn_gamma, a.gn_beta
                na.groups, na.cpg, na.count, na.eps = a.gn_groups, a.gn_cpg, a.gn_count, a.gn_eps
                na.film, na.film_row, na.film_step = a.film, a.film_row, a.film_step
                na.film_off, na.film_C, na.film_ld = a.film_off, a.film_C, a.film_ld
            nref = C.byref(na)
            nfn = lambda s, nref=nref, lib=lib: L.check(lib.jen1_norm_apply(nref, s), "jen1_norm_apply")
            nfn.label = f"norm_apply[{label}] mode={pro} B={a.B} L={a.L_in} C={ctot}"
            nfn.kind = "norm_apply"
            ops.append(nfn)
            self._keep.append((na, xh))
            a.x0, a.c0, a.ld0 = xh.data_ptr(), ctot, ctot
            a.x1, a.c1, a.ld1 = None, 0, 0
            a.src1_scale = 1.0
            a.pro_mode = L.PRO_NONE
            a.direct = 1
        elif want_direct and pro == L.PRO_NONE:
            a.direct = 1
        if a.direct and a.c1 and a.src1_scale != 1.0:
            a.direct = 0          # a raw scaled second source needs the LDS path (or pre-scaled weights)
        if a.direct:
            a.kc_stage = max(1, (a.c0 + a.c1) // 32)
        if m_split:
            assert a.direct, "a dual-range GEMM needs the streaming (direct) mode"
            a.m_split, a.k_split = m_split, k_split
        if extra_segs and not tiled:
            assert a.direct, "extra K segments need the streaming (direct) mode"
            segs = []
            for tap in range(taps):
                segs.append((a.x0, a.ld0, tap - pad_left, a.c0 // 32))
                if a.c1:
                    segs.append((a.x1, a.ld1, tap - pad_left, a.c1 // 32))
            for e, sh in extra_segs:
                assert e.B == a.B and e.L == a.L_in and e.t.dtype == eng.tdtype
                segs.append((e.t.data_ptr(), e.ld, sh, e.cp // 32))
            assert len(segs) <= L.MAX_SEG
            a.nseg = len(segs)
            for i, (xp, ld, sh, kch) in enumerate(segs):
                a.seg[i].x, a.seg[i].ld, a.seg[i].shift, a.seg[i].kch = xp, ld, sh, kch
        if a.splitk > 1:
            self._splitk_args.append(a)
        # the prepared launch holds raw pointers: keep every tensor it references alive
        self._keep.append((a, src0, src1, w, bias, out, residual, row_scale, gn, film, ln, extra_segs))
        ref = C.byref(a)
        fn = lambda s, ref=ref, lib=lib: L.check(lib.jen1_conv_gemm(ref, s), "jen1_conv_gemm")
        # algorithmic traffic / work of this launch (weights once + conv input + output)
        es = 4 if eng.dt == L.F32 else 2
        c_real = src0.C + (src1.C if src1 is not None else 0)
        c_extra = sum(e.C for e, _ in extra_segs) if extra_segs else 0
        fn.kind = "conv_gemm"
        fn.w_bytes = (taps * c_real + c_extra) * a.M * es
        fn.act_bytes = a.B * a.L_in * (c_real + c_extra) * es + a.B * a.L_y * out_C * (4 if y_f32 else es)
        fn.flops = 2 * (taps * c_real + c_extra) * a.M * a.B * a.L_out
        fn.label = (f"conv[{label}] B={a.B} Lin={a.L_in} Lout={a.L_out} c0={a.c0} c1={a.c1} taps={a.taps} s={a.stride} M={a.M} "
                    f"ps={a.ps_f}/{a.ps_off} Ly={a.L_y} pro={a.pro_mode} cfg={a.cfg} tb={a.tb} nb={a.nb} kst={a.kc_stage} sk={a.splitk} "
                    f"direct={a.direct}")
        ops.append(fn)
        if det_gn:
            self.stats_launch(ops, out)
        if det_rs:
            self.rowstats_launch(ops, out, m_split if m_split else out_C)
        return out

    # ---------------------------------------------------------------- tile phases of the persistent launch (long levels)
    def _conv_tile(self, a: "L.ConvArgs", *, src0, src1, w, bias, out, pro, gn, film, ln, act, residual, row_scale, y_f32, extra_segs,
                   m_split, flat_w, label, taps, out_C):
        eng = self.eng
        if pro not in (L.PRO_NONE, L.PRO_GN, L.PRO_GN_SILU) or act != L.ACT_NONE or row_scale is not None or m_split or out.rs is not None:
            raise DeepIneligible(f"{label}: option outside the tile phases")
        if out.ld != out.C and not y_f32:
            raise DeepIneligible(f"{label}: {out.C} output channels are not a multiple of 32 (padding columns would stay poisoned)")
        if extra_segs and not (len(extra_segs) <= 2 and all(sh == 0 and e.cp == e.C and e.cp % 32 == 0 for e, sh in extra_segs)):
            raise DeepIneligible(f"{label}: extra K segments outside the tile phases")
        if a.M % 128 != 0:
            raise DeepIneligible(f"{label}: {a.M} output rows are not a multiple of 128")
        if self._prog is not None and any(k in self._prog.kinds for k in ("gemm", "attn")):
            # tile phases and GEMM / attention phases are different kernels: the deep program ends here, with the totals of what this
            # layer normalises
            self._prog_close([s_ for s_ in (src0, src1) if s_ is not None and pro in (L.PRO_GN, L.PRO_GN_SILU) and not s_.gn_valid and s_.part is None])
        prog_was_open = self._prog is not None
        self._prog_open()
        prog = self._prog
        a.out_gn_stats = a.out_rowstats = None
        a.dtype = eng.dt                                     # (JEN1_FP8: the long levels stay bf16)
        if pro in (L.PRO_GN, L.PRO_GN_SILU) and film is not None:
            a.film = self.film2.data_ptr()
        a.nseg = 0
        if extra_segs:
            for i, (e, sh) in enumerate(extra_segs):
                assert e.B == a.B and e.L == a.L_in and e.t.dtype == eng.tdtype
                a.seg[i].x, a.seg[i].ld, a.seg[i].shift, a.seg[i].kch = e.t.data_ptr(), e.ld, 0, e.cp // 32
            a.nseg = len(extra_segs)
        srcs_ = [src0] + ([src1] if src1 is not None else []) + [e for e, _ in (extra_segs or [])]
        a.live_mask = sum(1 << i for i, s_ in enumerate(srcs_) if prog.is_live(s_.t)) | \
            (256 if residual is not None and prog.is_live(residual.t) else 0)
        # statistics of the normalised sources: per-tile partials of a tile phase, or the totals somebody wrote before the launch
        st = []
        if pro in (L.PRO_GN, L.PRO_GN_SILU):
            for s_ in [src0] + ([src1] if src1 is not None else []):
                if s_.part is not None:
                    pt, tiles, nfg = s_.part
                    st.append((pt.data_ptr(), tiles, nfg, prog.is_live(pt)))
                elif s_.gn is not None and s_.gn_valid:
                    st.append((s_.gn.data_ptr(), 0, 32, False))
                else:
                    if not prog_was_open:
                        self._prog = None
                    raise DeepIneligible(f"{label}: no GroupNorm statistics for a normalised source")
        # geometry: all output rows in one unit up to 256; the smallest tile that needs the fewest rounds of the resident workgroups
        bm = 256 if a.M % 256 == 0 else 128
        mblocks = a.M // bm
        best = None
        for tb in ((16, 32, 48, 64) if bm == 128 else (16, 32)):
            units = mblocks * a.B * -(-a.L_out // tb)
            rounds = -(-units // max(prog.nwg, 1))
            if best is None or rounds < best[0]:
                best = (rounds, tb)
        tb = best[1]
        part, nfg_out = None, 0
        if out.gn is not None and not y_f32:
            nfg_out = 8 if (out_C % 8 == 0 and (out_C // 8) % 16 == 0) else 0
            if not nfg_out:
                if not prog_was_open:
                    self._prog = None
                raise DeepIneligible(f"{label}: {out_C} output channels do not split into 8 statistics groups of whole M tiles")
            tiles = int(eng.lib.jen1_deep_tile_count(a.L_out, tb, a.M, bm))
            part = torch.empty((a.B, tiles, nfg_out, 2), dtype=torch.float32, device=eng.device)
        try:
            prog.add_tile(a, tb, bm, st, part, nfg_out,
                          f"tile[{label}] B={a.B} Lin={a.L_in} Lout={a.L_out} c0={a.c0} c1={a.c1} taps={a.taps} s={a.stride} M={a.M} tb={tb} bm={bm}", out)
        except DeepIneligible:
            if not prog_was_open:
                self._prog = None
            raise
        self._keep.append((a, src0, src1, w, bias, out, residual, gn, film, extra_segs, part))
        if part is not None:
            out.part = (part, tiles, nfg_out)
        out.gn_valid = False
        es_ = 4 if eng.dt == L.F32 else 2
        c_real_ = src0.C + (src1.C if src1 is not None else 0)
        c_extra_ = sum(e.C for e, _ in extra_segs) if extra_segs else 0
        prog.w_bytes += (taps * c_real_ + c_extra_) * a.M * es_
        prog.act_bytes += a.B * a.L_in * (c_real_ + c_extra_) * es_ + a.B * a.L_y * out_C * (4 if y_f32 else es_)
        prog.flops += 2 * (taps * c_real_ + c_extra_) * a.M * a.B * a.L_out
        return out

    # ---------------------------------------------------------------- phases of the sample-resident long-level launches
    def _conv_long(self, a: "L.ConvArgs", *, src0, src1, w, bias, out, pro, gn, film, act, residual, row_scale, y_f32, extra_segs, m_split, label,
                   taps, out_C):
        eng = self.eng
        if pro not in (L.PRO_NONE, L.PRO_GN, L.PRO_GN_SILU) or act != L.ACT_NONE or row_scale is not None or m_split or out.rs is not None or y_f32:
            raise LongIneligible(f"{label}: option outside the long-level phases")
        if out.ld != out.C:
            raise LongIneligible(f"{label}: {out.C} output channels are not a multiple of 32 (padding columns would stay poisoned)")
        if extra_segs and not (len(extra_segs) <= 2 and all(sh == 0 and e.cp % 32 == 0 for e, sh in extra_segs)):
            raise LongIneligible(f"{label}: extra K segments outside the long-level phases")
        normed = [s_ for s_ in (src0, src1) if s_ is not None and pro in (L.PRO_GN, L.PRO_GN_SILU)]
        if self._prog is not None:
            # the deep program ends here, with the totals of what this layer normalises (tensors a deep phase produced)
            self._prog_close([s_ for s_ in normed if not s_.gn_valid and s_.lp is None])
        prog = self._lprog
        assert prog is not None
        a.out_gn_stats = a.out_rowstats = None
        a.dtype = eng.dt                                     # (JEN1_FP8: the long levels stay bf16)
        if pro in (L.PRO_GN, L.PRO_GN_SILU) and film is not None:
            a.film = self.film2.data_ptr()
        a.nseg = 0
        if extra_segs:
            for i, (e, sh) in enumerate(extra_segs):
                assert e.B == a.B and e.L == a.L_in and e.t.dtype == eng.tdtype
                a.seg[i].x, a.seg[i].ld, a.seg[i].shift, a.seg[i].kch = e.t.data_ptr(), e.ld, 0, e.cp // 32
            a.nseg = len(extra_segs)
        srcs_ = [src0] + ([src1] if src1 is not None else []) + [e for e, _ in (extra_segs or [])]
        a.live_mask = sum(1 << i for i, s_ in enumerate(srcs_) if prog.is_live(s_.t)) | \
            (256 if residual is not None and prog.is_live(residual.t) else 0)
        st, st_live = [], 0
        for k, s_ in enumerate(normed):
            if s_.lp is not None:
                pt, entries, mblocks, nsub = s_.lp
                st.append((pt.data_ptr(), entries, mblocks, nsub))
                st_live |= (1 << k) if prog.is_live(pt) else 0
            elif s_.gn is not None and s_.gn_valid:
                st.append((s_.gn.data_ptr(), 0, 1, 32))
            else:
                raise LongIneligible(f"{label}: no GroupNorm statistics for a normalised source")
        part, lp = None, None
        if out.gn is not None:
            mb, tiles, _ = prog.geometry(a.M, a.L_out)
            if out_C not in (128, 256):
                raise LongIneligible(f"{label}: statistics partials of {out_C} output channels")
            part = torch.empty((a.B, 8, tiles * mb, 2), dtype=torch.float32, device=eng.device)
            lp = (part, tiles * mb, mb, out_C // 16)
        prog.add_long(a, st, st_live, part, f"long[{label}] B={a.B} Lin={a.L_in} Lout={a.L_out} c0={a.c0} c1={a.c1} taps={a.taps} s={a.stride} M={a.M}", out)
        self._keep.append((a, src0, src1, w, bias, out, residual, gn, film, extra_segs, part))
        out.lp = lp
        out.gn_valid = False
        es_ = 4 if eng.dt == L.F32 else 2
        c_real_ = src0.C + (src1.C if src1 is not None else 0)
        c_extra_ = sum(e.C for e, _ in extra_segs) if extra_segs else 0
        prog.w_bytes += (taps * c_real_ + c_extra_) * a.M * es_
        prog.act_bytes += a.B * a.L_in * (c_real_ + c_extra_) * es_ + a.B * a.L_y * out_C * es_
        prog.flops += 2 * (taps * c_real_ + c_extra_) * a.M * a.B * a.L_out
        return out

    def _prog_open(self):
        raise DeepIneligible("no persistent program outside a Plan")

    def _prog_close(self, need_totals=()):
        raise AssertionError("no persistent program outside a Plan")

    # ---------------------------------------------------------------- deterministic statistics (Plan(deterministic=True))
    def stats_launch(self, ops, x: Act):
        """fine-group GroupNorm statistics of ``x`` in a fixed summation order (jen1_gn_stats: written, not accumulated)"""
        lib = self.eng.lib
        a = (x.t.data_ptr(), x.gn.data_ptr(), x.B, x.L, x.ld, self.eng.dt)
        fn = lambda s, a=a, lib=lib: L.check(lib.jen1_gn_stats(*a, s), "jen1_gn_stats")
        fn.kind = "stats"
        fn.label = f"gn_stats[B={x.B} L={x.L} ld={x.ld}]"
        self._keep.append(x)
        ops.append(fn)

    def rowstats_launch(self, ops, x: Act, C: int):
        """per-row LayerNorm sums over the first ``C`` columns of ``x`` (jen1_row_stats: one wave per row, fixed order)"""
        lib = self.eng.lib
        a = (x.t.data_ptr(), x.rs.data_ptr(), x.B * x.L, C, x.ld, self.eng.dt)
        fn = lambda s, a=a, lib=lib: L.check(lib.jen1_row_stats(*a, s), "jen1_row_stats")
        fn.kind = "stats"
        fn.label = f"row_stats[rows={x.B * x.L} C={C}]"
        self._keep.append(x)
        ops.append(fn)

    @staticmethod
    def tile_geometry(B: int, L_out: int, BN: int):
        """positions per tile (balanced over the tiles of a row: 24 positions in 16-wide tiles are 12 + 12,
        not 16 + 8) and batch elements per tile"""
        nt = -(-L_out // BN)
        tb = -(-L_out // nt)
        nb = max(1, min(B, BN // tb))
        return tb, nb, nt * -(-B // nb)

    TILE_CFGS = (L.CFG_T256x32, L.CFG_T256x16, L.CFG_T128x64, L.CFG_T128x32, L.CFG_T128x16, L.CFG_T64x64)

    def pick_tile_cfg(self, B: int, L_out: int, M: int, out_C: int, ps_f: int, lds_ok) -> Optional[int]:
        """long levels (>= 512 positions in the launch): the lean tile kernel, with all output channels in one
        workgroup when M <= 256 and the position tile chosen so that about one workgroup per CU exists"""
        lib, eng = self.eng.lib, self.eng
        if not eng.use_tile_kernel or B * L_out < eng.tile_min_rows:
            return None
        best = None
        bms = (64,) if M <= 64 else (128,) if M <= 128 else (256, 128)
        for cfg in self.TILE_CFGS:
            BM, BN = lib.jen1_cfg_bm(cfg), lib.jen1_cfg_bn(cfg)
            if BM not in bms:
                continue
            if ps_f > 1 and out_C % BM != 0:
                continue
            if BN > L_out and BN > 16:
                continue
            tb = self.tile_geometry(B, L_out, BN)[0]
            if not lds_ok(cfg, tb):
                continue
            wgs = -(-M // BM) * -(-L_out // tb) * B
            tgt = eng.tile_target_wgs
            score = (wgs if wgs <= tgt else tgt - (1 if eng.tile_one_round else 0) * (wgs - tgt) / wgs, -(-(-M // BM)), BN)
            if best is None or score > best[0]:
                best = (score, cfg)
        return None if best is None else best[1]

    def pick_cfg(self, B: int, L_out: int, M: int) -> int:
        """wide tile when the positions alone fill the chip with 64-row M tiles, else a 16-row streaming tile"""
        lib, eng = self.eng.lib, self.eng

        def ntiles(cfg):
            BM, BN = lib.jen1_cfg_bm(cfg), lib.jen1_cfg_bn(cfg)
            return self.tile_geometry(B, L_out, BN)[2] * -(-M // BM)

        wide = L.CFG_W128x64 if M >= 512 else L.CFG_W64x64
        if ntiles(wide) >= eng.target_wgs * 3 // 4:
            return wide
        # streaming: the narrowest tile (least activation traffic through each CU's L1) unless that makes
        # more workgroups than the chip holds at once
        for cfg, bn in ((L.CFG_S16x16, 16), (L.CFG_S16x32, 32)):
            if bn <= eng.stream_bn and (ntiles(cfg) <= eng.stream_max_wgs or bn == eng.stream_bn):
                return cfg
        return L.CFG_S16x64

    def streams(self, B: int, L_out: int, M: int) -> bool:
        """will a plain conv of this shape run on the streaming kernel (where K-segment fusions pay)?"""
        if self._deep_on:
            return True
        if self.eng.use_tile_kernel and B * L_out >= self.eng.tile_min_rows:
            return False
        return self.pick_cfg(B, L_out, M) in (L.CFG_S16x64, L.CFG_S16x32, L.CFG_S16x16)

    def _choose_tiles(self, a: L.ConvArgs, force=None, k_extra=0, tile_ok=False):
        """Tile heuristics.  Wide (W*) tiles when there are enough positions to fill the chip with
        64-row M tiles; otherwise 16-row streaming (S*) tiles whose 4 waves split K: a deep level
        is pure weight streaming and needs many small workgroups, not a big tile."""
        eng = self.eng
        lib = eng.lib
        rows = a.B * a.L_out
        M = a.M
        kch = (a.c0 + a.c1) // 32

        def tiles(cfg):
            BM, BN = lib.jen1_cfg_bm(cfg), lib.jen1_cfg_bn(cfg)
            tb, nb, nt = self.tile_geometry(a.B, a.L_out, BN)
            return BM, BN, tb, nb, nt * -(-M // BM)

        es = 4 if eng.dt == L.F32 else 2

        def tile_lds_ok(cfg, tb):
            rows_in = (tb - 1) * a.stride + a.taps
            ctot = a.c0 + a.c1
            return rows_in * (ctot + 32 * k_extra + 8) * es + 8 * ctot + 8 * (lib.jen1_cfg_bm(cfg) // 2 + 2) + 64 <= 150 * 1024

        if force is not None and "cfg" in force:
            cfg = force["cfg"]
        else:
            cfg = self.pick_tile_cfg(a.B, a.L_out, M, a.out_C, a.ps_f, tile_lds_ok) if tile_ok else None
            if cfg is None:
                cfg = self.pick_cfg(a.B, a.L_out, M)
        BM, BN, tb, nb, wgs = tiles(cfg)
        if cfg in self.TILE_CFGS:
            a.cfg, a.tb, a.nb, a.splitk, a.kc_stage = cfg, tb, 1, 1, 1
            return
        a.cfg, a.tb, a.nb = cfg, tb, nb
        splitk = 1
        if force is not None and "splitk" in force:
            splitk = force["splitk"]
        elif cfg in (L.CFG_S16x64, L.CFG_S16x32, L.CFG_S16x16) and a.pro_mode != L.PRO_SILU:
            # weight streaming needs ~2000 waves with a full prefetch ring in flight to approach HBM
            # bandwidth: split K across workgroups when the M tiles alone give too few of them
            G = a.taps * kch + k_extra          # the streaming kernel splits the flat (segment, chunk) list
            wbytes = G * 32 * M * (4 if eng.dt == L.F32 else 2)
            if wbytes >= eng.splitk_min_bytes and wgs < eng.splitk_target_wgs:
                want = -(-eng.splitk_target_wgs // wgs)
                # every wave keeps >= 2 chunks; 4 waves share one K slice
                max_sk = max(1, G // 8)
                splitk = max(1, min(want, max_sk, kch))
                while splitk > 1 and ((splitk - 1) * -(-G // splitk) >= G or (splitk - 1) * -(-kch // splitk) >= kch):
                    splitk -= 1
        a.splitk = splitk
        cps = -(-kch // splitk)
        # LDS stage: the whole K range of the workgroup in one stage whenever it fits (a second stage
        # costs a barrier and a global round trip); wide tiles stay <= 64 KiB to keep 2 WGs per CU
        limit = 64 * 1024 if cfg in (L.CFG_W64x64, L.CFG_W128x64) else 150 * 1024
        stage = cps
        while True:
            a.kc_stage = stage
            ok_boundary = (not a.c1) or ((a.c0 // 32) % min(stage, cps) == 0)
            nbytes = lib.jen1_conv_gemm_lds_bytes(C.byref(a))
            if ok_boundary and (nbytes <= limit or stage == 1):
                break
            stage -= 1
        assert nbytes <= 160 * 1024, f"LDS {nbytes} B too large (tb={tb}, nb={nb}, taps={a.taps}, stride={a.stride})"
        if splitk > 1:
            self._slab_floats = max(self._slab_floats, wgs * splitk * BM * BN)
            self._max_tiles = max(self._max_tiles, wgs)

    def attention(self, ops, *, q: Act, q_off, kv_t: torch.Tensor, ldkv, k_off, v_off, out: Act, H, d, Nk, causal,
                  kv_row=None, kv_extra=None, extra_row=None, ld_extra=0, kx_off=0, vx_off=0, extra_step=None, fin=None):
        """fin = (rowstats, u, b, ln_C, eps, finish_q, finish_kv): deferred LayerNorm finish (jen1_attention_fin)"""
        eng = self.eng
        rs_, u_, b_, lnC, eps, fq, fkv = fin if fin is not None else (None, None, None, 0, 0.0, 0, 0)
        if self._deep_on:
            kv_live = 1 if kv_row is None and kv_extra is None else 0      # self-attention: K / V come from the previous phase
            assert not kv_live or self.deep.is_live(kv_t)
            kv_live |= 2 if self.deep.is_live(q.t) else 0                  # bit 1: q's tensor was produced inside the launch
            dargs = (q.t.data_ptr(), kv_t.data_ptr(), kv_t.data_ptr(), out.t.data_ptr(), _ptr(kv_row), _ptr(kv_extra),
                     _ptr(extra_row), _ptr(extra_step), ld_extra, kx_off, vx_off, q.B, H, d, q.L, Nk, q.ld, q_off, ldkv, k_off, v_off,
                     out.ld, 1 if causal else 0, float(d) ** -0.5, _ptr(u_), _ptr(b_), lnC, float(eps), fq, fkv, kv_live, eng.deep_dt)
            self._keep.append((q, kv_t, out, kv_row, kv_extra, extra_row, extra_step, fin))
            self.deep.add_attention(dargs, f"attention B={q.B} H={H} d={d} Nq={q.L} Nk={Nk} causal={causal}", out)
            return
        if self._prog is not None:
            self._prog_close()
        args = (q.t.data_ptr(), kv_t.data_ptr(), kv_t.data_ptr(), out.t.data_ptr(), _ptr(kv_row), _ptr(kv_extra),
                _ptr(extra_row), _ptr(extra_step), ld_extra, kx_off, vx_off, q.B, H, d, q.L, Nk, q.ld, q_off, ldkv, k_off, v_off, out.ld,
                1 if causal else 0, float(d) ** -0.5, _ptr(rs_), _ptr(u_), _ptr(b_), lnC, float(eps), fq, fkv, eng.deep_dt)
        lib = eng.lib
        self._keep.append((q, kv_t, out, kv_row, kv_extra, extra_row, extra_step, fin))
        fn = lambda s, args=args, lib=lib: L.check(lib.jen1_attention_fin(*args, s), "jen1_attention")
        fn.label = f"attention B={q.B} H={H} d={d} Nq={q.L} Nk={Nk} causal={causal}"
        ops.append(fn)



class Plan(OpBuilder):
    """Pre-allocated buffers + prepared launches for one (B, T, nrep, causal) shape."""

    def __init__(self, eng: "Engine", B: int, T: int, nrep: int, causal: bool, n_t: Optional[int] = None, deep: bool = True,
                 deterministic: bool = False):
        """n_t = None: one timestep per batch element (the general forward).  n_t = S: *table mode* of a
        sampler -- the timestep-only work (time MLP, FiLM GEMM, time-token K/V GEMM) is evaluated once for
        all S schedule entries (``run_time``) and every kernel of the step indexes the tables through the
        device-side counter ``step_idx``, so a captured step replays with no host-side update.

        deep: run the levels with few positions as ONE persistent launch (DeepProgram).  The first level of that launch
        is the shallowest one whose every layer fits (``deep_level``); levels above it keep one launch per layer.

        deterministic: GroupNorm / LayerNorm statistics of the launch-per-layer levels are summed in a fixed order by a statistics
        launch behind each producer (instead of float atomics in the producers' epilogues): two runs are bit-identical, the step
        is slower by those launches.  The persistent deep-level launch is deterministic either way."""
        self.det = bool(deterministic)
        self.B, self.T, self.nrep, self.causal = B, T, nrep, causal
        self.Beff = B * nrep
        self.table_mode = n_t is not None
        self.n_t = n_t if n_t is not None else B
        lens = eng.spec.level_lengths(T)
        n_lv = len(eng.spec.downs)
        first = n_lv
        if deep and eng.use_deep:
            # candidates: levels whose length is within the persistent kernel's reach, shallowest first
            first = next((i for i in range(n_lv) if lens[i + 1] <= eng.deep_max_len), n_lv)
        self.deep_errors: List[str] = []
        # long levels as tile phases of the persistent launch(es): layers whose input has at least eng.tile_phase_min_len positions
        self.tile_lens = frozenset(l for l in lens if l >= eng.tile_phase_min_len) if (deep and eng.use_deep and eng.use_tile_phases
                                                                                       and not self.det) else frozenset()
        # to_in, the levels above the deep ones and to_out as two sample-resident launches (LongProgram): needs the persistent deep-level launch
        # behind it; a layer that does not fit sends the plan back to one launch per layer on those levels
        self.use_long = bool(deep and eng.use_deep and eng.use_long and not self.tile_lens)
        self.long_errors: List[str] = []
        while True:
            super().__init__(eng)
            self.progs: List[DeepProgram] = []
            self._deep_prog: Optional[DeepProgram] = None
            self._deep_err = None
            self.time_ops: List[Callable[[int], None]] = []
            self.ctx_ops: List[Callable[[int], None]] = []
            self.pack_ops: List[Callable[[int], None]] = []
            self.pack_rows = None
            self.poison_op, self.poison_args = None, None
            self.taps: Dict[str, Act] = {}
            self.acts: List[Act] = []
            self.n_launch = 0
            self.deep_level = first if first < n_lv else None
            try:
                self._build()
                break
            except LongIneligible as e:
                self.long_errors.append(str(e))
                self.use_long = False
            except DeepIneligible as e:
                self.deep_errors.append(f"level {first}: {e}")
                if first >= n_lv:
                    assert self.tile_lens, f"persistent launch: {e}"
                    self.tile_lens = frozenset()           # not even the tile phases alone link: one launch per layer everywhere
                else:
                    first += 1

    # ---------------------------------------------------------------- allocation helpers
    def _stats(self, nfloats: int) -> torch.Tensor:
        o = self._arena_used
        self._arena_used += _ceil_to(nfloats, 64)
        assert self._arena_used <= self.arena.numel(), "statistics arena overflow"
        return self.arena[o: o + nfloats]

    def new_act(self, B, L, C, gn=False, rs=False, dtype=None) -> Act:
        ld = _ceil_to(C, 32)
        if ld != C:
            t = torch.zeros((B, L, ld), dtype=dtype or self.eng.tdtype, device=self.eng.device)
        else:
            t = self._empty((B, L, ld), dtype)
        a = Act(t, B, L, C, ld, self._stats(B * 64) if gn else None, self._stats(B * L * 2) if rs else None)
        self.acts.append(a)
        return a

    # ---------------------------------------------------------------- persistent deep-level launch
    def _prog_open(self):
        """phases are recorded into ONE program as long as consecutive layers are phases (tile phases of the long levels, GEMM /
        attention phases of the deep levels); a layer that runs as a launch closes it (_prog_close) and the next phase opens another"""
        if self._prog is None:
            if self._deep_err is None:
                self._deep_err = torch.zeros((16,), dtype=torch.int32, device=self.eng.device)
            self._prog = DeepProgram(self.eng, leader=self.progs[0] if self.progs else None, err=self._deep_err)
            self.deep = self._prog

    def _long_begin(self):
        """from here on layers are phases of ONE sample-resident launch (LongProgram) until _long_end"""
        self._long_open()               # (a deep program that is still open is closed by the first long layer, with the totals it needs)

    def _long_open(self):
        if self._deep_err is None:
            self._deep_err = torch.zeros((16,), dtype=torch.int32, device=self.eng.device)
        self._lprog = LongProgram(self.eng, self.Beff, leader=self.progs[0] if self.progs else None, err=self._deep_err)
        self._long_on = True

    def _long_end(self):
        prog, self._lprog, self._long_on = self._lprog, None, False
        if prog is None or len(prog) == 0:
            return
        sync = self._stats(prog.sync_words()).view(torch.int32)
        prog.finalize(sync)
        self.progs.append(prog)
        pz = lambda s, prog=prog: prog.poison(s)
        pz.kind = "deep_poison"
        pz.prog = prog
        pz.label = f"long_poison[{prog.poison_tab.shape[0]} rows, {prog.poison_bytes} B]"
        self.ops.insert(1, pz)
        fn = lambda s, prog=prog: prog.launch(s)
        fn.kind = "long"
        fn.prog = prog
        fn.label = f"long[{len(prog)} phases, {prog.Bs} samples x {prog.G} workgroups, {prog.lds} B LDS]"
        fn.w_bytes, fn.act_bytes, fn.flops = prog.w_bytes, prog.act_bytes, prog.flops
        self.ops.append(fn)

    def _deep_begin(self):
        if self._prog is not None and "tile" in self._prog.kinds:
            self._prog_close()        # tile phases and GEMM / attention phases are different kernels (jen1_deep_run_kinds)
        self._prog_open()
        self._deep_prog = self._prog
        self._deep_on = True

    def _deep_end(self, last: Act):
        """the deep levels end here; the program stays open: the next layer either continues it (a tile phase of a long level) or
        closes it (a launch; conv() then asks for the GroupNorm totals of what it normalises)"""
        self._deep_on = False

    def _prog_close(self, need_totals=()):
        """close the recorded program: fine-group totals of the tensors a launch-per-layer consumer normalises next, the
        synchronisation area inside the per-step arena, ONE launch op (+ the poisoning of its tensors at the head of the step)"""
        prog = self._prog
        assert prog is not None and len(prog) > 0
        for x in need_totals:
            prog.add_stats(x, x.gn, f"stats B={x.B} L={x.L} ld={x.ld}")
            x.gn_valid = True
        self._prog = None
        self._deep_on = False
        n = prog.sync_words()
        sync = self._stats(n).view(torch.int32)
        prog.finalize(sync)
        self.progs.append(prog)
        pz = lambda s, prog=prog: prog.poison(s)
        pz.kind = "deep_poison"
        pz.prog = prog
        pz.label = f"deep_poison[{prog.poison_tab.shape[0]} tensors, {prog.poison_bytes} B]"
        self.ops.insert(1, pz)          # right behind the arena reset: long before the launch, after the previous step's last reader
        fn = lambda s, prog=prog: prog.launch(s)
        fn.kind = "deep"
        fn.prog = prog
        fn.label = f"deep[{len(prog)} phases ({prog.kinds.count('tile')} tile), {prog.nwg} workgroups, {prog.lds} B LDS]"
        fn.w_bytes, fn.act_bytes, fn.flops = prog.w_bytes, prog.act_bytes, prog.flops
        self.ops.append(fn)

    def take_error(self) -> int:
        """the error word of the plan's persistent launches (shared; see DeepProgram.take_error)"""
        return self.progs[0].take_error() if self.progs else 0

    # ---------------------------------------------------------------- network blocks
    def resblock(self, r: ResSpec, src0: Act, src1: Optional[Act], causal: bool, gn=True) -> Act:
        """ResnetBlock1d.forward (blocks.py:219-231): 2 launches (+1 for a 1x1 shortcut)."""
        W, ops = self.eng.W, self.ops
        n = r.name
        sc = self.eng.skip_scale if src1 is not None else 1.0
        pad = 2 if causal else 1
        h = self.new_act(src0.B, src0.L, r.c_out, gn=True)
        self.conv(ops, src0=src0, src1=src1, src1_scale=sc, w=W.w[f"{n}.conv1"], bias=W.v[f"{n}.conv1.bias"], out=h,
                  taps=3, pad_left=pad, pro=L.PRO_GN_SILU,
                  gn=(r.groups, r.c_in, W.v[f"{n}.gn1.g"], W.v[f"{n}.gn1.b"], 1e-5))
        y = self.new_act(src0.B, src0.L, r.c_out, gn=gn)
        gn2 = (r.groups, r.c_out, W.v[f"{n}.gn2.g"], W.v[f"{n}.gn2.b"], 1e-5)
        film = (self.film, self.film_row, W.film_off[n], r.c_out, self.step_idx if self.table_mode else None)
        srcs_raw = [src0] + ([src1] if src1 is not None else [])
        # (a source with padded channels -- the 257-channel network input -- rides along only in the sample-resident launches: their extra
        # segments read the padded pitch, the padding columns meet zero weights)
        if r.has_shortcut and self.eng.fuse_shortcut and f"{n}.conv2s" in W.w \
                and (self._long_on or all(s_.cp == s_.C for s_ in srcs_raw)) and 3 + len(srcs_raw) <= L.MAX_SEG:
            # the 1x1 shortcut is one or two more K segments of the second conv (streaming levels, the persistent kernel and the
            # tiled long levels; a wide-tile launch declines and the separate shortcut launch below is used)
            try:
                if self.conv(ops, src0=h, w=W.w[f"{n}.conv2s"], bias=W.v[f"{n}.conv2s.bias"], out=y, taps=3, pad_left=pad,
                             pro=L.PRO_GN_SILU, gn=gn2, film=film, extra_segs=[(s_, 0) for s_ in srcs_raw]) is not None:
                    return y
            except DeepIneligible:
                if not (self._deep_on or self._long_on):
                    raise
                # (the persistent kernel stages the whole batch element: with the block's input riding along a 94-position, 768-channel
                # tile does not fit LDS -- the shortcut becomes a phase of its own below; the same in a sample-resident launch whose tile
                # with the extra segment does not fit: float32, CFG pair of 8, the 288-channel network input)
        if r.has_shortcut:
            res = self.new_act(src0.B, src0.L, r.c_out)
            self.conv(ops, src0=src0, src1=src1, src1_scale=1.0, w=W.w[f"{n}.short"], bias=W.v[f"{n}.short.bias"], out=res)
        else:
            assert src1 is None
            res = src0
        self.conv(ops, src0=h, w=W.w[f"{n}.conv2"], bias=W.v[f"{n}.conv2.bias"], out=y, taps=3, pad_left=pad,
                  pro=L.PRO_GN_SILU, gn=gn2, film=film, residual=res)
        return y

    def transformer(self, t: TransformerSpec, x: Act, causal: bool) -> Act:
        """Transformer1d.forward + TransformerBlock.forward (blocks.py:528-537, :483-489): 10 launches."""
        W, ops, eng = self.eng.W, self.ops, self.eng
        n, Cc, H, d = t.name, t.channels, t.heads, t.head_features
        mid = H * d
        Bf, Lx = x.B, x.L
        gn_in = (32, Cc, W.v[f"{n}.gn.g"], W.v[f"{n}.gn.b"], 1e-6)
        kv = self.kv_ctx[n]
        xattn = dict(kv_t=kv, ldkv=2 * mid, k_off=0, v_off=mid, H=H, d=d, Nk=eng.spec.ctx_len, causal=False, kv_row=self.kv_row,
                     kv_extra=self.kvx.t if eng.spec.use_xattn_time else None,
                     extra_row=self.extra_row if eng.spec.use_xattn_time else None, ld_extra=W.kvx_ld,
                     kx_off=W.kvx_off[n], vx_off=W.kvx_off[n] + mid,
                     extra_step=self.step_idx if (self.table_mode and eng.spec.use_xattn_time) else None)
        fold1 = (eng.fold_single_position and Lx == 1 and eng.fuse_ln_proj and self.streams(Bf, Lx, Cc) and Cc % 256 == 0 and mid % 32 == 0
                 and Cc <= 1024)
        a1 = None if fold1 else self.new_act(Bf, Lx, mid)
        a2 = self.new_act(Bf, Lx, mid)
        if fold1:
            # one position: self-attention == to_out(to_v(norm_context(x1))) exactly (see Weights: s1q2); two GEMMs instead of
            # GEMM -> attention -> GEMM, and the first one is 4x narrower (no q / k / v rows)
            x1 = self.new_act(Bf, Lx, Cc, gn=True)
            self.conv(ops, src0=x, w=W.w[f"{n}.proj"], bias=W.v[f"{n}.proj.bias"], out=x1, pro=L.PRO_GN, gn=gn_in)
            xq2 = self.new_act(Bf, Lx, Cc + mid, rs=True)
            self.conv(ops, src0=x1, w=W.w[f"{n}.s1q2"], bias=W.v[f"{n}.o1q2.bias"], out=xq2, pro=L.PRO_GN,
                      gn=(1, Cc, W.v[f"{n}.nc.g"], W.v[f"{n}.nc.b"], 1e-5), residual=x1, extra_segs=[(x1, 0)], m_split=Cc, k_split=Cc // 32,
                      flat_w=True)
            x2 = xq2.cols(0, Cc, rs=xq2.rs)
            self.attention(ops, q=xq2, q_off=Cc, out=a2, fin=(xq2.rs, W.v[f"{n}.o1q2.u"], W.v[f"{n}.o1q2.b"], Cc, 1e-5, 1, 0), **xattn)
        elif eng.fuse_ln_proj and self.streams(Bf, Lx, Cc) and Cc % 32 == 0 and mid % 32 == 0 and Lx * (d // 8) <= 512:
            # 5 launches instead of 7: each projection rides with the LayerNorm-folded projection that follows it,
            # the LayerNorm finish is applied by the attention kernel (jen1_attention_fin)
            xq1 = self.new_act(Bf, Lx, Cc + 3 * mid, rs=True)
            self.conv(ops, src0=x, w=W.w[f"{n}.pqkv"], bias=W.v[f"{n}.pqkv.bias"], out=xq1, pro=L.PRO_GN, gn=gn_in,
                      m_split=Cc, k_split=Cc // 32, flat_w=True)
            x1 = xq1.cols(0, Cc, rs=xq1.rs)
            self.attention(ops, q=xq1, q_off=Cc, kv_t=xq1.t, ldkv=xq1.ld, k_off=Cc + mid, v_off=Cc + 2 * mid, out=a1, H=H, d=d, Nk=Lx,
                           causal=causal, fin=(xq1.rs, W.v[f"{n}.pqkv.u"], W.v[f"{n}.pqkv.b"], Cc, 1e-5, 1, 1))
            xq2 = self.new_act(Bf, Lx, Cc + mid, rs=True)
            self.conv(ops, src0=a1, w=W.w[f"{n}.o1q2"], bias=W.v[f"{n}.o1q2.bias"], out=xq2, residual=x1,
                      extra_segs=[(x1, 0)], m_split=Cc, k_split=mid // 32)
            x2 = xq2.cols(0, Cc, rs=xq2.rs)
            self.attention(ops, q=xq2, q_off=Cc, out=a2, fin=(xq2.rs, W.v[f"{n}.o1q2.u"], W.v[f"{n}.o1q2.b"], Cc, 1e-5, 1, 0), **xattn)
        else:
            x1 = self.new_act(Bf, Lx, Cc, rs=True)
            self.conv(ops, src0=x, w=W.w[f"{n}.proj"], bias=W.v[f"{n}.proj.bias"], out=x1, pro=L.PRO_GN, gn=gn_in)
            qkv = self.new_act(Bf, Lx, 3 * mid)
            self.conv(ops, src0=x1, w=W.w[f"{n}.qkv"], bias=W.v[f"{n}.qkv.bias"], out=qkv, pro=L.PRO_LN,
                      ln=(Cc, None, None, W.v[f"{n}.qkv.u"]))
            self.attention(ops, q=qkv, q_off=0, kv_t=qkv.t, ldkv=qkv.ld, k_off=mid, v_off=2 * mid, out=a1, H=H, d=d, Nk=Lx,
                           causal=causal)
            x2 = self.new_act(Bf, Lx, Cc, rs=True)
            self.conv(ops, src0=a1, w=W.w[f"{n}.o1"], bias=W.v[f"{n}.o1.bias"], out=x2, residual=x1)
            q2 = self.new_act(Bf, Lx, mid)
            self.conv(ops, src0=x2, w=W.w[f"{n}.q2"], bias=W.v[f"{n}.q2.bias"], out=q2, pro=L.PRO_LN,
                      ln=(Cc, None, None, W.v[f"{n}.q2.u"]))
            self.attention(ops, q=q2, q_off=0, out=a2, **xattn)
        Cf = Cc * t.multiplier
        if eng.fuse_o2_ff1 and eng.fuse_ff_out and self.streams(Bf, Lx, Cc) and Cc % 32 == 0 and mid % 32 == 0 and x2.cp == Cc:
            xf = self.new_act(Bf, Lx, Cc + Cf)
            self.conv(ops, src0=a2, w=W.w[f"{n}.o2f1"], bias=W.v[f"{n}.o2f1.bias"], out=xf, residual=x2, act=L.ACT_GELU,
                      extra_segs=[(x2, 0)], m_split=Cc, k_split=mid // 32)
            x3, f1 = xf.cols(0, Cc), xf.cols(Cc, Cf)
        else:
            x3 = self.new_act(Bf, Lx, Cc)
            self.conv(ops, src0=a2, w=W.w[f"{n}.o2"], bias=W.v[f"{n}.o2.bias"], out=x3, residual=x2)
            f1 = self.new_act(Bf, Lx, Cf)
            self.conv(ops, src0=x3, w=W.w[f"{n}.ff1"], bias=W.v[f"{n}.ff1.bias"], out=f1, act=L.ACT_GELU)
        y = self.new_act(Bf, Lx, Cc, gn=True)
        if eng.fuse_ff_out and self.streams(Bf, Lx, Cc) and x3.cp == Cc and f1.cp == f1.C:
            self.conv(ops, src0=x3, w=W.w[f"{n}.ffp"], bias=W.v[f"{n}.ffp.bias"], out=y, extra_segs=[(f1, 0)])
            return y
        x4 = self.new_act(Bf, Lx, Cc)
        self.conv(ops, src0=f1, w=W.w[f"{n}.ff2"], bias=W.v[f"{n}.ff2.bias"], out=x4, residual=x3)
        self.conv(ops, src0=x4, w=W.w[f"{n}.proj"], bias=W.v[f"{n}.proj.bias"], out=y)
        return y

    # ---------------------------------------------------------------- whole network
    def _build(self):
        eng, spec, W = self.eng, self.eng.spec, self.eng.W
        lib, dev = eng.lib, eng.device
        B, Be, T, causal = self.B, self.Beff, self.T, self.causal
        f32 = torch.float32
        n_tr = len(spec.transformers())
        lens = spec.level_lengths(T)
        Ltr = max([lens[i + 1] for i, d in enumerate(spec.downs) if d.transformer] + [lens[-1]])
        deep_sync = 4 * (int(eng.lib.jen1_deep_sync_bytes(256)) // 4) if (self.deep_level is not None or self.tile_lens) else 0    # <= 4 programs of <= 256 phases
        self.arena = torch.zeros(600 * Be * 64 + (4 * n_tr + 8) * Be * Ltr * 2 + 4096 + deep_sync + 64, dtype=f32, device=dev)
        self._arena_used = 0
        ops = self.ops

        # ---- static inputs --------------------------------------------------------------------
        Cx, Cc = spec.in_channels, spec.ctx_ch0
        self.x_in = torch.zeros((B, Cx, T), dtype=f32, device=dev)
        self.ctx_in = torch.zeros((B, max(Cc, 1), T), dtype=f32, device=dev)
        NT_ = self.n_t
        self.t_in = torch.zeros((NT_,), dtype=torch.int64, device=dev)
        # continuous times (VDM: t in [0, 1], vdm/vdm.py:44): ``t_float`` switches the two time-feature launches to this buffer
        self.t_in_f = torch.zeros((NT_,), dtype=torch.float32, device=dev)
        self.t_float = False
        self.step_idx = torch.zeros((1,), dtype=torch.int32, device=dev)
        F, NL = spec.ctx_features, spec.ctx_max_length
        self.emb_in = torch.zeros((B, NL, F), dtype=f32, device=dev)
        self.mask_in = torch.ones((B, spec.ctx_len), dtype=f32, device=dev)
        ar = torch.arange(B, dtype=torch.int32, device=dev)
        self.film_row = (torch.arange(Be, dtype=torch.int32, device=dev) % B).contiguous()
        # rows [0,B): conditional half reads text slot b + per-step time row; rows [B,2B): fixed slot
        self.kv_row = torch.cat([ar, ar + B])[:Be].contiguous() if self.nrep == 2 else ar.clone()
        self.extra_row = torch.cat([ar, torch.full_like(ar, -1)])[:Be].contiguous() if self.nrep == 2 else ar.clone()

        arena_bytes = self.arena.numel() * 4
        arena_ptr = self.arena.data_ptr()
        ops.append(lambda s: L.check(lib.jen1_memset_zero(arena_ptr, arena_bytes, s), "memset"))

        # ---- 1. pack [B,C,T] + context channels -> channel-last, CFG pair replicated -------------
        X0 = self.new_act(Be, T, Cx + Cc, gn=True)
        if eng.pack_fixed_order:
            # the GroupNorm sums of the network input in a fixed order (per-block partials + one small launch that adds them): with the
            # persistent launches' fixed-order statistics the whole default step is run-to-run bit-reproducible
            self._pack_parts = torch.empty((B, (T + 31) // 32, X0.ld, 2), dtype=f32, device=dev)
            # (statistics that are written, not accumulated: outside the arena that the head of every step zeroes, so that the previous
            # step's last kernel may already have written them -- DDIMStepper's fused step)
            X0.gn = torch.zeros((Be * 64,), dtype=f32, device=dev)
            a = (self.x_in.data_ptr(), self.ctx_in.data_ptr() if Cc else None, X0.t.data_ptr(), self._pack_parts.data_ptr(), B, Cx, Cc, T,
                 X0.ld, self.nrep, eng.dt)
            pk = lambda s, a=a: L.check(lib.jen1_pack_input_parts(*a, s), "jen1_pack_input_parts")
            a2 = (self._pack_parts.data_ptr(), X0.gn.data_ptr(), B, T, X0.ld, self.nrep)
            st = lambda s, a=a2: L.check(lib.jen1_gn_stats_from_parts(*a, s), "jen1_gn_stats_from_parts")
            pk.kind = st.kind = "pack"
            pk.label, st.label = "pack_input_parts", "gn_stats_from_parts"
            ops += [pk, st]
            self.pack_ops, self.pack_stats_op = [pk, st], st
            self.pack_rows = (X0.t.data_ptr(), self._pack_parts.data_ptr(), X0.ld)      # where a fused step kernel writes the next input
        else:
            a = (self.x_in.data_ptr(), self.ctx_in.data_ptr() if Cc else None, X0.t.data_ptr(), None if self.det else X0.gn.data_ptr(), B, Cx, Cc, T,
                 X0.ld, self.nrep, eng.dt)
            ops.append(lambda s, a=a: L.check(lib.jen1_pack_input(*a, s), "jen1_pack_input"))
            if self.det:
                self.stats_launch(ops, X0)

        # ---- 2. time -> mapping -> FiLM scale/shift of all ResBlocks (model.py:204-223) -------------
        # (timestep-only work: ``time_ops``; one row per batch element, or per schedule entry in table mode)
        tops = self.time_ops
        mf, half = spec.mapping_features, spec.channels // 2
        tf = torch.empty((NT_, mf), dtype=f32, device=dev)
        m1 = torch.empty((NT_, mf), dtype=f32, device=dev)
        self.mapping = torch.empty((NT_, mf), dtype=f32, device=dev)
        self._keep += [tf, m1]          # referenced by raw pointer below
        v = W.v
        a = (v["to_time.0.0.weights"].data_ptr(), v["to_time.0.1.weight"].data_ptr(), v["to_time.0.1.bias"].data_ptr(), tf.data_ptr(), NT_, half, mf)
        tops.append(lambda s, a=a: self._time_features(a, s))
        a = (tf.data_ptr(), v["to_mapping.0.weight"].data_ptr(), v["to_mapping.0.bias"].data_ptr(), m1.data_ptr(), NT_, mf, mf, L.ACT_GELU)
        tops.append(lambda s, a=a: L.check(lib.jen1_linear_f32(*a, s), "jen1_linear_f32"))
        a = (m1.data_ptr(), v["to_mapping.2.weight"].data_ptr(), v["to_mapping.2.bias"].data_ptr(), self.mapping.data_ptr(), NT_, mf, mf, L.ACT_GELU)
        tops.append(lambda s, a=a: L.check(lib.jen1_linear_f32(*a, s), "jen1_linear_f32"))
        map_t = Act(self._empty((1, NT_, mf)), 1, NT_, mf, mf)
        self._add_cast(tops, self.mapping, map_t.t)
        self.film = torch.empty((NT_, W.film_ld), dtype=f32, device=dev)
        film_act = Act(self.film.view(1, NT_, W.film_ld), 1, NT_, W.film_ld, W.film_ld)
        self.conv(tops, src0=map_t, w=W.w["film"], bias=W.v["film.bias"], out=film_act, pro=L.PRO_SILU, y_f32=True)
        # fused GroupNorm-FiLM table of the persistent kernel: y = xhat * gamma (scale + 1) + beta (scale + 1) + shift
        # (blocks.py:141-143); timestep-only work like the FiLM GEMM itself
        self.film2 = torch.empty_like(self.film)
        if self.deep_level is not None or self.tile_lens:
            film, film2, fa, fb, fp = self.film, self.film2, W.film2_a, W.film2_b, W.film2_partner
            tmp_a, tmp_b = torch.empty_like(film), torch.empty_like(film)

            def film2_op(s, film=film, film2=film2, fa=fa, fb=fb, fp=fp, tmp_a=tmp_a, tmp_b=tmp_b, dev=dev):
                # torch elementwise plumbing on the stream the other time ops were given (not necessarily torch's current one)
                cur = torch.cuda.current_stream(dev)
                ctx = torch.cuda.stream(torch.cuda.ExternalStream(s, device=dev)) if s != cur.cuda_stream else contextlib.nullcontext()
                with ctx:
                    torch.index_select(film, 1, fp, out=tmp_a)
                    tmp_a.add_(1.0).mul_(fa)
                    torch.mul(film, fb, out=tmp_b)
                    torch.add(tmp_a, tmp_b, out=film2)
            tops.append(film2_op)

        # ---- 3. time token of the text context -> its K/V row for every cross-attention ------------
        self.kv_ctx: Dict[str, torch.Tensor] = {}
        if n_tr:
            for t in spec.transformers():
                self.kv_ctx[t.name] = torch.zeros((2 * B, spec.ctx_len, 2 * t.heads * t.head_features),
                                                  dtype=eng.tdtype, device=dev)
        self.kvx = None
        if spec.use_xattn_time and n_tr:
            tok = torch.empty((NT_, F), dtype=f32, device=dev)
            self._keep.append(tok)
            a = (v["to_time_embedding.0.0.weights"].data_ptr(), v["to_time_embedding.0.1.weight"].data_ptr(),
                 v["to_time_embedding.0.1.bias"].data_ptr(), tok.data_ptr(), NT_, half, F)
            tops.append(lambda s, a=a: self._time_features(a, s))
            tok_rs = torch.zeros((NT_ * 2,), dtype=f32, device=dev)     # outside the per-step arena
            tok_t = Act(self._empty((1, NT_, F)), 1, NT_, F, F, rs=tok_rs)
            self._add_cast(tops, tok, tok_t.t)
            a = (tok_t.t.data_ptr(), tok_t.rs.data_ptr(), NT_, F, F, eng.dt)
            tops.append(lambda s, a=a: L.check(lib.jen1_row_stats(*a, s), "jen1_row_stats"))
            self.kvx = Act(self._empty((1, NT_, W.kvx_ld)), 1, NT_, W.kvx_ld, W.kvx_ld)
            self.conv(tops, src0=tok_t, w=W.w["kvx"], bias=W.v["kvx.bias"], out=self.kvx, pro=L.PRO_LN,
                      ln=(F, None, None, W.v["kvx.u"]))

        # ---- 4. UNet1d.forward (model.py:243-262) ------------------------------------------------
        n_lv = len(spec.downs)
        first_tr = next((i for i, d in enumerate(spec.downs) if d.transformer), n_lv)
        # levels [0, long_levels) (+ to_in / to_out) run as phases of the two sample-resident launches
        self.long_levels = min(self.deep_level, first_tr) if (self.use_long and self.deep_level is not None) else 0
        use_long = self.long_levels >= 1
        if use_long:
            self._long_begin()
        x = self.resblock(spec.to_in, X0, None, causal=False)
        self.taps["to_in"] = x
        skip0 = x
        skips_list: List[List[Act]] = []
        n_lv = len(spec.downs)
        for i, d in enumerate(spec.downs):
            if use_long and i == self.long_levels:
                self._long_end()
            if i == self.deep_level:
                self._deep_begin()        # from this level's downsampling conv on, layers are phases of one launch
            f, k = d.factor, d.kernel
            Lo = (x.L + f - 1) // f
            y = self.new_act(Be, Lo, d.c_out, gn=True)
            self.conv(ops, src0=x, w=W.w[f"{d.name}.down"], bias=W.v[f"{d.name}.down.bias"], out=y, taps=k, stride=f,
                      pad_left=(k - 1) if causal else (k - 1) // 2, L_out=Lo)
            x = y
            skips = []
            for r in d.blocks:
                x = self.resblock(r, x, None, causal)
                skips.append(x)
            if d.transformer:
                x = self.transformer(d.transformer, x, causal)
                skips.append(x)
            skips_list.append(skips)
            self.taps[f"down{i}"] = x
        x = self.resblock(spec.bott_pre, x, None, causal)
        if spec.bott_tr:
            x = self.transformer(spec.bott_tr, x, causal)
        x = self.resblock(spec.bott_post, x, None, causal)
        self.taps["bottleneck"] = x
        for idx, u in enumerate(spec.ups):
            if use_long and n_lv - 1 - idx == self.long_levels - 1:
                self._long_begin()
            skips = skips_list.pop()
            for r in u.blocks:
                sk = skips.pop()
                assert x.L == sk.L, (x.L, sk.L)      # crop already folded into the producing upsample
                x = self.resblock(r, x, sk, causal)
            if u.transformer:
                x = self.transformer(u.transformer, x, causal)
            f = u.factor
            last = idx == len(spec.ups) - 1
            # length the next consumer needs: the skips of the next level (or T at the top)
            # (a next level without blocks or transformer has no skip to crop against: its upsample takes the full f * L)
            L_need = skip0.L if last else (skips_list[-1][-1].L if skips_list[-1] else f * x.L)
            y = self.new_act(Be, L_need, u.c_out, gn=True)
            if f == 1:
                assert L_need == x.L
                self.conv(ops, src0=x, w=W.w[f"{u.name}.up"], bias=W.v[f"{u.name}.up.bias"], out=y, taps=3, pad_left=1,
                          residual=skip0 if last else None)
            else:
                p = f // 2 + f % 2
                diff = f * x.L - L_need
                assert diff >= 0
                self.conv(ops, src0=x, w=W.w[f"{u.name}.up"], bias=W.v[f"{u.name}.up.bias"], out=y, taps=2, pad_left=1,
                          L_out=x.L + 1, ps_f=f, ps_off=p + diff // 2, out_C=u.c_out,
                          residual=skip0 if last else None)
            x = y
            self.taps[f"up{idx}"] = x
            if self.deep_level is not None and n_lv - 1 - idx == self.deep_level:
                self._deep_end(x)         # this level's upsampling conv was the last phase
        out = self.resblock(spec.to_out, x, None, causal=False, gn=False)
        self.net_out = out
        self.taps["out"] = out
        if use_long:
            self._long_end()
        if self._prog is not None:
            self._prog_close()            # (the network's output is read by a launch that normalises nothing)
        self.deep = self._deep_prog if self._deep_prog is not None else (self.progs[-1] if self.progs else None)
        pzs = [op for op in self.ops if getattr(op, "kind", "") == "deep_poison"]
        if pzs and arena_bytes % 16 == 0 and arena_ptr % 16 == 0:
            # ONE node at the head of the step: the arena reset and the poisoning of every persistent launch's tensors (every program's
            # synchronisation words live inside the arena: zeroed with it)
            tab = torch.cat([op.prog.poison_tab for op in pzs], 0).contiguous()
            self._poison_tab = tab
            sync0 = pzs[0].prog.sync.data_ptr()
            fused = lambda s, a=(tab.data_ptr(), tab.shape[0], sync0, arena_ptr, arena_bytes): L.check(lib.jen1_deep_poison_zero(*a, s), "jen1_deep_poison_zero")
            fused.kind, fused.prog = "deep_poison", pzs[0].prog
            fused.label = f"poison[{tab.shape[0]} rows, {sum(op.prog.poison_bytes for op in pzs)} B] + arena reset"
            # (a fused sampler runs this node's job in the tail launch of the previous step: DDIMStepper, jen1_step_tail)
            # -- without the rows of the network's output: that launch READS it (nobody polls it: its sentinels are not needed at all)
            lo = out.t.data_ptr()
            hi = lo + out.t.numel() * out.t.element_size()
            rows_host = tab.cpu()
            keep = [(int(p_), int(n_)) for p_, n_ in rows_host.tolist() if not (lo <= int(p_) < hi)]
            assert all(p_ + n_ <= lo or p_ >= hi for p_, n_ in keep)
            self._tail_tab = torch.tensor(keep, dtype=rows_host.dtype).to(dev).contiguous()
            self.poison_op, self.poison_args = fused, (self._tail_tab.data_ptr(), self._tail_tab.shape[0], sync0, arena_ptr, arena_bytes)
            self.ops = [fused] + [op for op in self.ops[1:] if getattr(op, "kind", "") != "deep_poison"]

        # ---- 5. context ops: text K/V (hoisted out of the step loop) -------------------------------
        if n_tr:
            cops = self.ctx_ops
            # (blocks.py:427-434: k, v = to_kv(norm_context(context)) * mask.)  One standardisation of the B * 128 text rows, ONE
            # grouped matrix-core GEMM for the 13 layers (LayerNorm's gamma / beta live in the stacked weights / the group biases),
            # mask and row map into the [2B][129][2C] caches in its epilogue, one launch for the unconditional slots.
            self.emb_t = Act(self._empty((B, NL, F)), B, NL, F, F)
            a = (self.emb_t.t.data_ptr(), B * NL, F, F, F, 1e-5, eng.dt)
            # (the source pointer is the caller's float32 embedding when it can be read in place, else the plan's copy: _ctx_src)
            cops.append(lambda s, a=a: L.check(lib.jen1_standardize_rows(self._ctx_src, *a, s), "jen1_standardize_rows"))
            groups, fixed_rows = [], []
            for t in spec.transformers():
                kv = self.kv_ctx[t.name]
                mid2 = kv.shape[-1]
                groups.append((kv.data_ptr(), W.v[f"{t.name}.kv2.bias"].data_ptr(), W.kvx_off[t.name], mid2, mid2))
                fixed_rows.append((eng.kv_fixed[t.name].data_ptr(), kv[B:].data_ptr(), mid2, 0))
            self._kv_gemms = []
            for part in _bgemm_parts(groups):
                # (one launch for all layers whenever every layer's width is a multiple of the 128-column tile; else one per layer)
                tab = L.bgemm_group_table([(c, b_, n0 - part[0][2], N, ldc) for c, b_, n0, N, ldc in part], dev)
                g = L.BGemmArgs()
                n_lo, n_hi = part[0][2], part[-1][2] + part[-1][3]
                g.a, g.groups, g.row_scale = self.emb_t.t.data_ptr(), tab.data_ptr(), self.mask_in.data_ptr()
                g.b = W.w["kv2_all"].data_ptr() + n_lo * F * W.w["kv2_all"].element_size()
                g.M, g.Ntot, g.K, g.lda, g.ldb, g.n_groups = B * NL, n_hi - n_lo, F, F, F, len(part)
                g.rows_in, g.rows_out, g.dtype, g.alpha = NL, spec.ctx_len, eng.dt, 1.0
                g.group_align = _group_align(part)      # (256-column multiples: the product may run on the 256 x 256 tile form)
                self._kv_gemms.append((g, tab))
                fn = lambda s, g=g: L.check(lib.jen1_big_gemm(C.byref(g), s), "jen1_big_gemm")
                fn.kind, fn.label = "big_gemm", f"to_kv of {len(part)} layers: [{B * NL} x {F}] x [{n_hi - n_lo} x {F}]^T"
                fn.flops = 2.0 * B * NL * (n_hi - n_lo) * F
                fn.bytes = (B * NL * F + (n_hi - n_lo) * F + B * NL * (n_hi - n_lo)) * W.w["kv2_all"].element_size()
                cops.append(fn)
            self._kv_fixed_tab = torch.tensor(fixed_rows, dtype=torch.int64).to(dev)
            a = (self._kv_fixed_tab.data_ptr(), len(fixed_rows), self.mask_in.data_ptr(), B, spec.ctx_len, eng.dt)
            cops.append(lambda s, a=a: L.check(lib.jen1_kv_fixed_fill(*a, s), "jen1_kv_fixed_fill"))

        # ---- split-K workspace shared by all launches of the plan (stream-ordered reuse) -----------
        self.finalize_workspace()
        self.n_launch = len(self.ops) + (0 if self.table_mode else len(self.time_ops))

    def _time_features(self, a, s):
        lib = self.eng.lib
        if self.t_float:
            L.check(lib.jen1_time_features_f32(self.t_in_f.data_ptr(), *a, s), "jen1_time_features_f32")
        else:
            L.check(lib.jen1_time_features(self.t_in.data_ptr(), *a, s), "jen1_time_features")

    def set_times(self, time: torch.Tensor):
        """raw int64 timesteps (GaussianDiffusion) or continuous float times (VDM) of the next ``run`` / ``run_time``"""
        if time.is_floating_point():
            self.t_in_f.copy_(time.to(torch.float32))
            self.t_float = True
        else:
            self.t_in.copy_(time.to(torch.int64))
            self.t_float = False

    def run_time(self, stream: Optional[int] = None):
        """timestep-only work (time MLP -> FiLM table, time token -> K/V rows) for every entry of t_in"""
        if stream is None:
            stream = torch.cuda.current_stream(self.eng.device).cuda_stream
        for op in self.time_ops:
            op(stream)

    def run(self, stream: Optional[int] = None, pack: bool = True, poison: bool = True):
        if stream is None:
            stream = torch.cuda.current_stream(self.eng.device).cuda_stream
        if not self.table_mode:
            self.run_time(stream)       # general forward: the timesteps change with every call
        super().run(stream, pack, poison)

    def run_pack(self, stream: Optional[int] = None):
        """only the ops that write the network input from x_in / ctx_in (a fused sampler runs them once per trajectory: its step kernel
        writes the next step's input itself, DDIMStepper)"""
        if stream is None:
            stream = torch.cuda.current_stream(self.eng.device).cuda_stream
        for op in self.pack_ops:
            op(stream)

    def run_poison(self, stream: Optional[int] = None):
        """only the head-of-step sentinel / arena reset node (a fused sampler runs it once per trajectory; jen1_step_tail after that)"""
        if stream is None:
            stream = torch.cuda.current_stream(self.eng.device).cuda_stream
        self.poison_op(stream)

    def _add_cast(self, ops, src: torch.Tensor, dst: torch.Tensor):
        """dtype cast through torch (device plumbing, captured like any other node)."""
        ops.append(lambda s, src=src, dst=dst: dst.view(src.shape).copy_(src))

    # ---------------------------------------------------------------- running
    def set_context(self, embedding: torch.Tensor, mask: Optional[torch.Tensor], stream: int):
        """Project the text tokens' K/V for every cross-attention layer (once per conditioning)."""
        eng, B = self.eng, self.B
        if not self.kv_ctx:
            return
        # the standardisation launch reads B * ctx_max_length * features floats through a raw pointer: the shape is checked HERE (a
        # shorter token axis or another batch would be an out-of-range device read, not an exception).  The plan's K/V cache, masks
        # and cross-attention units are laid out for exactly ctx_max_length text tokens (+ the time token); the conditioner pads to it
        # (conditioners.py:84-111: tokenizer max_length = 128)
        if tuple(embedding.shape) != tuple(self.emb_in.shape):
            raise ValueError(f"cross-attention embedding of shape {tuple(embedding.shape)}: this plan was built for {tuple(self.emb_in.shape)} "
                             "(batch, context_embedding_max_length, context_embedding_features); pad the tokens to the maximum length with a False mask")
        if mask is not None and tuple(mask.shape) != (B, eng.spec.ctx_max_length):
            raise ValueError(f"embedding_mask of shape {tuple(mask.shape)}: expected {(B, eng.spec.ctx_max_length)}")
        if embedding.dtype == torch.float32 and embedding.is_contiguous() and embedding.device == self.emb_in.device:
            self._ctx_src, self._ctx_keep = embedding.data_ptr(), embedding       # read in place by the standardisation launch
        else:
            self.emb_in.copy_(embedding)
            self._ctx_src, self._ctx_keep = self.emb_in.data_ptr(), None
        if mask is not None:
            self.mask_in[:, : eng.spec.ctx_max_length].copy_(mask)                # (bool -> float in the copy; column 128, the time token, stays 1)
        else:
            self.mask_in.fill_(1.0)
        # three launches: standardise, the grouped to_kv GEMM, the unconditional slots (learned fixed embedding masked with the SAME
        # text mask, model.py:337)
        for op in self.ctx_ops:
            op(stream)

    def set_rows(self, drop_rows: Optional[torch.Tensor], uncond_only: bool = False):
        """Select, per effective batch row, which cached K/V slot cross-attention reads.
        drop_rows[b] = True swaps row b to the fixed embedding (CFG dropout, model.py:323-328)."""
        B = self.B
        ar = torch.arange(B, dtype=torch.int32, device=self.eng.device)
        if uncond_only:
            cond_row, cond_extra = ar + B, torch.full_like(ar, -1)
        elif drop_rows is None:
            cond_row, cond_extra = ar, ar
        else:
            d = drop_rows.to(device=self.eng.device, dtype=torch.bool)
            cond_row = torch.where(d, ar + B, ar)
            cond_extra = torch.where(d, torch.full_like(ar, -1), ar)
        if self.nrep == 2:
            self.kv_row.copy_(torch.cat([cond_row, ar + B]))
            self.extra_row.copy_(torch.cat([cond_extra, torch.full_like(ar, -1)]))
        else:
            self.kv_row.copy_(cond_row)
            self.extra_row.copy_(cond_extra)


class Engine:
    def __init__(self, spec: UNetSpec, params: Dict[str, torch.Tensor], dtype: str = "bf16", device="cuda"):
        assert dtype in ("f32", "bf16", "fp8")
        self.lib = L.load()
        self.spec = spec
        self.device = torch.device(device)
        self.dt = L.F32 if dtype == "f32" else L.BF16
        self.tdtype = torch.float32 if dtype == "f32" else torch.bfloat16
        # "fp8" (BASELINE configs[4]): activations and the launch-per-layer levels as in "bf16"; the persistent deep-level launch --
        # 94 % of the weights at T = 1500, 87 % at T = 9000 -- runs its GEMM and attention units on OCP e4m3 operands
        self.deep_dt = L.FP8 if dtype == "fp8" else self.dt
        self.skip_scale = 2 ** -0.5 if spec.use_skip_scale else 1.0
        self.target_wgs = 256
        self.splitk_target_wgs = int(os.environ.get("JEN1_SPLITK_WGS", "128"))
        self.splitk_min_bytes = int(os.environ.get("JEN1_SPLITK_MIN_BYTES", str(8 << 20)))
        self.stream_bn = int(os.environ.get("JEN1_STREAM_BN", "64"))
        self.stream_max_wgs = int(os.environ.get("JEN1_STREAM_MAX_WGS", "768"))
        self.fuse_shortcut = os.environ.get("JEN1_FUSE_SHORTCUT", "1") != "0"
        self.fuse_shortcut_tiles = os.environ.get("JEN1_FUSE_SHORTCUT_TILES", "1") != "0"      # ... also on the tiled long levels
        self.fuse_ff_out = os.environ.get("JEN1_FUSE_FF_OUT", "1") != "0"
        self.fuse_o2_ff1 = os.environ.get("JEN1_FUSE_O2_FF1", "1") != "0"
        self.fuse_ln_proj = os.environ.get("JEN1_FUSE_LN_PROJ", "1") != "0"
        self.fold_single_position = os.environ.get("JEN1_FOLD_N1", "1") != "0"        # Nq = Nk = 1 self-attention as one GEMM (exact)
        self.use_tile_kernel = os.environ.get("JEN1_TILE_KERNEL", "1") != "0"
        self.tile_min_rows = int(os.environ.get("JEN1_TILE_MIN_ROWS", "512"))
        self.tile_target_wgs = int(os.environ.get("JEN1_TILE_TARGET_WGS", "256"))
        self.tile_one_round = os.environ.get("JEN1_TILE_ONE_ROUND", "0") != "0"
        # persistent deep-level kernel (DeepProgram): levels of at most deep_max_len positions
        self.use_deep = os.environ.get("JEN1_DEEP", "1") != "0"
        # to_in / the levels above the deep ones / to_out as two sample-resident launches (include/jen1_long.h; JEN1_LONG=0: one launch per layer)
        self.use_long = os.environ.get("JEN1_LONG", "1") != "0"
        self.deep_all_slots = os.environ.get("JEN1_DEEP_ALL_SLOTS", "0") != "0"
        self.deterministic = os.environ.get("JEN1_DETERMINISTIC", "0") != "0"
        # the network input's GroupNorm sums in a fixed order (jen1_pack_input_parts + jen1_gn_stats_from_parts) instead of float atomics
        self.pack_fixed_order = os.environ.get("JEN1_PACK_FIXED_ORDER", "1") != "0"
        # (a library built with -DJEN1_DEEP_CHUNKS runs a level of more than 64 positions in column chunks -- a unit computes <= 64
        # positions but stages, and normalises over, the whole batch element: JEN1_DEEP_MAX_LEN=96 then takes the 94-position level of
        # T = 1500 in, 188 phases and 30 launches per step; measured 765 against 777 steps/s -- the 17 new phases cost 9.4 us each,
        # what the launches they replace cost -- so the default build leaves the chunk decode out)
        self.deep_max_len = int(os.environ.get("JEN1_DEEP_MAX_LEN", "64"))
        # long levels as tile phases of persistent launches (jen1_deep_phase_tile): layers over at least this many positions.  Off by
        # default: measured at B = 8, T = 1500 the 26 tile phases of levels 0 - 1 take 10 - 11 us each against 9.8 us for the launches
        # they replace (726 against 776 steps/s; DESIGN.md section 4b says where the time goes) -- kept because the tile programs are
        # bit-reproducible without the extra statistics launches of Plan(deterministic=True), and as the base of further work
        self.use_tile_phases = os.environ.get("JEN1_TILE_PHASES", "0") != "0"
        self.tile_phase_min_len = int(os.environ.get("JEN1_TILE_PHASE_MIN_LEN", "200"))
        self.deep_nb_max = int(os.environ.get("JEN1_DEEP_NB_MAX", "0"))
        # units per phase the batch split aims for (0: as many batch elements per unit as fit) and the weight bytes a phase may stream
        # in total when every batch group reads the layer's weights again; measured at B = 8, T = 1500 (deep launch, us):
        # 0 -> 922, 128 / 2 MB -> 927, 128 / 8 MB -> 893, 128 / 16 MB -> 889, 192 / 4 MB -> 918, 256 / 8 MB -> 909, 256 / 32 MB -> 1001
        self.deep_unit_target = int(os.environ.get("JEN1_DEEP_UNIT_TARGET", "128"))
        self.deep_reread_cap = int(float(os.environ.get("JEN1_DEEP_REREAD_MB", "16")) * (1 << 20))
        self.plans: Dict[tuple, Plan] = {}
        self.load_params(params)

    def load_params(self, params: Dict[str, torch.Tensor]):
        self.W = Weights(self.spec, params, self.tdtype, self.device)
        self.plans.clear()
        self.kv_fixed: Dict[str, torch.Tensor] = {}
        self._project_fixed_embedding()

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def _project_fixed_embedding(self):
        """K/V of the learned fixed embedding (the unconditional CFG context, model.py:321):
        batch- and step-invariant, projected once per weight load."""
        spec, W = self.spec, self.W
        if not spec.transformers():
            return
        lib, dev = self.lib, self.device
        n, F = spec.ctx_len, spec.ctx_features
        s = self._stream()
        fx = torch.empty((n, F), dtype=self.tdtype, device=dev)
        src = W.fixed[:n].contiguous()
        L.check(lib.jen1_standardize_rows(src.data_ptr(), fx.data_ptr(), n, F, F, F, 1e-5, self.dt, s), "jen1_standardize_rows")
        groups = []
        for t in spec.transformers():
            mid2 = 2 * t.heads * t.head_features
            out = torch.empty((n, mid2), dtype=self.tdtype, device=dev)
            self.kv_fixed[t.name] = out
            groups.append((out.data_ptr(), W.v[f"{t.name}.kv2.bias"].data_ptr(), W.kvx_off[t.name], mid2, mid2))
        for part in _bgemm_parts(groups):
            tab = L.bgemm_group_table([(c, b_, n0 - part[0][2], N, ldc) for c, b_, n0, N, ldc in part], dev)
            g = L.BGemmArgs()
            n_lo, n_hi = part[0][2], part[-1][2] + part[-1][3]
            g.a, g.groups = fx.data_ptr(), tab.data_ptr()
            g.b = W.w["kv2_all"].data_ptr() + n_lo * F * W.w["kv2_all"].element_size()
            g.M, g.Ntot, g.K, g.lda, g.ldb, g.n_groups, g.dtype, g.alpha = n, n_hi - n_lo, F, F, F, len(part), self.dt, 1.0
            g.group_align = _group_align(part)
            L.check(lib.jen1_big_gemm(C.byref(g), s), "jen1_big_gemm")
        torch.cuda.synchronize(dev)

    def plan(self, B: int, T: int, nrep: int, causal: bool, slot: int = 0, n_t: Optional[int] = None,
             deep: Optional[bool] = None, deterministic: Optional[bool] = None) -> Plan:
        """``slot`` distinguishes plans of the same shape that must own separate buffers because
        they run concurrently on different streams (sub-batches of one sampler step); ``n_t`` selects
        the sampler's table mode (see Plan).  ``deep``: use the persistent deep-level launch (default: yes).
        ``deterministic``: fixed-order statistics (see Plan; default: ``self.deterministic``, env JEN1_DETERMINISTIC)."""
        if deep is None:
            # concurrent samplers (slot > 0) keep one launch per layer by default: persistent launches are safe to share the GPU
            # (tickets) but each one wants every CU, while independent launch-per-layer chains overlap -- 4 batches in flight:
            # 1 109 steps/s aggregate with four persistent launches against 1 412 with one (bench.py extra.concurrent_batches)
            deep = slot == 0 or self.deep_all_slots
        deep = bool(deep) and self.use_deep
        det = self.deterministic if deterministic is None else bool(deterministic)
        key = (B, T, nrep, bool(causal), slot, n_t, deep, det)
        if key not in self.plans:
            self.plans[key] = Plan(self, B, T, nrep, bool(causal), n_t, deep=deep, deterministic=det)
        return self.plans[key]
