"""Training path of the denoiser: forward with saved activations + backward, every heavy operator in HIP.

The reference trains ``UNetCFG1d`` through torch.autograd (trainer.py:139-150: ``scaler.scale(loss / n).backward()``,
gdm.py:245-272 ``training_loosses``).  Here every operator of the path that carries FLOPs or bytes -- the 1-D
convolutions / transposed convolutions / linears (forward, data gradient, weight gradient), GroupNorm+FiLM+SiLU,
LayerNorm, GELU/SiLU, the attention products and softmax -- is a ``torch.autograd.Function`` whose forward and
backward are kernels of libjen1_hip.so (include/jen1_train.h).  torch itself only does plumbing on the way:
concatenation / cropping / residual adds of activations, the 129 Fourier features of the timestep, the CFG
combine on the [B, 128, T] output and the loss reduction.

Activations are channel-last ``[B, L, C]`` (C padded to a multiple of 8 with zero columns), so the transformer's
``[B, N, C]`` is the native layout and no permute exists.  Parameters stay in the reference's ``state_dict`` layout
(float32 master copies, flat-buffer views when ``FusedAdamW`` owns them); compute copies ``[k][C_out][C_in]`` in the
compute dtype are re-packed once per optimiser step.  Weight gradients are accumulated by the kernels straight into
``param.grad`` (float32 atomics, reference layout): gradient accumulation over micro-batches costs nothing extra.

There is no fallback: without libjen1_hip.so / a ROCm device every entry point raises.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import math
import os
import weakref
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch.autograd import Function, Variable

from . import lib as L
from .graphs import capture as capture_graph
from .config import ResSpec, TransformerSpec, UNetSpec


def pad8(c: int) -> int:
    return (c + 7) // 8 * 8


class Map:
    """index map of ``jen1_gemm_operand`` (include/jen1_train.h)"""

    def __init__(self, axis: int, L_idx: int, L_src: int, mul: int = 1, tapmul: int = 0, shift: int = 0, div: int = 1, reflect: bool = False,
                 per_batch: bool = False):
        self.axis, self.L, self.Lsrc, self.mul, self.tapmul, self.shift, self.div, self.reflect = axis, L_idx, L_src, mul, tapmul, shift, div, reflect
        self.per_batch = per_batch        # the shift of batch element b comes from jen1_gemm_args.map_shift_b[b] (CausalRows)


def _operand(ptr: int, ld_r: int, ld_k: int, tap_stride: int = 0, zs0: int = 0, zs1: int = 0, zdiv: int = 1,
             m: Optional[Map] = None) -> L.GemmOperand:
    o = L.GemmOperand()
    o.p, o.zs0, o.zs1, o.ld_r, o.ld_k, o.tap_stride, o.zdiv = ptr, zs0, zs1, ld_r, ld_k, tap_stride, zdiv
    if m is None:
        o.map_axis, o.map_L, o.map_Lsrc, o.map_mul, o.map_tapmul, o.map_shift, o.map_div = 0, 1, 1, 1, 0, 0, 1
    else:
        o.map_axis, o.map_L, o.map_Lsrc, o.map_mul, o.map_tapmul, o.map_shift, o.map_div = m.axis, m.L, m.Lsrc, m.mul, m.tapmul, m.shift, m.div
        o.map_reflect = int(m.reflect)
        o.reserved = 1 if m.per_batch else 0
    return o


class TrainRuntime:
    """Handle on the library + the per-optimiser-step cache of packed compute weights."""

    def __init__(self, compute_dtype: str = "bf16", device="cuda"):
        self.lib = L.load()                      # raises when the HIP extension is missing
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.Jen1HipError("the training path needs a ROCm GPU (device='cuda'); no CPU path exists in this package")
        assert compute_dtype in ("f32", "bf16")
        self.dt = L.F32 if compute_dtype == "f32" else L.BF16
        self.tdtype = torch.float32 if compute_dtype == "f32" else torch.bfloat16
        self._packed: Dict[tuple, list] = {}          # (id, kind, dtype) -> [weakref, buffer, epoch of last refresh, kind]
        self.epoch = 0
        self._fresh_epoch = -1
        self._acc32: Optional[torch.Tensor] = None        # persistent float32 split-K accumulator, zero at rest
        # keep a second, transposed compute copy of every weight so that the data gradient is K-contiguous on both operands
        self.dgrad_copies = os.environ.get("JEN1_TRAIN_DGRAD_COPIES", "1") == "1"
        self.wgrad_plain_rmw = os.environ.get("JEN1_TRAIN_WGRAD_RMW", "1") == "1"
        # plain many-row linears (the text-context K/V projections) on the large-M matrix-core kernels (BigLinearFn, csrc/big_gemm.hip)
        self.big_linears = os.environ.get("JEN1_TRAIN_BIG_LINEARS", "1") == "1"
        # the ends of a pass (q_sample / cat / layout change, context rows, time features, CFG combine + loss) as own launches
        # (csrc/train_glue.hip) instead of ATen elementwise / cat / reduce kernels
        self.fused_glue = os.environ.get("JEN1_TRAIN_FUSED_GLUE", "1") == "1"
        self._consts: Dict[tuple, torch.Tensor] = {}
        self.stats: Optional[dict] = None        # {"family": [launches, flops, algorithmic bytes]} while a counting pass runs (bench.py)
        self.skinny_gemm = os.environ.get("JEN1_TRAIN_SKINNY", "1") == "1"
        self.dual_norms = os.environ.get("JEN1_TRAIN_DUAL_NORMS", "1") == "1"          # self-attention's two LayerNorms of x in one launch
        self.fused_repack = os.environ.get("JEN1_TRAIN_FUSED_REPACK", "1") == "1"      # every compute copy in one launch (jen1_repack)
        self.repack_twins = os.environ.get("JEN1_TRAIN_REPACK_TWINS", "1") == "1"      # ... a weight's two copies from one read of it
        self._repack_tab, self._repack_meta, self._repack_n = None, (0, 0), -1
        self.skinny_max_steps = int(os.environ.get("JEN1_TRAIN_SKINNY_STEPS", "64"))      # K steps per wave
        self.target_wgs = int(os.environ.get("JEN1_TRAIN_TARGET_WGS", "256"))
        self.min_steps = int(os.environ.get("JEN1_TRAIN_MIN_STEPS", "4"))      # K steps (of 32) a split keeps at least
        # weight gradients over many rows (>= big_wgrad_rows) on jen1_big_gemm_tn_conv: 23 us at 24 000 x 128 x (3 x 128) with the bias
        # gradient against 40 on train_gemm's weight-gradient form; the pass 13.58 -> 13.31 ms.  (First version: 45 us -- a tile of one tap adds
        # floats `taps` apart in lines it shares with the other taps' workgroups, and every reduction slice's bias atomics land on the same
        # four lines; the kernel now holds a chunk of every tap per tile and interleaves them through LDS, bias sums have workgroups of
        # their own.)  big_wgrad_unpair: also take the mid-size layers out of the paired launches (measured: nothing, 13.34)
        self.big_wgrads = os.environ.get("JEN1_TRAIN_BIG_WGRADS", "1") == "1"
        self.big_wgrad_unpair = os.environ.get("JEN1_TRAIN_BIG_WGRAD_UNPAIR", "0") == "1"
        # forward / data gradient of convolutions over many rows on jen1_big_gemm_conv
        self.big_convs = os.environ.get("JEN1_TRAIN_BIG_CONVS", "1") == "1"
        self.big_conv_rows = int(os.environ.get("JEN1_TRAIN_BIG_CONV_ROWS", "4096"))
        self.big_wgrad_rows = int(os.environ.get("JEN1_TRAIN_BIG_WGRAD_ROWS", "4096"))
        # weight gradients on their own stream (weight_grad below)
        # layers per fork; 0 (default): on the pass's own stream.  The fork was worth 1 ms while every weight gradient went through it; since
        # a layer's two gradients share a launch only the FiLM / many-row / library-GEMM ones are left, and the branch costs the replayed
        # graph more than it hides (14.63 ms with groups of 64, 14.23 ms inline)
        self.wgrad_group = int(os.environ.get("JEN1_TRAIN_WGRAD_GROUP", "0"))
        self._wstream: Optional[torch.cuda.Stream] = None
        self._wqueue: list = []
        self._wheld: list = []
        self._wforked = False
        self._wjoin: list = []                   # streams the weight-gradient stream was forked from in the running backward pass
        # every block's FiLM projection as one GEMM (FilmBankFn)
        self.film_bank = os.environ.get("JEN1_TRAIN_FILM_BANK", "1") == "1"
        # a layer's weight gradient and data gradient in one launch (jen1_train_gemm_pair)
        self.pair_grads = os.environ.get("JEN1_TRAIN_PAIR_GRADS", "1") == "1"
        # short sequences: the whole attention core in one launch each way (jen1_attn_small_forward / _backward)
        self.small_attn = os.environ.get("JEN1_TRAIN_SMALL_ATTN", "1") == "1"
        # a tensor that feeds a norm / linear AND a branch around it: forked, its gradients merge inside the layer's backward kernel
        self.fork_norms = os.environ.get("JEN1_TRAIN_FORK", "1") == "1"
        # ... and a skip connection of the U-Net / the text context of the 13 cross-attentions: handed on as aliases by their consumers
        self.fork_skips = os.environ.get("JEN1_TRAIN_FORK_SKIPS", "1") == "1"
        # the unconditional half of the CFG pair reads ONE shared set of context rows (the fixed embedding) instead of B copies
        self.share_fixed_context = os.environ.get("JEN1_TRAIN_SHARE_FIXED", "1") == "1"
        self._banks: Dict[tuple, list] = {}      # (ids of the weights, dtype) -> [weakrefs, weight matrix, weakrefs of the biases, bias vector, epoch]
        # the text-context K / V projections of all cross-attention layers as ONE product per pass (KvBank, ContextKVFn; csrc/train_kvbank.hip)
        self.kv_grouped = os.environ.get("JEN1_TRAIN_KV_GROUPED", "1") == "1"
        self._kv_banks: Dict[tuple, "KvBank"] = {}

    def kv_rows(self, B: int) -> torch.Tensor:
        """[0 .. B - 1, B, B, ... B] (int32, 2 B entries): the K / V row every element of a CFG pair reads when the unconditional
        half shares one set of context rows"""
        t = self._consts.get(("kv_rows", B))
        if t is None:
            t = self._consts[("kv_rows", B)] = torch.cat([torch.arange(B, dtype=torch.int32), torch.full((B,), B, dtype=torch.int32)]).to(self.device)
        return t

    def const(self, n: int, v: float) -> torch.Tensor:
        """a cached float32 vector of n copies of v (read-only operands of the glue kernels)"""
        t = self._consts.get((n, v))
        if t is None:
            t = self._consts[(n, v)] = torch.full((n,), float(v), dtype=torch.float32, device=self.device)
        return t

    # ------------------------------------------------------------------ plumbing
    def stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    def invalidate(self) -> None:
        """the parameters changed (optimiser step / load): every packed compute copy is stale"""
        self.epoch += 1

    def weight_grad(self, launch, *keep: torch.Tensor) -> None:
        """Nothing in the backward pass reads a weight gradient, but every launch costs the chain ~4.5 us + its run time, and one third
        of the pass's GEMM launches are weight gradients.  Inside a backward node: ``launch()`` (the weight / bias gradient launches of
        one layer) is put off; every ``wgrad_group`` layers the queue is issued on a second stream forked from the current one -- a
        parallel branch of the captured graph, in eager mode a second hardware queue -- and the stream the pass runs on joins that
        stream when the autograd engine finishes (a final callback), so whatever reads ``.grad`` after ``backward()`` is ordered behind
        it as before.  The queue keeps its order (two uses of one weight accumulate into the same buffer).  One fork per GROUP, not per
        layer: a cross-stream edge of a replayed graph costs more than the launch it hides (a fork per layer: 22.0 -> 25.0 ms).
        ``keep``: the operands, held until the join so that the allocator cannot hand their memory to a later launch of the main stream
        while the second stream still reads them."""
        if self.wgrad_group <= 0:
            launch()
            return
        if not self._wjoin:
            Variable._execution_engine.queue_callback(self.join_weight_grads)
        cur = torch.cuda.current_stream(self.device)
        if cur not in self._wjoin:
            self._wjoin.append(cur)
        self._wqueue.append(launch)
        self._wheld.extend(keep)
        if len(self._wqueue) >= self.wgrad_group:
            self.flush_weight_grads()

    def flush_weight_grads(self, frm: Optional[torch.cuda.Stream] = None) -> Optional[torch.cuda.Stream]:
        """issue the queued weight-gradient launches; returns their stream if the running backward pass has used it (a collective over
        gradients waits on it too)"""
        if self._wqueue:
            if self._wstream is None:
                self._wstream = torch.cuda.Stream(self.device)
            self._wstream.wait_stream(torch.cuda.current_stream(self.device) if frm is None else frm)
            with torch.cuda.stream(self._wstream):
                for launch in self._wqueue:
                    launch()
            self._wqueue.clear()
            self._wforked = True
        return self._wstream if self._wforked else None

    def abandon_weight_grads(self) -> None:
        """called when a new forward pass starts: anything still queued belongs to a backward pass that raised before its final
        callback ran (the callback is registered once per pass, so stale state would keep the next pass from registering its own)"""
        if self._wjoin or self._wqueue or self._wheld:
            if self._wforked and self._wstream is not None:
                torch.cuda.current_stream(self.device).wait_stream(self._wstream)
            self._wqueue.clear()
            self._wjoin.clear()
            self._wheld.clear()
            self._wforked = False

    def join_weight_grads(self) -> None:
        if not self._wjoin:
            return
        self.flush_weight_grads(self._wjoin[-1])         # (the stream the backward nodes ran on: the callback runs on the caller's thread)
        cur = torch.cuda.current_stream(self.device)
        for s in [cur] + [s for s in self._wjoin if s != cur]:
            s.wait_stream(self._wstream)
        self._wjoin.clear()
        self._wheld.clear()
        self._wforked = False

    def dt_of(self, t: torch.Tensor) -> int:
        if t.dtype == torch.float32:
            return L.F32
        if t.dtype == torch.bfloat16:
            return L.BF16
        raise L.Jen1HipError(f"unsupported activation dtype {t.dtype}")

    @staticmethod
    def _layout(w: torch.Tensor, kind: str) -> torch.Tensor:
        d = w.detach()
        if kind == "linear":
            return d.unsqueeze(0)                          # [1][Co][Ci]
        if kind == "conv":
            return d.permute(2, 0, 1)                      # [k][Co][Ci]
        if kind == "convT":
            return d.permute(2, 1, 0)                      # ConvTranspose1d [Ci][Co][k] -> [k][Co][Ci]
        # the data-gradient copies: rows = input channels, K = output channels contiguous (register-direct GEMM path)
        if kind == "linearD":
            return d.t().unsqueeze(0)                      # [1][Ci][Co]
        if kind == "convD":
            return d.permute(2, 1, 0)                      # [Co][Ci][k] -> [k][Ci][Co]
        assert kind == "convTD"
        return d.permute(2, 0, 1)                          # [Ci][Co][k] -> [k][Ci][Co]

    def packed(self, w: torch.Tensor, kind: str, dtype: torch.dtype) -> torch.Tensor:
        """compute copy [k][C_out][pad8(C_in)] of a Conv1d [Co, Ci, k] / ConvTranspose1d [Ci, Co, k] / Linear [Co, Ci]
        weight.  The buffer is allocated once and refreshed IN PLACE when the parameters moved, so a captured graph
        keeps reading the same address (GraphedLossStep refreshes outside the graph, once per optimiser step)."""
        key = (id(w), kind, dtype)
        hit = self._packed.get(key)
        if hit is not None and hit[0]() is not w:          # id() of a dead parameter can be reused
            hit = None
        if hit is None:
            with torch.no_grad():
                d = self._layout(w, kind)
                k, co, ci = d.shape
                if kind == "linear" and dtype == d.dtype and ci % 8 == 0:
                    hit = [weakref.ref(w), d, -1, kind]    # float32 mode: the parameter itself is the compute copy
                else:
                    hit = [weakref.ref(w), torch.zeros((k, co, pad8(ci)), dtype=dtype, device=w.device), -2, kind]
            self._packed[key] = hit
        if hit[2] != self.epoch and hit[2] != -1:
            self._refresh(hit, w)
        return hit[1]

    def packed_bank(self, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor], dtype: torch.dtype) -> Tuple[torch.Tensor, torch.Tensor]:
        """Compute copies of several Linear weights of one input width as ONE matrix [1][sum C_out][pad8(C_in)] (+ their biases as one
        float32 vector): each weight's copy is a row slice of it, registered like any other packed copy (refreshed by the same
        jen1_repack launch).  The C_out must be multiples of 8 (slices start on 16-byte boundaries in either dtype)."""
        key = (tuple(id(w) for w in weights), dtype)
        hit = self._banks.get(key)
        if hit is not None and any(r() is not w for r, w in zip(hit[0], weights)):
            hit = None
        if hit is None:
            ci = weights[0].shape[1]
            assert all(w.shape[1] == ci and w.shape[0] % 8 == 0 and w.dtype == torch.float32 for w in weights)
            assert all(b.shape[0] == w.shape[0] for w, b in zip(weights, biases))
            total = sum(w.shape[0] for w in weights)
            dev = weights[0].device
            mat = torch.zeros((1, total, pad8(ci)), dtype=dtype, device=dev)
            off = 0
            for w in weights:
                self._packed[(id(w), "linear", dtype)] = [weakref.ref(w), mat[:, off:off + w.shape[0], :], -2, "linear"]
                off += w.shape[0]
            self._repack_tab = None                # (entries may have been replaced: rebuild the table of jen1_repack)
            hit = [[weakref.ref(w) for w in weights], mat, [weakref.ref(b) for b in biases],
                   torch.zeros(total, dtype=torch.float32, device=dev), -2]
            self._banks[key] = hit
        for w in weights:
            self.packed(w, "linear", dtype)        # (refreshes a stale slice; after refresh_all none is)
        self._refresh_bank_bias(hit)
        return hit[1], hit[3]

    def kv_bank(self, weights: Sequence[torch.Tensor], gammas: Sequence[torch.Tensor], betas: Sequence[torch.Tensor]) -> "KvBank":
        """the folded operands of the stacked text-context projection (KvBank), cached on the identity of the parameters and kept up to
        date like the other compute copies (one jen1_kv_fold launch per optimiser step)"""
        key = tuple(id(w) for w in weights)
        hit = self._kv_banks.get(key)
        if hit is not None and not hit.same(weights, gammas, betas):
            hit = None
        if hit is None:
            hit = self._kv_banks[key] = KvBank(self, weights, gammas, betas)
        hit.refresh()
        return hit

    def _refresh_bank_bias(self, hit) -> None:
        if hit[4] != self.epoch:
            bs = [r() for r in hit[2]]
            if all(b is not None for b in bs):
                with torch.no_grad():
                    torch.cat([b.detach() for b in bs], out=hit[3])
            hit[4] = self.epoch

    def _refresh(self, hit, w) -> None:
        with torch.no_grad():
            d = self._layout(w, hit[3])
            hit[1][:, :, :d.shape[2]].copy_(d)
        hit[2] = self.epoch

    def refresh_all(self) -> None:
        """bring every packed copy up to date now (outside any graph capture)"""
        if self._fresh_epoch == self.epoch:
            return
        for bank in self._banks.values():
            self._refresh_bank_bias(bank)
        # (banks whose parameters were freed or replaced are dropped: their weak references are dead)
        self._kv_banks = {k: kvb for k, kvb in self._kv_banks.items() if kvb.alive()}
        for kvb in self._kv_banks.values():
            kvb.refresh()
        live = [(hit, hit[0]()) for hit in self._packed.values() if hit[2] != -1]
        live = [(hit, w) for hit, w in live if w is not None]
        if self.fused_repack and live:
            # ONE launch for every copy (jen1_repack): the table of (source view, destination) is rebuilt when a copy was added
            if self._repack_tab is None or self._repack_n != len(self._packed):
                self._repack_tab = []              # one table per destination dtype (the time MLPs keep float32 copies in bf16 mode)
                for dt_t, dt_c in ((torch.float32, L.F32), (torch.bfloat16, L.BF16)):
                    ents, t0 = [], 0
                    # a weight's forward copy [k][Co][Ci] and its data-gradient transpose [k][Ci][Co] come from ONE read (dst2)
                    twins = {(id(w), hit[3][:-1]): hit for hit, w in live if hit[1].dtype == dt_t and hit[3].endswith("D")} if self.repack_twins else {}
                    fwd = {(id(w), hit[3]) for hit, w in live if hit[1].dtype == dt_t and not hit[3].endswith("D")}
                    for hit, w in live:
                        if hit[1].dtype != dt_t or (hit[3].endswith("D") and twins and (id(w), hit[3][:-1]) in fwd):
                            continue                   # (a transpose whose forward copy is in this table rides on that entry)
                        d = self._layout(w, hit[3])
                        assert w.dtype == torch.float32
                        e = L.RepackEntry()
                        e.src, e.dst = d.data_ptr(), hit[1].data_ptr()
                        e.d0, e.d1, e.d2, e.ld = d.shape[0], d.shape[1], d.shape[2], hit[1].shape[2]
                        e.s0, e.s1, e.s2 = d.stride(0), d.stride(1), d.stride(2)
                        twin = twins.get((id(w), hit[3]))
                        if twin is not None and twin[0]() is w:
                            assert twin[1].shape[0] == d.shape[0] and twin[1].shape[1] == d.shape[2] and twin[1].shape[2] >= d.shape[1]
                            e.dst2, e.ld2 = twin[1].data_ptr(), twin[1].shape[2]
                        e.tile0 = t0
                        t0 += d.shape[0] * ((d.shape[1] + 31) // 32) * ((d.shape[2] + 31) // 32)
                        ents.append(e)
                    if ents:
                        arr = (L.RepackEntry * len(ents))(*ents)
                        self._repack_tab.append((torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device), len(ents), t0, dt_c))
                self._repack_n = len(self._packed)
            for tab, n, tiles, dt_c in self._repack_tab:
                L.check(self.lib.jen1_repack(tab.data_ptr(), n, tiles, dt_c, self.stream()), "jen1_repack")
            for hit, _ in live:
                hit[2] = self.epoch
        else:
            for hit, w in live:
                if hit[2] != self.epoch:
                    self._refresh(hit, w)
        self._fresh_epoch = self.epoch

    @staticmethod
    def grad_of(p: torch.Tensor) -> torch.Tensor:
        """the float32 gradient buffer of a parameter (created zeroed on first use; a flat-buffer view under FusedAdamW)"""
        if p.grad is None:
            p.grad = torch.zeros_like(p, dtype=torch.float32)
        assert p.grad.dtype == torch.float32 and p.grad.is_contiguous()
        return p.grad

    # ------------------------------------------------------------------ the GEMM
    def gemm(self, a: L.GemmOperand, b: L.GemmOperand, c_ptr: int, M: int, N: int, K: int, *, dtype: int, taps: int = 1,
             batches: int = 1, taps_in_z: bool = False, ldc_m: int, ldc_n: int = 1, c_tap_stride: int = 0, c_zs0: int = 0,
             c_zs1: int = 0, c_zdiv: int = 1, bias: Optional[torch.Tensor] = None, splitk: int = 1, atomic: bool = False,
             accumulate: bool = False, c_f32: bool = False, alpha: float = 1.0, rowsum: Optional[torch.Tensor] = None,
             residual: Optional[torch.Tensor] = None, skinny: bool = False, defer: bool = False, pair_with=None,
             shift_b: Optional[torch.Tensor] = None):
        """``defer``: no launch, the filled argument block comes back; ``pair_with`` (such a block): both products in ONE launch
        (jen1_train_gemm_pair: the deferred one first)"""
        g = L.GemmArgs()
        g.a, g.b, g.c, g.bias = a, b, c_ptr, (None if bias is None else bias.data_ptr())
        g.c_zs0, g.c_zs1, g.ldc_m, g.ldc_n, g.c_tap_stride, g.c_zdiv = c_zs0, c_zs1, ldc_m, ldc_n, c_tap_stride, c_zdiv
        g.M, g.N, g.K, g.taps, g.batches = M, N, K, taps, batches
        g.taps_in_z, g.splitk, g.atomic, g.accumulate, g.c_f32, g.dtype = int(taps_in_z), splitk, int(atomic), int(accumulate), int(c_f32), dtype
        g.alpha = alpha
        g.rowsum = None if rowsum is None else rowsum.data_ptr()
        g.residual = None if residual is None else residual.data_ptr()
        g.reserved = 1 if skinny else 0
        g.map_shift_b = None if shift_b is None else shift_b.data_ptr()
        if self.stats is not None:
            # executed work of the launch (bench.py's training roofline counts the pass's OWN launch list): 2 M N K per tap and batch;
            # algorithmic bytes = the two operands once + the result (float32 read-modify-write when it accumulates)
            es = 4 if dtype == L.F32 else 2
            fl = 2.0 * M * N * K * taps * batches
            by = (M * K * taps + N * K * taps) * batches * es + M * N * batches * (taps if taps_in_z else 1) * ((8 if (atomic or accumulate) else 4) if c_f32 else es)
            e = self.stats.setdefault("train_gemm", [0, 0.0, 0.0])
            e[0] += 0 if pair_with is not None else 1
            e[1] += fl
            e[2] += by
        if defer:
            return g
        if pair_with is not None:
            L.check(self.lib.jen1_train_gemm_pair(pair_with, g, self.stream()), "jen1_train_gemm_pair")
        else:
            L.check(self.lib.jen1_train_gemm(g, self.stream()), "jen1_train_gemm")
        return None

    def split_accumulator(self, n: int) -> torch.Tensor:
        """the persistent zeroed float32 scratch every split-K forward / data-gradient GEMM of the stream accumulates
        into; ``hand_over`` converts it into the output and zeroes it again in the same launch"""
        if self._acc32 is None or self._acc32.numel() < n:
            assert not torch.cuda.is_current_stream_capturing(), "the split-K accumulator must be sized before graph capture"
            self._acc32 = torch.zeros(max(n, 1 << 23), dtype=torch.float32, device=self.device)
        return self._acc32

    def hand_over(self, acc: torch.Tensor, out: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        n = out.numel()
        if residual is not None:
            L.check(self.lib.jen1_convert_clear_add(acc.data_ptr(), out.data_ptr(), residual.data_ptr(), n, self.dt_of(out), self.stream()),
                    "jen1_convert_clear_add")
            return out
        L.check(self.lib.jen1_convert_clear(acc.data_ptr(), out.data_ptr(), n, self.dt_of(out), self.stream()), "jen1_convert_clear")
        return out

    def want_skinny(self, M: int, N: int, ksteps: int) -> bool:
        """few rows against a big weight: 32 x 16 tiles with the K split inside the workgroup (train_gemm_skinny_kernel) instead of a
        split-K launch + conversion; as long as that gives the chip enough workgroups of a reasonable length"""
        if not self.skinny_gemm:
            return False
        tiles = ((M + 63) // 64) * ((N + 63) // 64)
        wgs = ((M + 31) // 32) * ((N + 15) // 16)
        return tiles < self.target_wgs // 2 and wgs <= 4096 and ksteps <= 4 * self.skinny_max_steps

    def count(self, family: str, flops: float, nbytes: float) -> None:
        """one launch of ``family`` in the counting pass of bench.py"""
        if self.stats is not None:
            e = self.stats.setdefault(family, [0, 0.0, 0.0])
            e[0] += 1; e[1] += flops; e[2] += nbytes

    def pick_splitk(self, M: int, N: int, ksteps: int, z: int = 1) -> int:
        tiles = ((M + 63) // 64) * ((N + 63) // 64) * z
        if tiles >= self.target_wgs // 2:
            return 1
        # (a few rows against a very long K -- the data gradient of the stacked FiLM projections, 16 x 512 x 55 296: the output is tiny, the
        # atomics cost nothing, and 256 workgroups of 54 serial steps took 82 us)
        wgs = self.target_wgs * (4 if (M <= 64 and ksteps >= 1024) else 1)
        s = min(max(1, wgs // tiles), max(1, ksteps // self.min_steps))
        return max(1, min(s, 65535 // max(1, z)))


# =====================================================================================================================
# convolution family: _Conv1d (blocks.py:34-53), nn.Conv1d / nn.ConvTranspose1d of Upsample1d (blocks.py:69-95), nn.Linear
# =====================================================================================================================
class ConvGeom:
    """static description of one convolution call"""

    def __init__(self, kind: str, taps: int, stride: int, pad: int, L_in: int, L_out: int, ci: int, co: int, reflect: bool = False,
                 pad_b: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        self.kind, self.taps, self.stride, self.pad, self.L_in, self.L_out, self.ci, self.co = kind, taps, stride, pad, L_in, L_out, ci, co
        self.reflect = reflect        # forward only: F.pad(mode="reflect") instead of zeros (SEANet convolutions)
        # (-pad, +pad) per batch element as int32 tensors: a pass that mixes causal and non-causal clips ("conv" kind only)
        self.pad_b = pad_b
        assert pad_b is None or kind == "conv"

    def fwd_map(self, axis: int) -> Optional[Map]:
        """activation index (b, t_out) [+ tap] -> input row"""
        if self.kind == "linear":
            return None
        if self.kind == "conv":
            return Map(axis, self.L_out, self.L_in, mul=self.stride, tapmul=1, shift=-self.pad, reflect=self.reflect,
                       per_batch=self.pad_b is not None)
        return Map(axis, self.L_out, self.L_in, mul=1, tapmul=-1, shift=self.pad, div=self.stride)

    def bwd_map(self, axis: int) -> Optional[Map]:
        """activation index (b, t_in) [+ tap] -> output row"""
        if self.kind == "linear":
            return None
        if self.kind == "conv":
            return Map(axis, self.L_in, self.L_out, mul=1, tapmul=-1, shift=self.pad, div=self.stride, per_batch=self.pad_b is not None)
        return Map(axis, self.L_in, self.L_out, mul=self.stride, tapmul=1, shift=-self.pad)

    @property
    def fwd_shift_b(self) -> Optional[torch.Tensor]:
        return None if self.pad_b is None else self.pad_b[0]

    @property
    def bwd_shift_b(self) -> Optional[torch.Tensor]:
        return None if self.pad_b is None else self.pad_b[1]


def _conv_forward(rt: TrainRuntime, x: torch.Tensor, wp: torch.Tensor, bias: Optional[torch.Tensor], g: ConvGeom,
                  residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x: [..rows.., ldx] channel-last with ldx == wp.shape[2]; returns [B, L_out, pad8(co)] (+ ``residual`` of that shape, added in
    the epilogue of the GEMM or of the split-K hand-over)"""
    dt = rt.dt_of(x)
    ldx = x.shape[-1]
    k, co, cip = wp.shape
    assert ldx == cip and x.is_contiguous() and k == g.taps and co == g.co, (x.shape, wp.shape, g.__dict__)
    rows_in = x.numel() // ldx
    B = rows_in // g.L_in
    M = B * g.L_out
    ldy = pad8(co)
    a = _operand(x.data_ptr(), ldx, 1, m=g.fwd_map(1))
    b = _operand(wp.data_ptr(), cip, 1, tap_stride=co * cip)
    ksteps = k * ((cip + 31) // 32)
    if residual is not None:
        assert residual.is_contiguous() and residual.numel() == B * g.L_out * ldy and residual.dtype == x.dtype, (residual.shape, B, g.L_out, ldy)
    if (rt.big_convs and dt == L.BF16 and g.kind in ("conv", "linear") and M >= rt.big_conv_rows and not g.reflect
            and cip % 8 == 0 and co % 4 == 0):
        # many rows (the long levels): the 128 x 128 matrix-core kernel with the taps in its row map (jen1_big_gemm_conv: 10 us at
        # 24 000 x 128 x (3 x 128) against 25 on the register-direct form)
        y = (torch.zeros if ldy != co else torch.empty)((B, g.L_out, ldy), dtype=x.dtype, device=x.device)
        conv = g.kind == "conv"
        L.check(rt.lib.jen1_big_gemm_conv(x.data_ptr(), wp.data_ptr(), None if bias is None else bias.data_ptr(),
                                          None if residual is None else residual.data_ptr(), y.data_ptr(), B if conv else M, g.L_in if conv else 1,
                                          g.L_out if conv else 1, cip, co, k, g.stride if conv else 1, (0 if g.pad_b is not None else g.pad) if conv else 0,
                                          0, ldx, cip, co * cip, ldy, None if g.pad_b is None else g.fwd_shift_b.data_ptr(), 1, rt.stream()), "jen1_big_gemm_conv")
        rt.count("big_gemm", 2.0 * M * co * cip * k, 2.0 * (rows_in * ldx + k * co * cip + M * ldy))
        return y
    if (rt.big_convs and dt == L.BF16 and g.kind == "convT" and M >= rt.big_conv_rows and g.ci % 64 == 0 and cip == g.ci and co % 4 == 0):
        # ConvTranspose1d: output position t reads input (t + padding - tap) / stride where that is whole -- the conv form with the taps
        # reversed and the division in its row map (the other taps' rows arrive as zeros: no traffic, wasted MFMA steps only)
        y = (torch.zeros if ldy != co else torch.empty)((B, g.L_out, ldy), dtype=x.dtype, device=x.device)
        L.check(rt.lib.jen1_big_gemm_conv(x.data_ptr(), wp.data_ptr(), None if bias is None else bias.data_ptr(),
                                          None if residual is None else residual.data_ptr(), y.data_ptr(), B, g.L_in, g.L_out, g.ci, co, k, 1,
                                          k - 1 - g.pad, 1, ldx, cip, co * cip, ldy, None, g.stride, rt.stream()), "jen1_big_gemm_conv")
        rt.count("big_gemm", 2.0 * M * co * cip * k / g.stride, 2.0 * (rows_in * ldx + k * co * cip + M * ldy))
        return y
    skinny = rt.want_skinny(M, co, ksteps)
    sk = 1 if skinny else rt.pick_splitk(M, co, ksteps)
    alloc = torch.zeros if (ldy != co or sk > 1) else torch.empty
    if sk > 1:
        acc = rt.split_accumulator(B * g.L_out * ldy)
        rt.gemm(a, b, acc.data_ptr(), M, co, cip, dtype=dt, taps=k, ldc_m=ldy, bias=bias, splitk=sk, atomic=True, c_f32=True, shift_b=g.fwd_shift_b)
        return rt.hand_over(acc, torch.empty((B, g.L_out, ldy), dtype=x.dtype, device=x.device), residual)
    if residual is not None and ldy != co:
        y = residual.clone()                   # (padding columns: keep the residual's)
        rt.gemm(a, b, y.data_ptr(), M, co, cip, dtype=dt, taps=k, ldc_m=ldy, bias=bias, accumulate=True, skinny=skinny, shift_b=g.fwd_shift_b)
        return y
    y = alloc((B, g.L_out, ldy), dtype=x.dtype, device=x.device)
    rt.gemm(a, b, y.data_ptr(), M, co, cip, dtype=dt, taps=k, ldc_m=ldy, bias=bias, residual=residual, skinny=skinny, shift_b=g.fwd_shift_b)
    return y


def _conv_dgrad(rt: TrainRuntime, dy: torch.Tensor, wp: torch.Tensor, g: ConvGeom, wd: Optional[torch.Tensor] = None, pair_with=None,
                residual: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``wd``: the data-gradient copy [k][Ci][pad8(Co)] of the weight (both GEMM operands K-contiguous: the register-direct
    path of jen1_train_gemm); without it the forward copy ``wp`` is read transposed through LDS.  ``pair_with``: the deferred
    weight-gradient product of the same layer, launched together with this one.  ``residual`` (dx's shape): added in the epilogue
    (the gradient that reached the layer's input along a branch around it: ConvFn's ``fork``)"""
    dt = rt.dt_of(dy)
    ldy = dy.shape[-1]
    k, co, cip = wp.shape
    assert dy.is_contiguous() and ldy == pad8(co)
    B = dy.numel() // ldy // g.L_out
    M = B * g.L_in
    a = _operand(dy.data_ptr(), ldy, 1, m=g.bwd_map(1))
    if wd is not None:
        assert wd.shape[0] == k and wd.shape[2] == ldy
        ci_rows = wd.shape[1]
        b = _operand(wd.data_ptr(), ldy, 1, tap_stride=ci_rows * ldy)
        co = ldy                                               # K runs over the padded (zero) output channels too
        cip_n = ci_rows
    else:
        b = _operand(wp.data_ptr(), 1, cip, tap_stride=co * cip)
        cip_n = cip
    if (rt.big_convs and wd is not None and pair_with is None and dt == L.BF16 and M >= rt.big_conv_rows and ldy % 64 == 0
            and g.kind in ("linear", "conv") and cip_n % 4 == 0):
        # the data gradient over many rows: the same kernel on dy, taps reversed, pad' = taps - 1 - pad (a strided convolution: the
        # stride as the divisor of the row map)
        dx = (torch.zeros if cip_n != cip else torch.empty)((B, g.L_in, cip), dtype=dy.dtype, device=dy.device)
        conv = g.kind == "conv"
        L.check(rt.lib.jen1_big_gemm_conv(dy.data_ptr(), wd.data_ptr(), None, None if residual is None else residual.data_ptr(), dx.data_ptr(),
                                          B if conv else M, g.L_out if conv else 1, g.L_in if conv else 1, ldy, cip_n, k, 1,
                                          ((k - 1) if g.pad_b is not None else (k - 1 - g.pad)) if conv else 0, 1, ldy, ldy, cip_n * ldy, cip,
                                          None if g.pad_b is None else g.bwd_shift_b.data_ptr(), g.stride if conv else 1, rt.stream()), "jen1_big_gemm_conv")
        rt.count("big_gemm", 2.0 * M * cip_n * ldy * k, 2.0 * (B * g.L_out * ldy + k * cip_n * ldy + M * cip))
        return dx
    if (rt.big_convs and wd is not None and pair_with is None and dt == L.BF16 and M >= rt.big_conv_rows // 4 and g.kind == "convT" and ldy % 64 == 0
            and cip_n % 4 == 0):
        # ConvTranspose1d's data gradient IS a strided convolution of dY (row t_in stride + tap - padding) with the [k][Ci][Co] copy
        dx = (torch.zeros if cip_n != cip else torch.empty)((B, g.L_in, cip), dtype=dy.dtype, device=dy.device)
        L.check(rt.lib.jen1_big_gemm_conv(dy.data_ptr(), wd.data_ptr(), None, None if residual is None else residual.data_ptr(), dx.data_ptr(),
                                          B, g.L_out, g.L_in, ldy, cip_n, k, g.stride, g.pad, 0, ldy, ldy, cip_n * ldy, cip, None, 1, rt.stream()),
                "jen1_big_gemm_conv")
        rt.count("big_gemm", 2.0 * M * cip_n * ldy * k, 2.0 * (B * g.L_out * ldy + k * cip_n * ldy + M * cip))
        return dx
    ksteps = k * ((co + 31) // 32)
    skinny = wd is not None and rt.want_skinny(M, cip_n, ksteps)
    sk = 1 if skinny else rt.pick_splitk(M, cip, ksteps)
    if sk > 1:
        acc = rt.split_accumulator(B * g.L_in * cip)
        rt.gemm(a, b, acc.data_ptr(), M, cip_n, co, dtype=dt, taps=k, ldc_m=cip, splitk=sk, atomic=True, c_f32=True, pair_with=pair_with, shift_b=g.bwd_shift_b)
        return rt.hand_over(acc, torch.empty((B, g.L_in, cip), dtype=dy.dtype, device=dy.device), residual)
    if residual is not None and cip_n != cip:
        dx = residual.clone()                  # (padding columns: keep the residual's)
        rt.gemm(a, b, dx.data_ptr(), M, cip_n, co, dtype=dt, taps=k, ldc_m=cip, skinny=skinny, pair_with=pair_with, accumulate=True, shift_b=g.bwd_shift_b)
        return dx
    dx = (torch.zeros if cip_n != cip else torch.empty)((B, g.L_in, cip), dtype=dy.dtype, device=dy.device)
    rt.gemm(a, b, dx.data_ptr(), M, cip_n, co, dtype=dt, taps=k, ldc_m=cip, skinny=skinny, pair_with=pair_with, residual=residual, shift_b=g.bwd_shift_b)
    return dx


def _conv_wgrad(rt: TrainRuntime, x: torch.Tensor, dy: torch.Tensor, gw: torch.Tensor, g: ConvGeom, gb: Optional[torch.Tensor] = None,
                defer: bool = False):
    """gw (float32, reference layout) += the weight gradient.  ``gb`` (bias gradient) is accumulated by the same launch
    when dy is the row operand (Conv1d / Linear); returns whether it was -- with ``defer`` (whether, the argument block of the
    launch that has NOT been issued)."""
    dt = rt.dt_of(x)
    ldx, ldy = x.shape[-1], dy.shape[-1]
    rows_dy = dy.numel() // ldy
    if not dy.is_contiguous():                 # a column slice of a wider matrix (FilmBankFn): rows dy.stride(0) apart
        assert dy.dim() == 2 and dy.stride(1) == 1 and g.kind == "linear"
        ldy = dy.stride(0)
    k = g.taps
    if g.kind == "convT":
        # W[ci][co][k]: rows m = ci from x (K = (b, t_in)), rows n = co from dy at the mapped row
        K = x.numel() // ldx
        a = _operand(x.data_ptr(), 1, ldx)
        b = _operand(dy.data_ptr(), 1, ldy, m=g.bwd_map(2))
        M, N = g.ci, g.co
    else:
        # W[co][ci][k]: rows m = co from dy (K = (b, t_out)), rows n = ci from x at the mapped row
        K = rows_dy
        a = _operand(dy.data_ptr(), 1, ldy)
        b = _operand(x.data_ptr(), 1, ldx, m=g.fwd_map(2))
        M, N = g.co, g.ci
    if (rt.big_wgrads and not defer and g.kind in ("conv", "linear") and dt == L.BF16 and K >= rt.big_wgrad_rows
            and not g.reflect and x.is_contiguous() and dy.is_contiguous() and g.co % 8 == 0 and K % g.L_out == 0
            and (g.co % 128 == 0 or ldy >= -(-g.co // 128) * 128)
            and (x.numel() // ldx) * g.L_out == K * g.L_in and gw.is_contiguous()):
        # many reduction rows against a small output (the long levels): the transposing matrix-core kernel with the tap shift in its
        # row map and the bias gradient as one more MFMA (switch: TrainRuntime.big_wgrads, off by default)
        L.check(rt.lib.jen1_big_gemm_tn_conv(dy.data_ptr(), x.data_ptr(), gw.data_ptr(), None if gb is None else gb.data_ptr(), K // g.L_out,
                                             g.L_out, g.L_in, g.co, g.ci, k, g.stride if g.kind == "conv" else 1,
                                             (0 if g.pad_b is not None else g.pad) if g.kind == "conv" else 0, ldy, ldx, 1.0,
                                             None if g.pad_b is None else g.fwd_shift_b.data_ptr(), rt.stream()), "jen1_big_gemm_tn_conv")
        if rt.stats is not None:
            e = rt.stats.setdefault("big_gemm", [0, 0.0, 0.0])
            e[0] += 1; e[1] += 2.0 * K * g.co * g.ci * k; e[2] += 2.0 * K * (g.co + g.ci) + 8.0 * g.co * g.ci * k
        return gb is not None
    if (rt.big_wgrads and not defer and g.kind == "convT" and dt == L.BF16 and K >= rt.big_wgrad_rows // 4 and x.is_contiguous() and dy.is_contiguous()
            and g.co % 8 == 0 and g.ci % 8 == 0 and k <= 16 and K % g.L_in == 0 and rows_dy * g.L_in == K * g.L_out and gw.is_contiguous()
            and (g.ci % 128 == 0 or ldx >= -(-g.ci // 128) * 128)):
        # ConvTranspose1d W[ci][co][k]: dW[ci][co][tap] = sum x[b, t][ci] dy[b, t stride + tap - padding][co]: the same kernel with the input
        # as its row operand and dY as the shifted one (82 us on train_gemm's weight-gradient form at 6 000 x 128 x (8 x 128))
        L.check(rt.lib.jen1_big_gemm_tn_conv(x.data_ptr(), dy.data_ptr(), gw.data_ptr(), None, K // g.L_in, g.L_in, g.L_out, g.ci, g.co, k, g.stride,
                                             g.pad, ldx, ldy, 1.0, None, rt.stream()), "jen1_big_gemm_tn_conv")
        rt.count("big_gemm", 2.0 * K * g.co * g.ci * k, 2.0 * (K * g.ci + rows_dy * g.co) + 8.0 * g.co * g.ci * k)
        return (False, None) if defer else False
    sk = rt.pick_splitk(M, N, (K + 31) // 32, z=k)
    fused_bias = gb is not None and g.kind != "convT"
    # one K slice: every (tap, tile) of the gradient belongs to exactly one workgroup of this launch and launches are
    # stream-ordered, so a plain read-modify-write accumulates; float atomics only when the K range is split
    na = sk == 1 and rt.wgrad_plain_rmw
    blk = rt.gemm(a, b, gw.data_ptr(), M, N, K, dtype=dt, taps=k, taps_in_z=True, ldc_m=N * k, ldc_n=k, c_tap_stride=1,
                  splitk=sk, atomic=not na, accumulate=na, c_f32=True, rowsum=gb if fused_bias else None, defer=defer,
                  shift_b=g.fwd_shift_b if g.kind == "conv" else None)
    return (fused_bias, blk) if defer else fused_bias


class ConvFn(Function):
    """y = conv(x, weight) + bias; backward writes the parameter gradients into ``.grad`` itself"""

    @staticmethod
    def forward(ctx, x, weight, bias, rt: TrainRuntime, g: ConvGeom, residual=None, fork: bool = False):
        """``fork``: also return ``x`` (an alias) for a branch around the layer (the feed-forward's residual, blocks.py:488); its
        gradient comes back as ``backward``'s second argument and is added in the epilogue of the data-gradient GEMM"""
        wp = rt.packed(weight, g.kind, x.dtype)
        ctx.rt, ctx.g, ctx.weight, ctx.bias, ctx.wp = rt, g, weight, bias, wp
        ctx.wd = rt.packed(weight, g.kind + "D", x.dtype) if (ctx.needs_input_grad[0] and rt.dgrad_copies) else None
        ctx.has_res = residual is not None
        ctx.save_for_backward(x)
        y = _conv_forward(rt, x, wp, None if bias is None else bias.detach(), g, None if residual is None else residual.detach().contiguous())
        ctx.set_materialize_grads(False)
        return (y, x.view_as(x)) if fork else y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        (x,) = ctx.saved_tensors
        rt, g = ctx.rt, ctx.g
        if dy is None:
            return None, None, None, None, None, None, None
        dy = dy.contiguous()
        if dskip is not None:
            dskip = dskip.contiguous().view(-1, g.L_in, x.shape[-1])
        gb = None if ctx.bias is None else rt.grad_of(ctx.bias)
        gw = rt.grad_of(ctx.weight)
        has_bias = ctx.bias is not None

        def colsum():
            ldy = dy.shape[-1]
            L.check(rt.lib.jen1_colsum(dy.data_ptr(), gb.data_ptr(), dy.numel() // ldy, g.co, ldy, rt.dt_of(dy), rt.stream()), "jen1_colsum")

        # (many-row layers: both products fill the chip, a pair would last their sum; the data gradient runs as the lean register-direct
        # kernel instead and the weight gradient goes to the queue)
        rows = (x.numel() // x.shape[-1])
        many_rows = ((rows + 63) // 64) * ((g.ci + 63) // 64) >= rt.target_wgs or (
            rt.big_wgrads and rt.big_wgrad_unpair and rows >= rt.big_wgrad_rows and g.kind in ("conv", "linear") and rt.dt_of(x) == L.BF16)
        if ctx.needs_input_grad[0] and rt.pair_grads and not many_rows:
            # both gradients of the layer in one launch: they share dY and nothing orders them
            fused, blk = _conv_wgrad(rt, x, dy, gw, g, gb, defer=True)
            dx = _conv_dgrad(rt, dy, ctx.wp, g, ctx.wd, pair_with=blk, residual=dskip).view(x.shape)
            if not fused and has_bias:
                rt.weight_grad(colsum, dy)
            return dx, None, None, None, None, (dy if ctx.has_res else None), None

        def wgrad():
            if not _conv_wgrad(rt, x, dy, gw, g, gb) and has_bias:
                colsum()
        rt.weight_grad(wgrad, x, dy)
        dx = _conv_dgrad(rt, dy, ctx.wp, g, ctx.wd, residual=dskip).view(x.shape) if ctx.needs_input_grad[0] else None
        return dx, None, None, None, None, (dy if ctx.has_res else None), None       # (the residual's gradient IS dy: no launch)


class CausalRows:
    """``causal`` per batch element (int32 [B] on the device, 1 = causal) for a pass that holds causal and non-causal clips side by
    side (the reference runs one pass per task, trainer.py:189-211; the samples of a pass do not interact, so one pass over all of
    them computes the same thing).  The flag enters the network in two places: _Conv1d's padding (blocks.py:45-50) -> a map shift
    per batch element (jen1_gemm_args.map_shift_b), and the causal mask of self-attention (blocks.py:315-319) -> a flag per batch
    element (jen1_attn_small_forward's causal_b)."""

    def __init__(self, flags: torch.Tensor):
        self.flags = flags.to(torch.int32).contiguous()
        self._pads: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}

    def pads(self, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """(-pad_left, +pad_left) per batch element for a kernel of k taps, each one entry longer than the batch"""
        hit = self._pads.get(k)
        if hit is None:
            f = torch.cat([self.flags, self.flags[-1:]])
            pos = torch.where(f != 0, k - 1, (k - 1) // 2).to(torch.int32)
            hit = ((-pos).contiguous(), pos.contiguous())
            self._pads[k] = hit
        return hit

    def twice(self) -> "CausalRows":
        """for the CFG pair: the batch stacked on itself (model.py:349-353)"""
        return CausalRows(torch.cat([self.flags, self.flags]))


def conv1d_same(rt: TrainRuntime, x: torch.Tensor, weight, bias, stride: int, causal, residual=None, fork: bool = False) -> torch.Tensor:
    """_Conv1d (blocks.py:34-53): total padding k - 1, all on the left when causal else split evenly.  ``residual`` (the output's
    shape) is added in the GEMM's epilogue.  ``causal``: bool, or CausalRows (the flag per batch element)."""
    co, ci, k = weight.shape
    B, Lin, _ = x.shape
    if isinstance(causal, CausalRows) and k > 1:
        assert causal.flags.shape[0] == B
        g = ConvGeom("conv", k, stride, (k - 1) // 2, Lin, (Lin - 1) // stride + 1, ci, co, pad_b=causal.pads(k))
        return ConvFn.apply(x, weight, bias, rt, g, residual, fork)
    pad = (k - 1) if (causal is True) else (k - 1) // 2
    return ConvFn.apply(x, weight, bias, rt, ConvGeom("conv", k, stride, pad, Lin, (Lin - 1) // stride + 1, ci, co), residual, fork)


def conv1d_zero_pad(rt: TrainRuntime, x: torch.Tensor, weight, bias, padding: int) -> torch.Tensor:
    """nn.Conv1d(k, padding=p) (Upsample1d with factor 1, blocks.py:76-79)"""
    co, ci, k = weight.shape
    B, Lin, _ = x.shape
    return ConvFn.apply(x, weight, bias, rt, ConvGeom("conv", k, 1, padding, Lin, Lin + 2 * padding - k + 1, ci, co))


def conv_transpose1d(rt: TrainRuntime, x: torch.Tensor, weight, bias, stride: int, padding: int, output_padding: int) -> torch.Tensor:
    """nn.ConvTranspose1d (Upsample1d, blocks.py:80-88)"""
    ci, co, k = weight.shape
    B, Lin, _ = x.shape
    Lout = (Lin - 1) * stride - 2 * padding + k + output_padding
    return ConvFn.apply(x, weight, bias, rt, ConvGeom("convT", k, stride, padding, Lin, Lout, ci, co))


def _big_gemm(rt: "TrainRuntime", a2d: torch.Tensor, b2d: torch.Tensor, out: torch.Tensor, K: int, bias: Optional[torch.Tensor] = None) -> None:
    """out[M][N] = a2d[M][:K] @ b2d[N][:K]^T (+ bias[N], float32) on jen1_big_gemm (one column group)"""
    g = L.BGemmArgs()
    g.a, g.b, g.c, g.ldc = a2d.data_ptr(), b2d.data_ptr(), out.data_ptr(), out.stride(0)      # (one inline group: nothing to copy while capturing)
    g.bias = None if bias is None else bias.data_ptr()
    g.M, g.Ntot, g.K, g.lda, g.ldb, g.n_groups = a2d.shape[0], b2d.shape[0], K, a2d.stride(0), b2d.stride(0), 1
    g.dtype, g.alpha = rt.dt_of(a2d), 1.0
    if rt.stats is not None:
        e = rt.stats.setdefault("big_gemm", [0, 0.0, 0.0])
        e[0] += 1
        e[1] += 2.0 * a2d.shape[0] * b2d.shape[0] * K
        e[2] += (a2d.shape[0] * K + b2d.shape[0] * K + a2d.shape[0] * b2d.shape[0]) * a2d.element_size()
    L.check(rt.lib.jen1_big_gemm(C.byref(g), rt.stream()), "jen1_big_gemm")


class BigLinearFn(Function):
    """A PLAIN bias-free linear with many rows and a big weight -- the to_kv projections of the text context (blocks.py:402-407,
    :428: 2B x 129 rows, 1024 -> 512 / 1024 / 2048) -- on the large-M matrix-core kernels of csrc/big_gemm.hip: forward and data
    gradient are jen1_big_gemm (LDS-DMA staged 128 x 128 tiles; the data gradient reads the transposed compute copy, so both are
    K-contiguous products), the weight gradient is jen1_big_gemm_tn (both operands as they lie in memory, transposing LDS reads,
    float32 atomics straight into ``.grad``).  bf16 compute only (float32 mode keeps jen1_train_gemm)."""

    @staticmethod
    def forward(ctx, x2d, weight, rt: TrainRuntime):
        wp = rt.packed(weight, "linear", x2d.dtype)[0]              # [co][pad8(ci)] in the compute dtype
        ctx.rt, ctx.weight = rt, weight
        ctx.save_for_backward(x2d)
        y = torch.empty((x2d.shape[0], wp.shape[0]), dtype=x2d.dtype, device=x2d.device)
        _big_gemm(rt, x2d, wp, y, wp.shape[1])
        return y

    @staticmethod
    def backward(ctx, dy):
        (x2d,) = ctx.saved_tensors
        rt, weight = ctx.rt, ctx.weight
        dy = dy.contiguous()
        co, ci = weight.shape
        gw = rt.grad_of(weight)
        rows = x2d.shape[0]

        if rt.stats is not None:
            e = rt.stats.setdefault("big_gemm", [0, 0.0, 0.0])
            e[0] += 1
            e[1] += 2.0 * rows * co * ci
            e[2] += (rows * co + rows * ci) * 2 + co * ci * 8

        def wgrad():
            L.check(rt.lib.jen1_big_gemm_tn(dy.data_ptr(), x2d.data_ptr(), gw.data_ptr(), rows, co, ci, dy.stride(0), x2d.stride(0), gw.stride(0), 1.0,
                                            rt.stream()), "jen1_big_gemm_tn")
        rt.weight_grad(wgrad, x2d, dy)
        dx = None
        if ctx.needs_input_grad[0]:
            wt = rt.packed(weight, "linearD", x2d.dtype)[0]         # [ci][pad8(co)]: the data gradient is K-contiguous too
            dx = torch.empty_like(x2d)
            if x2d.shape[1] != ci:
                dx.zero_()
            _big_gemm(rt, dy, wt, dx, co)
        return dx, None, None


class KvBank:
    """Folded operands of ``to_kv_l(norm_context_l(x))`` for all cross-attention layers l (blocks.py:400-407, :427-434; the algebra is in
    csrc/train_kvbank.hip): wf [Ntot][K] bf16 = W diag(gamma) row-stacked, wft [K][Ntot] its transpose (the data gradient's operand),
    bias [Ntot] float32 = W beta.  ``refresh`` folds again when the parameters moved (TrainRuntime.epoch); the device table carries the
    parameters' AND their gradients' addresses (``jen1_kv_layer``) and is rebuilt when one of them changed."""

    def __init__(self, rt: "TrainRuntime", weights, gammas, betas):
        self.rt = rt
        self.refs = [[weakref.ref(t) for t in ts] for ts in (weights, gammas, betas)]
        self.K = weights[0].shape[1]
        self.widths = [w.shape[0] for w in weights]
        assert all(w.shape[1] == self.K and w.shape[0] % 32 == 0 and w.dtype == torch.float32 and w.is_contiguous() for w in weights)
        assert all(g.shape == (self.K,) and b.shape == (self.K,) for g, b in zip(gammas, betas)) and self.K % 64 == 0
        self.offs = [0]
        for n in self.widths:
            self.offs.append(self.offs[-1] + n)
        self.Ntot = self.offs[-1]
        dev = weights[0].device
        self.wf = torch.zeros((self.Ntot, self.K), dtype=torch.bfloat16, device=dev)
        self.wft = torch.zeros((self.K, self.Ntot), dtype=torch.bfloat16, device=dev)
        self.bias = torch.zeros((self.Ntot,), dtype=torch.float32, device=dev)
        self.ones = torch.ones((self.K,), dtype=torch.float32, device=dev)
        self.zeros = torch.zeros((self.K,), dtype=torch.float32, device=dev)
        self.scratch = torch.zeros((2, self.K), dtype=torch.float32, device=dev)       # dgamma / dbeta of the unit affine map (discarded)
        self.dwf = torch.empty((self.Ntot * self.K + self.Ntot,), dtype=torch.float32, device=dev)      # dWf | dbias, zeroed per pass
        self.epoch = -2
        self._tab, self._tab_key = None, None

    def params(self):
        return [[r() for r in rs] for rs in self.refs]

    def alive(self) -> bool:
        return all(r() is not None for rs in self.refs for r in rs)

    def same(self, weights, gammas, betas) -> bool:
        return all(r() is t for rs, ts in zip(self.refs, (weights, gammas, betas)) for r, t in zip(rs, ts))

    def table(self) -> torch.Tensor:
        ws, gs, bs = self.params()
        rt = self.rt
        key = tuple(t.data_ptr() for t in ws + gs + bs) + tuple(rt.grad_of(t).data_ptr() for t in ws + gs + bs)
        if key != self._tab_key:
            ents = (L.KvLayer * len(ws))()
            for i, (w, g, b) in enumerate(zip(ws, gs, bs)):
                e = ents[i]
                e.w, e.gamma, e.beta = w.data_ptr(), g.data_ptr(), b.data_ptr()
                e.gw, e.ggamma, e.gbeta = rt.grad_of(w).data_ptr(), rt.grad_of(g).data_ptr(), rt.grad_of(b).data_ptr()
                e.n0, e.N = self.offs[i], self.widths[i]
            raw = bytes(ents)
            assert not torch.cuda.is_current_stream_capturing(), "the table of a KvBank must exist before a pass is recorded"
            self._tab = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.wf.device)
            self._tab_key = key
        return self._tab

    def refresh(self) -> None:
        if self.epoch == self.rt.epoch:
            return
        rt = self.rt
        L.check(rt.lib.jen1_kv_fold(self.table().data_ptr(), len(self.widths), self.Ntot, self.K, self.wf.data_ptr(), self.wft.data_ptr(), self.Ntot,
                                    self.bias.data_ptr(), rt.stream()), "jen1_kv_fold")
        self.epoch = rt.epoch


class KvSlot:
    """where the attention backward of one cross-attention layer writes dK | dV: a column window [B_eff, Nk, 2C] of the stacked gradient matrix
    (rows ``Ntot`` apart) -- the operand of ONE weight-gradient and ONE data-gradient product (ContextKVFn.backward)"""

    def __init__(self, view: torch.Tensor):
        self.view = view


class ContextKVFn(Function):
    """K | V of every cross-attention layer from the context rows in one standardisation + one product (see KvBank); outputs: one
    [Bk, Nk, 2 C_l] column window of the stacked result per layer (rows Ntot apart: the attention kernels take the pitch)."""

    @staticmethod
    def forward(ctx, rows, rt: TrainRuntime, bank: KvBank, slots, n_share: int):
        Bk, Nk, K = rows.shape
        assert rows.is_contiguous() and rows.dtype == torch.bfloat16 and K == bank.K
        R = Bk * Nk
        x2d = rows.view(R, K)
        xhat = torch.empty_like(x2d)
        stats = torch.empty((R, 2), dtype=torch.float32, device=rows.device)
        L.check(rt.lib.jen1_ln_forward(x2d.data_ptr(), bank.ones.data_ptr(), bank.zeros.data_ptr(), xhat.data_ptr(), stats.data_ptr(), R, K, K, 1e-5,
                                       L.BF16, rt.stream()), "jen1_ln_forward")
        kv_all = torch.empty((R, bank.Ntot), dtype=torch.bfloat16, device=rows.device)
        _big_gemm(rt, xhat, bank.wf, kv_all, K, bias=bank.bias)
        ctx.rt, ctx.bank, ctx.slots, ctx.n_share, ctx.dims = rt, bank, slots, n_share, (Bk, Nk, K)
        ctx.save_for_backward(x2d, xhat, stats)
        kv3 = kv_all.view(Bk, Nk, bank.Ntot)
        return tuple(kv3[:, :, o:o + n] for o, n in zip(bank.offs, bank.widths))

    @staticmethod
    def backward(ctx, *grads):
        x2d, xhat, stats = ctx.saved_tensors
        rt, bank, slots = ctx.rt, ctx.bank, ctx.slots
        Bk, Nk, K = ctx.dims
        R, Ntot = Bk * Nk, bank.Ntot
        dall = slots[0].view._base                                   # [B_eff * Nk, Ntot]
        for slot, gr in zip(slots, grads):
            # ``dall`` is uninitialised memory: a layer whose attention core did NOT write its whole [b_eff, Nk, 2C] window in place (the
            # GEMM attention path of Nq > 64, or an unused output) gets the sharers' blocks zeroed -- autograd has summed them into the
            # [Bk] gradient already -- so that the strided sum below adds nothing for its columns
            if gr is None:
                slot.view.zero_()
            elif gr.data_ptr() != slot.view.data_ptr() or gr.stride() != slot.view[:Bk].stride():
                slot.view[:Bk].copy_(gr)        # (a gradient that did not come from the one-launch attention core's in-place write)
                if slot.view.shape[0] > Bk:
                    slot.view[Bk:].zero_()
        s = rt.stream()
        if ctx.n_share > 1:
            # the unconditional half of the CFG pair read ONE set of context rows: its batch elements' dK | dV blocks, all layers at once
            L.check(rt.lib.jen1_sum_rows_strided(dall.data_ptr() + (Bk - 1) * Nk * Ntot * 2, ctx.n_share, Nk, Ntot, Ntot, L.BF16, s), "jen1_sum_rows_strided")
        dwf, dbias = bank.dwf[: Ntot * K], bank.dwf[Ntot * K:]
        L.check(rt.lib.jen1_memset_zero(dbias.data_ptr(), Ntot * 4, s), "jen1_memset_zero")           # (the column sums accumulate; dWf is stored)
        L.check(rt.lib.jen1_colsum(dall.data_ptr(), dbias.data_ptr(), R, Ntot, Ntot, L.BF16, s), "jen1_colsum")
        L.check(rt.lib.jen1_big_gemm_tn_store(dall.data_ptr(), xhat.data_ptr(), dwf.data_ptr(), R, Ntot, K, Ntot, K, K, 1.0, s), "jen1_big_gemm_tn_store")
        L.check(rt.lib.jen1_kv_fold_backward(bank.table().data_ptr(), len(bank.widths), Ntot, K, dwf.data_ptr(), dbias.data_ptr(), s), "jen1_kv_fold_backward")
        if rt.stats is not None:
            e = rt.stats.setdefault("big_gemm", [0, 0.0, 0.0])
            e[0] += 1
            e[1] += 2.0 * R * Ntot * K
            e[2] += (R * Ntot + R * K) * 2 + Ntot * K * 8
        dx = None
        if ctx.needs_input_grad[0]:
            dxh = torch.empty_like(xhat)
            _big_gemm(rt, dall[:R], bank.wft, dxh, Ntot)
            dx = torch.empty_like(x2d)
            L.check(rt.lib.jen1_ln_backward(dxh.data_ptr(), x2d.data_ptr(), stats.data_ptr(), bank.ones.data_ptr(), dx.data_ptr(), bank.scratch[0].data_ptr(),
                                            bank.scratch[1].data_ptr(), R, K, K, L.BF16, s), "jen1_ln_backward")
            dx = dx.view(Bk, Nk, K)
        return dx, None, None, None, None


def context_kv(rt: TrainRuntime, rows: torch.Tensor, weights, gammas, betas, b_eff: int):
    """-> [(kv_l, KvSlot_l)] for the cross-attention layers whose parameters are given: kv_l [Bk, Nk, 2 C_l]; ``b_eff``: batch elements of
    the pass (the last b_eff - (Bk - 1) of them read the last set of context rows: AttentionCoreFn's ``kv_row``)"""
    Bk, Nk, _ = rows.shape
    bank = rt.kv_bank(weights, gammas, betas)
    dall = torch.empty((b_eff * Nk, bank.Ntot), dtype=torch.bfloat16, device=rows.device)
    d3 = dall.view(b_eff, Nk, bank.Ntot)
    slots = tuple(KvSlot(d3[:, :, o:o + n]) for o, n in zip(bank.offs, bank.widths))
    outs = ContextKVFn.apply(rows, rt, bank, slots, b_eff - (Bk - 1))
    return list(zip(outs, slots))


def linear(rt: TrainRuntime, x: torch.Tensor, weight, bias=None, residual=None, fork: bool = False):
    """nn.Linear on the last axis; x [..., pad8(in)] -> [..., pad8(out)] (+ ``residual`` of the output's shape, in the epilogue).
    ``fork``: -> (y, x) with x's two gradients merged in the data-gradient GEMM (ConvFn.forward)"""
    co, ci = weight.shape
    lead = x.shape[:-1]
    rows = x.numel() // x.shape[-1]
    if fork:
        if not (x.is_contiguous() and x.requires_grad):
            return linear(rt, x, weight, bias, residual), x
        y, xa = ConvFn.apply(x.view(1, rows, x.shape[-1]), weight, bias, rt, ConvGeom("linear", 1, 1, 0, rows, rows, ci, co),
                             None if residual is None else residual.reshape(1, rows, residual.shape[-1]), True)
        return y.view(*lead, y.shape[-1]), xa.view(x.shape)
    if (rt.big_linears and residual is None and bias is None and x.dtype == torch.bfloat16 and rows >= 1024 and ci >= 512 and co >= 512 and co % 64 == 0
            and ci % 64 == 0 and x.shape[-1] == ci):
        return BigLinearFn.apply(x.reshape(rows, x.shape[-1]), weight, rt).view(*lead, co)
    y = ConvFn.apply(x.reshape(1, rows, x.shape[-1]), weight, bias, rt, ConvGeom("linear", 1, 1, 0, rows, rows, ci, co),
                     None if residual is None else residual.reshape(1, rows, residual.shape[-1]))
    return y.view(*lead, y.shape[-1])


# =====================================================================================================================
# GroupNorm (+FiLM) (+SiLU): ConvBlock1d prologue (blocks.py:137-143), Transformer1d.group_norm (blocks.py:509)
# =====================================================================================================================
class GroupNormFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, film, rt: TrainRuntime, C: int, groups: int, eps: float, silu: bool, dfilm_slot=None, fork: bool = False):
        """``fork``: also return ``x`` itself (an alias) for the branch that goes AROUND the norm (ResnetBlock1d's residual / shortcut,
        blocks.py:229-231): the gradient of that branch then arrives here as a second argument of ``backward`` and is added inside the
        GroupNorm backward kernel (jen1_gn_backward_add) instead of by an accumulation launch of autograd.
        ``film`` [B, >= 2C]: contiguous, or a column slice of a wider matrix (rows ``film.stride(0)`` apart: FilmBankFn); with
        ``dfilm_slot`` (FilmSlot: the same slice of the matrix of FiLM gradients) the backward kernel writes the gradient there itself"""
        B, Lx, ld = x.shape
        assert x.is_contiguous() and ld >= C
        dt = rt.dt_of(x)
        sums = torch.empty((B, groups, 2), dtype=torch.float32, device=x.device)
        y = (torch.zeros_like if ld != C else torch.empty_like)(x)
        s = rt.stream()
        film_ld = 0
        if film is not None:
            if not (film.dim() == 2 and film.stride(1) == 1 and film.stride(0) >= film.shape[1]):
                film = film.contiguous().view(B, -1)
            film_ld = film.stride(0)
            assert film.dtype == x.dtype and film.shape[-1] >= 2 * C and film.shape[0] == B
            if dfilm_slot is not None:
                sv = dfilm_slot.view
                assert sv.shape == film.shape and sv.stride() == film.stride() and sv.dtype == film.dtype
        L.check(rt.lib.jen1_gn_forward(x.data_ptr(), sums.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                       None if film is None else film.data_ptr(), film_ld,
                                       y.data_ptr(), B, Lx, C, ld, groups, float(eps), 1 if silu else 0, dt, s), "jen1_gn_forward")
        ctx.rt, ctx.C, ctx.groups, ctx.eps, ctx.silu, ctx.gamma, ctx.beta = rt, C, groups, eps, silu, gamma, beta
        ctx.has_film = film is not None
        ctx.film_ld, ctx.dfilm_slot = film_ld, (dfilm_slot if film is not None else None)
        ctx.save_for_backward(x, sums, film if film is not None else x.new_empty(0))
        ctx.set_materialize_grads(False)           # an alias nobody used comes back as None, not as a tensor of zeros
        if fork == 2:                              # ... and a second alias: the tensor is also a skip connection of the U-Net
            return y, x.view_as(x), x.view_as(x)
        if fork:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dskip=None, dskip2=None):
        x, sums, film = ctx.saved_tensors
        rt, C, groups = ctx.rt, ctx.C, ctx.groups
        B, Lx, ld = x.shape
        if dy is None:
            dy = torch.zeros_like(x)
        dy = dy.contiguous()
        if dskip is None:
            dskip, dskip2 = dskip2, None
        if dskip is not None:
            dskip = dskip.contiguous()
            assert dskip.shape == x.shape and dskip.dtype == x.dtype
        if dskip2 is not None:
            dskip2 = dskip2.contiguous()
            assert dskip2.shape == x.shape and dskip2.dtype == x.dtype
        dt = rt.dt_of(x)
        dx = (torch.zeros_like if ld != C else torch.empty_like)(x)
        P = torch.empty((B, C, 4), dtype=torch.float32, device=x.device)
        Gm = torch.empty((B, groups, 2), dtype=torch.float32, device=x.device)
        slot = ctx.dfilm_slot
        dfilm = None
        if ctx.has_film:
            dfilm = slot.view if slot is not None else torch.empty((B, 2 * C), dtype=torch.float32, device=x.device)
        flags = (1 if ctx.silu else 0) | (2 if slot is not None else 0)
        L.check(rt.lib.jen1_gn_backward_add2(dy.data_ptr(), x.data_ptr(), sums.data_ptr(), ctx.gamma.data_ptr(), ctx.beta.data_ptr(),
                                             film.data_ptr() if ctx.has_film else None, ctx.film_ld,
                                             dx.data_ptr(), None if dskip is None else dskip.data_ptr(),
                                             None if dskip2 is None else dskip2.data_ptr(),
                                             rt.grad_of(ctx.gamma).data_ptr(), rt.grad_of(ctx.beta).data_ptr(),
                                             None if dfilm is None else dfilm.data_ptr(), P.data_ptr(), Gm.data_ptr(), B, Lx, C, ld,
                                             groups, float(ctx.eps), flags, dt, rt.stream()), "jen1_gn_backward_add2")
        if slot is not None:
            df = slot.view                     # (written in place: FilmBankFn.backward recognises its own slice)
            slot.written()
        elif dfilm is not None:
            df = torch.zeros((B, film.shape[-1]), dtype=film.dtype, device=x.device) if film.shape[-1] != 2 * C else None
            if df is None:
                df = dfilm.to(film.dtype)
            else:
                df[:, :2 * C] = dfilm
        else:
            df = None
        return dx, None, None, df, None, None, None, None, None, None, None


def group_norm(rt, x, gamma, beta, C, groups, eps, film=None, silu=False, dfilm_slot=None, fork=False):
    """``fork``: -> (norm(x), x) with the two gradients of x merged inside the backward kernel (GroupNormFn.forward); ``fork=2``:
    -> (norm(x), x, x) for a tensor that has a third consumer (a skip connection)"""
    if fork and x.shape[-1] != C:
        y = GroupNormFn.apply(x, gamma, beta, film, rt, C, groups, eps, silu, dfilm_slot, False)     # (padded rows: plain accumulation)
        return (y, x, x) if fork == 2 else (y, x)
    return GroupNormFn.apply(x, gamma, beta, film, rt, C, groups, eps, silu, dfilm_slot, fork)


# =====================================================================================================================
# the FiLM projections of every block: MappingToScaleShift (blocks.py:177-196) = Linear(SiLU(mapping)) -> (scale, shift), one per
# ResnetBlock1d, all reading the same [B, features] vector
# =====================================================================================================================
class FilmSlot:
    """where block i's FiLM gradient goes (a column slice of the gradient matrix) + what to do once it is there: queue the block's
    weight / bias gradient.  That happens inside the block's own GroupNorm backward node, so "the gradient of a block's input exists =>
    all its parameter gradients are enqueued" (TrainGraph._mark, the overlapped exchange) still holds."""

    def __init__(self, view: torch.Tensor, written):
        self.view, self.written = view, written


class FilmBankFn(Function):
    """The ~40 projections as ONE GEMM against the row-stacked weights (TrainRuntime.packed_bank): y [B, sum 2 C_i], block i reads its
    column slice in place (GroupNormFn takes the row pitch).  Backward: every block's GroupNorm kernel writes its FiLM gradient into
    the same slice of ONE gradient matrix, which is the operand of one data-gradient GEMM (instead of ~40 GEMMs + ~40 conversions +
    ~40 accumulations of the mapping's gradient); the weight / bias gradients stay per block (their .grad buffers are separate) and
    go to the weight-gradient queue as soon as the block's slice is written (FilmSlot)."""

    @staticmethod
    def forward(ctx, smap, rt: TrainRuntime, weights, biases, slots):
        B, F = smap.shape
        wall, ball = rt.packed_bank(weights, biases, smap.dtype)
        total = wall.shape[1]
        g = ConvGeom("linear", 1, 1, 0, B, B, weights[0].shape[1], total)
        y = _conv_forward(rt, smap.view(1, B, F), wall, ball, g).view(B, total)
        ctx.rt, ctx.g, ctx.wall, ctx.slots = rt, g, wall, slots
        outs, off = [], 0
        for w in weights:
            outs.append(y[:, off:off + w.shape[0]])
            off += w.shape[0]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        rt = ctx.rt
        dY = ctx.slots[0].view._base
        for slot, gr in zip(ctx.slots, grads):
            if gr is not None and (gr.data_ptr() != slot.view.data_ptr() or gr.stride() != slot.view.stride()):
                slot.view.copy_(gr)            # (a gradient that did not come from GroupNormFn's in-place write)
                slot.written()
        B = dY.shape[0]
        dx = _conv_dgrad(rt, dY.view(1, B, dY.shape[1]), ctx.wall, ctx.g).view(B, -1) if ctx.needs_input_grad[0] else None
        return dx, None, None, None, None


def film_bank(rt: TrainRuntime, smap: torch.Tensor, weights, biases):
    """-> [(film_i, FilmSlot_i)]: block i's (scale | shift) [B, 2 C_i] and the ``dfilm_slot`` of its group_norm"""
    B, F = smap.shape
    total = sum(w.shape[0] for w in weights)
    dY = torch.zeros((B, total), dtype=smap.dtype, device=smap.device)
    xs = smap.detach().view(1, B, F)
    slots, off = [], 0
    for w, b in zip(weights, biases):
        co = w.shape[0]
        view = dY[:, off:off + co]
        lg = ConvGeom("linear", 1, 1, 0, B, B, w.shape[1], co)

        def written(w=w, b=b, view=view, lg=lg):
            gw, gb = rt.grad_of(w), rt.grad_of(b)
            rt.weight_grad(lambda: _conv_wgrad(rt, xs, view, gw, lg, gb), xs, dY)
        slots.append(FilmSlot(view, written))
        off += co
    outs = FilmBankFn.apply(smap, rt, tuple(weights), tuple(biases), tuple(slots))
    return list(zip(outs, slots))


# =====================================================================================================================
# LayerNorm (blocks.py:400-401), GELU / SiLU
# =====================================================================================================================
class LayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, rt: TrainRuntime, C: int, eps: float, fork: bool = False):
        """``fork``: also return ``x`` (an alias) for the residual branch around the sub-block (blocks.py:486-488); its gradient comes
        back as the second argument of ``backward`` and is added inside the kernel (jen1_ln_backward_add)"""
        assert not fork or x.is_contiguous()
        x = x.contiguous()
        ld = x.shape[-1]
        rows = x.numel() // ld
        y = (torch.zeros_like if ld != C else torch.empty_like)(x)
        stats = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
        L.check(rt.lib.jen1_ln_forward(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), stats.data_ptr(), rows, C, ld,
                                       float(eps), rt.dt_of(x), rt.stream()), "jen1_ln_forward")
        ctx.rt, ctx.C, ctx.gamma, ctx.beta = rt, C, gamma, beta
        ctx.save_for_backward(x, stats)
        ctx.set_materialize_grads(False)           # an alias nobody used comes back as None, not as a tensor of zeros
        if fork:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        x, stats = ctx.saved_tensors
        rt, C = ctx.rt, ctx.C
        if dy is None:
            dy = torch.zeros_like(x)
        dy = dy.contiguous()
        if dskip is not None:
            dskip = dskip.contiguous()
            assert dskip.shape == x.shape and dskip.dtype == x.dtype
        ld = x.shape[-1]
        rows = x.numel() // ld
        # (norm_context over the text embedding: the input needs no gradient, the kernel then only sums the columns)
        dx = (torch.zeros_like if ld != C else torch.empty_like)(x) if (ctx.needs_input_grad[0] or dskip is not None) else None
        L.check(rt.lib.jen1_ln_backward_add(dy.data_ptr(), x.data_ptr(), stats.data_ptr(), ctx.gamma.data_ptr(),
                                            None if dx is None else dx.data_ptr(),
                                            None if dskip is None else dskip.data_ptr(),
                                            rt.grad_of(ctx.gamma).data_ptr(), rt.grad_of(ctx.beta).data_ptr(), rows, C, ld, rt.dt_of(x),
                                            rt.stream()), "jen1_ln_backward_add")
        return dx, None, None, None, None, None, None


class DualLayerNormFn(Function):
    """(LN(x; gamma1, beta1), LN(x; gamma2, beta2), alias of x) in one launch each way -- self-attention's ``norm`` and ``norm_context``
    over the same x (blocks.py:427-429 with context = x): shared statistics; the backward is ONE LayerNorm backward of
    dy1 gamma1 + dy2 gamma2 with the residual branch's gradient added in the kernel (jen1_ln2_forward / jen1_ln2_backward_add)"""

    @staticmethod
    def forward(ctx, x, g1, b1, g2, b2, rt: TrainRuntime, eps: float):
        assert x.is_contiguous()
        C = ld = x.shape[-1]
        rows = x.numel() // ld
        y1, y2 = torch.empty_like(x), torch.empty_like(x)
        stats = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
        L.check(rt.lib.jen1_ln2_forward(x.data_ptr(), g1.data_ptr(), b1.data_ptr(), g2.data_ptr(), b2.data_ptr(), y1.data_ptr(), y2.data_ptr(),
                                        stats.data_ptr(), rows, C, ld, float(eps), rt.dt_of(x), rt.stream()), "jen1_ln2_forward")
        ctx.rt, ctx.params = rt, (g1, b1, g2, b2)
        ctx.save_for_backward(x, stats)
        ctx.set_materialize_grads(False)
        return y1, y2, x.view_as(x)

    @staticmethod
    def backward(ctx, dy1, dy2, dskip=None):
        x, stats = ctx.saved_tensors
        rt = ctx.rt
        g1, b1, g2, b2 = ctx.params
        dy1 = torch.zeros_like(x) if dy1 is None else dy1.contiguous()
        dy2 = torch.zeros_like(x) if dy2 is None else dy2.contiguous()
        if dskip is not None:
            dskip = dskip.contiguous()
            assert dskip.shape == x.shape and dskip.dtype == x.dtype
        C = ld = x.shape[-1]
        dx = torch.empty_like(x)
        L.check(rt.lib.jen1_ln2_backward_add(dy1.data_ptr(), dy2.data_ptr(), x.data_ptr(), stats.data_ptr(), g1.data_ptr(), g2.data_ptr(),
                                             dx.data_ptr(), None if dskip is None else dskip.data_ptr(), rt.grad_of(g1).data_ptr(),
                                             rt.grad_of(b1).data_ptr(), rt.grad_of(g2).data_ptr(), rt.grad_of(b2).data_ptr(),
                                             x.numel() // ld, C, ld, rt.dt_of(x), rt.stream()), "jen1_ln2_backward_add")
        return dx, None, None, None, None, None, None


def layer_norm(rt, x, gamma, beta, eps: float = 1e-5, fork: bool = False):
    """``fork``: -> (norm(x), x) with the two gradients of x merged inside the backward kernel (LayerNormFn.forward)"""
    if fork and (not x.is_contiguous() or x.shape[-1] != gamma.shape[0]):
        return LayerNormFn.apply(x, gamma, beta, rt, gamma.shape[0], eps, False), x
    return LayerNormFn.apply(x, gamma, beta, rt, gamma.shape[0], eps, fork)


class ActFn(Function):
    """mode 0: GELU(erf); mode 1: SiLU"""

    @staticmethod
    def forward(ctx, x, rt: TrainRuntime, mode: int):
        x = x.contiguous()
        y = torch.empty_like(x)
        L.check(rt.lib.jen1_act_forward(x.data_ptr(), y.data_ptr(), x.numel(), mode, rt.dt_of(x), rt.stream()), "jen1_act_forward")
        ctx.rt, ctx.mode = rt, mode
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        L.check(ctx.rt.lib.jen1_act_backward(dy.data_ptr(), x.data_ptr(), dx.data_ptr(), x.numel(), ctx.mode, ctx.rt.dt_of(x), ctx.rt.stream()),
                "jen1_act_backward")
        return dx, None, None


class ConcatScaleFn(Function):
    """torch.cat([a, b * scale], dim=-1) on dense channel-last rows (the skip concat of the up path, blocks.py:732-734)"""

    @staticmethod
    def forward(ctx, a, b, rt: TrainRuntime, scale: float):
        a, b = a.contiguous(), b.contiguous()
        Ca, Cb = a.shape[-1], b.shape[-1]
        out = torch.empty(a.shape[:-1] + (Ca + Cb,), dtype=a.dtype, device=a.device)
        L.check(rt.lib.jen1_concat2(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel() // Ca, Ca, Cb, float(scale), rt.dt_of(a), rt.stream()),
                "jen1_concat2")
        ctx.rt, ctx.scale, ctx.Ca, ctx.Cb = rt, scale, Ca, Cb
        return out

    @staticmethod
    def backward(ctx, d):
        d = d.contiguous()
        rt, Ca, Cb = ctx.rt, ctx.Ca, ctx.Cb
        da = torch.empty(d.shape[:-1] + (Ca,), dtype=d.dtype, device=d.device)
        db = torch.empty(d.shape[:-1] + (Cb,), dtype=d.dtype, device=d.device)
        L.check(rt.lib.jen1_split2(d.data_ptr(), da.data_ptr(), db.data_ptr(), d.numel() // (Ca + Cb), Ca, Cb, float(ctx.scale), rt.dt_of(d),
                                   rt.stream()), "jen1_split2")
        return da, db, None, None


def concat_scale(rt, a, b, scale: float):
    if a.shape[-1] % 8 or b.shape[-1] % 8 or a.dtype != b.dtype:
        return torch.cat([a, b * scale], dim=-1)
    return ConcatScaleFn.apply(a, b, rt, scale)


def gelu(rt, x):
    return ActFn.apply(x, rt, 0)


def silu(rt, x):
    return ActFn.apply(x, rt, 1)


# =====================================================================================================================
# attention core (AttentionBase.forward, math path: blocks.py:355-380)
# =====================================================================================================================
def _rows_view(t: torch.Tensor) -> Tuple[int, int]:
    """(pointer, row pitch) of a [B, N, C] tensor or last-axis slice of one (k / v halves of to_kv's output)"""
    assert t.dim() == 3 and t.stride(2) == 1 and t.stride(0) == t.shape[1] * t.stride(1), "rows must be uniformly strided"
    return t.data_ptr(), t.stride(1)


class AttentionCoreFn(Function):
    @staticmethod
    def forward(ctx, q, kv, rt: TrainRuntime, heads: int, causal: bool, kv_mask=None, kv_row=None, dkv_slot=None):
        """``dkv_slot`` (KvSlot, one-launch core only): kv is a column window of the stacked projection of all cross-attention layers
        (ContextKVFn; rows a pitch apart) and dK | dV are written into the same window of the stacked gradient matrix.
        kv: [B, Nk, 2 C] = to_kv's output (K | V): the gradient comes back as ONE tensor (two column windows written by the data-gradient
        GEMMs) instead of two slice gradients that autograd pads with zeros and adds.  ``kv_row`` (int32 [B], one-launch core only): batch
        element b attends to kv[kv_row[b]]; kv then has Bk = kv_row[-1] + 1 rows, the first Bk - 1 mapped one to one, the last one shared by
        all the others (the CFG pair's unconditional half)."""
        mid = kv.shape[-1] // 2
        k, v = kv[..., :mid], kv[..., mid:]
        B, Nq, C = q.shape
        Nk = k.shape[1]
        d = C // heads
        dt = rt.dt_of(q)
        Z = B * heads
        ldS = pad8(Nk)
        qp, ldq = _rows_view(q)
        kp, ldk = _rows_view(k)
        vp, ldv = _rows_view(v)
        scale = d ** -0.5
        P = torch.empty((Z, Nq, ldS), dtype=q.dtype, device=q.device)
        O = torch.empty((B, Nq, C), dtype=q.dtype, device=q.device)
        ctx.rt, ctx.heads, ctx.scale = rt, heads, scale
        ctx.small = bool(rt.small_attn and rt.lib.jen1_attn_small_fits(Nq, Nk, d, dt))
        rows = causal if isinstance(causal, CausalRows) else None
        if rows is not None and not ctx.small:
            raise L.Jen1HipError(f"a pass that mixes causal and non-causal clips needs the one-launch attention core "
                                 f"(Nq = {Nq}, Nk = {Nk}, d = {d} do not fit one workgroup)")
        assert kv_mask is None or ctx.small, "kv_mask rides on the one-launch attention core only (attention() multiplies otherwise)"
        ctx.kv_mask = None if kv_mask is None else kv_mask.to(torch.float32).contiguous()
        ctx.kv_row = kv_row
        ctx.dkv_slot = dkv_slot
        assert dkv_slot is None or ctx.small
        assert kv_row is None or (ctx.small and kv_row.dtype == torch.int32 and kv_row.shape[0] == B)
        if ctx.small:
            assert rows is None or rows.flags.shape[0] == B
            assert ctx.kv_mask is None or ctx.kv_mask.shape == (B, Nk)
            L.check(rt.lib.jen1_attn_small_forward_rows(qp, ldq, kp, ldk, vp, ldv, O.data_ptr(), C, P.data_ptr(), ldS, B, heads, Nq, Nk, d,
                                                        float(scale), 1 if (rows is None and causal) else 0,
                                                        None if rows is None else rows.flags.data_ptr(),
                                                        None if ctx.kv_mask is None else ctx.kv_mask.data_ptr(),
                                                        None if kv_row is None else kv_row.data_ptr(), dt, rt.stream()),
                    "jen1_attn_small_forward_rows")
            ctx.save_for_backward(q, kv, P)
            return O
        S = torch.empty((Z, Nq, ldS), dtype=torch.float32, device=q.device)
        rt.gemm(_operand(qp, ldq, 1, zs0=Nq * ldq, zs1=d, zdiv=heads), _operand(kp, ldk, 1, zs0=Nk * ldk, zs1=d, zdiv=heads),
                S.data_ptr(), Nq, Nk, d, dtype=dt, batches=Z, ldc_m=ldS, c_zs0=Nq * ldS, c_f32=True, alpha=scale)
        L.check(rt.lib.jen1_softmax_forward(S.data_ptr(), P.data_ptr(), Z * Nq, Nq, Nk, ldS, ldS, 1 if causal else 0, dt, rt.stream()),
                "jen1_softmax_forward")
        rt.gemm(_operand(P.data_ptr(), ldS, 1, zs0=Nq * ldS), _operand(vp, 1, ldv, zs0=Nk * ldv, zs1=d, zdiv=heads),
                O.data_ptr(), Nq, d, Nk, dtype=dt, batches=Z, ldc_m=C, c_zs0=Nq * C, c_zs1=d, c_zdiv=heads)
        ctx.rt, ctx.heads, ctx.scale = rt, heads, scale
        ctx.save_for_backward(q, kv, P)
        return O

    @staticmethod
    def backward(ctx, dO):
        q, kv, P = ctx.saved_tensors
        mid = kv.shape[-1] // 2
        k, v = kv[..., :mid], kv[..., mid:]
        rt, heads, scale = ctx.rt, ctx.heads, ctx.scale
        dO = dO.contiguous()
        B, Nq, C = q.shape
        Nk = k.shape[1]
        d = C // heads
        dt = rt.dt_of(q)
        Z = B * heads
        ldS = P.shape[-1]
        qp, ldq = _rows_view(q)
        kp, ldk = _rows_view(k)
        vp, ldv = _rows_view(v)
        dQ = torch.empty((B, Nq, C), dtype=q.dtype, device=q.device)
        slot = ctx.dkv_slot
        if slot is not None:
            # dK | dV of every batch element straight into this layer's window of the stacked gradient matrix; the blocks of the batch
            # elements that share context rows are added up there for all layers at once (ContextKVFn.backward)
            dW = slot.view
            assert dW.shape[0] == B and dW.shape[1] == Nk and dW.shape[2] == 2 * C and dW.stride(2) == 1 and dW.stride(0) == Nk * dW.stride(1)
            esz, ldw = dW.element_size(), dW.stride(1)
            L.check(rt.lib.jen1_attn_small_backward_rows(qp, ldq, kp, ldk, vp, ldv, P.data_ptr(), ldS, dO.data_ptr(), C, dQ.data_ptr(), C,
                                                         dW.data_ptr(), ldw, dW.data_ptr() + C * esz, ldw, B, heads, Nq, Nk, d,
                                                         float(scale), None if ctx.kv_mask is None else ctx.kv_mask.data_ptr(),
                                                         None if ctx.kv_row is None else ctx.kv_row.data_ptr(), dt, rt.stream()),
                    "jen1_attn_small_backward_rows")
            return dQ, dW[: kv.shape[0]], None, None, None, None, None, None
        dKV = torch.empty((B, Nk, 2 * C), dtype=q.dtype, device=q.device)
        esz = dKV.element_size()
        if ctx.small:
            kv_row = ctx.kv_row
            L.check(rt.lib.jen1_attn_small_backward_rows(qp, ldq, kp, ldk, vp, ldv, P.data_ptr(), ldS, dO.data_ptr(), C, dQ.data_ptr(), C,
                                                         dKV.data_ptr(), 2 * C, dKV.data_ptr() + C * esz, 2 * C, B, heads, Nq, Nk, d,
                                                         float(scale), None if ctx.kv_mask is None else ctx.kv_mask.data_ptr(),
                                                         None if kv_row is None else kv_row.data_ptr(), dt, rt.stream()),
                    "jen1_attn_small_backward_rows")
            if kv_row is not None:
                # dk / dv were written per batch element; the rows behind the one-to-one part all read the same K / V: their sum, in place
                Bk = kv.shape[0]
                n = Nk * 2 * C
                L.check(rt.lib.jen1_sum_rows_inplace(dKV.data_ptr() + (Bk - 1) * n * esz, B - (Bk - 1), n, dt, rt.stream()), "jen1_sum_rows_inplace")
                dKV = dKV[:Bk]
            return dQ, dKV, None, None, None, None, None, None
        dP = torch.empty((Z, Nq, ldS), dtype=torch.float32, device=q.device)
        dS = torch.empty((Z, Nq, ldS), dtype=q.dtype, device=q.device)
        o_do = lambda ld_r, ld_k: _operand(dO.data_ptr(), ld_r, ld_k, zs0=Nq * C, zs1=d, zdiv=heads)
        # dP = dO V^T
        rt.gemm(o_do(C, 1), _operand(vp, ldv, 1, zs0=Nk * ldv, zs1=d, zdiv=heads), dP.data_ptr(), Nq, Nk, d, dtype=dt, batches=Z,
                ldc_m=ldS, c_zs0=Nq * ldS, c_f32=True)
        L.check(rt.lib.jen1_softmax_backward(P.data_ptr(), dP.data_ptr(), dS.data_ptr(), Z * Nq, Nk, ldS, ldS, dt, rt.stream()),
                "jen1_softmax_backward")
        # dQ = scale dS K ; dK = scale dS^T Q ; dV = P^T dO
        rt.gemm(_operand(dS.data_ptr(), ldS, 1, zs0=Nq * ldS), _operand(kp, 1, ldk, zs0=Nk * ldk, zs1=d, zdiv=heads),
                dQ.data_ptr(), Nq, d, Nk, dtype=dt, batches=Z, ldc_m=C, c_zs0=Nq * C, c_zs1=d, c_zdiv=heads, alpha=scale)
        rt.gemm(_operand(dS.data_ptr(), 1, ldS, zs0=Nq * ldS), _operand(qp, 1, ldq, zs0=Nq * ldq, zs1=d, zdiv=heads),
                dKV.data_ptr(), Nk, d, Nq, dtype=dt, batches=Z, ldc_m=2 * C, c_zs0=Nk * 2 * C, c_zs1=d, c_zdiv=heads, alpha=scale)
        rt.gemm(_operand(P.data_ptr(), 1, ldS, zs0=Nq * ldS), o_do(1, C),
                dKV.data_ptr() + C * esz, Nk, d, Nq, dtype=dt, batches=Z, ldc_m=2 * C, c_zs0=Nk * 2 * C, c_zs1=d, c_zdiv=heads)
        return dQ, dKV, None, None, None, None, None, None


def attention_core(rt, q, kv, heads: int, causal, kv_mask=None, kv_row=None, dkv_slot=None):
    """``kv_mask`` [B, Nk]: multiplied into the rows of K and V (blocks.py:431-434) -- inside the one-launch kernels when the shape
    fits them, else as a tensor product before the GEMM path.  ``kv_row``: see AttentionCoreFn.forward (GEMM path: the rows are
    gathered first)"""
    B, Nq, C = q.shape
    small = bool(rt.small_attn and rt.lib.jen1_attn_small_fits(Nq, kv.shape[1], C // heads, rt.dt_of(q)))
    if not small:
        dkv_slot = None                 # (the GEMM path takes contiguous K | V; the stacked window's gradient is then copied in: ContextKVFn.backward)
    if kv_row is not None and not small:
        kv = kv.index_select(0, kv_row.to(torch.int64))
        kv_row = None
    if kv_mask is not None and not small:
        kv = kv * kv_mask.to(kv.dtype)[:, :, None]
        kv_mask = None
    return AttentionCoreFn.apply(q, kv if dkv_slot is not None else kv.contiguous(), rt, heads, causal, kv_mask, kv_row, dkv_slot)


# =====================================================================================================================
# the two ends of a pass (csrc/train_glue.hip): input packing, context rows, time features, CFG combine + loss
# =====================================================================================================================
class ContextRowsFn(Function):
    """cat([embedding, time token]) (model.py:315-316), CFG-dropout rows swapped to the fixed embedding (:323-328), the pair's
    unconditional half (:333), cast to the compute dtype: one launch.  Gradients: the time token's as a tensor, the fixed
    embedding's straight into its ``.grad``."""

    @staticmethod
    def forward(ctx, tok, fixed, emb, drop, rt: TrainRuntime, nrep: int):
        B, NL, F = emb.shape
        N = NL + (1 if tok is not None else 0)
        out = torch.empty(((B + 1) if nrep == 0 else nrep * B, N, F), dtype=rt.tdtype, device=emb.device)
        d8 = None if drop is None else drop.to(torch.uint8)
        L.check(rt.lib.jen1_train_context(emb.data_ptr(), None if tok is None else tok.data_ptr(), fixed.data_ptr(),
                                          None if d8 is None else d8.data_ptr(), out.data_ptr(), B, NL, N, F, nrep, rt.dt, rt.stream()), "jen1_train_context")
        ctx.rt, ctx.fixed, ctx.d8, ctx.dims, ctx.has_tok = rt, fixed, d8, (B, NL, N, F, nrep), tok is not None
        return out

    @staticmethod
    def backward(ctx, d):
        rt = ctx.rt
        B, NL, N, F, nrep = ctx.dims
        d = d.contiguous()
        gf = rt.grad_of(ctx.fixed)
        d_tok = torch.empty((B, F), dtype=torch.float32, device=d.device) if ctx.has_tok else None
        L.check(rt.lib.jen1_train_context_backward(d.data_ptr(), None if ctx.d8 is None else ctx.d8.data_ptr(), gf.data_ptr(),
                                                   None if d_tok is None else d_tok.data_ptr(), B, NL, N, F, nrep, rt.dt_of(d), rt.stream()),
                "jen1_train_context_backward")
        return d_tok, None, None, None, None, None


class TimeFeaturesFn(Function):
    """[t, sin(2 pi t w), cos(2 pi t w)] zero-padded to a multiple of 8 columns (LearnedPositionalEmbedding, utils/module.py:58-72)"""

    @staticmethod
    def forward(ctx, t, w, rt: TrainRuntime):
        B, half = t.shape[0], w.shape[0]
        ld = pad8(2 * half + 1)
        f = torch.empty((B, ld), dtype=torch.float32, device=w.device)
        tt = t if t.dtype in (torch.int64, torch.float32) else (t.to(torch.float32) if t.is_floating_point() else t.to(torch.int64))
        isf = 1 if tt.dtype == torch.float32 else 0
        L.check(rt.lib.jen1_time_features_fwd(tt.data_ptr(), isf, w.data_ptr(), f.data_ptr(), B, half, ld, rt.stream()), "jen1_time_features_fwd")
        ctx.rt, ctx.w, ctx.t, ctx.isf, ctx.dims = rt, w, tt, isf, (B, half, ld)
        return f

    @staticmethod
    def backward(ctx, df):
        rt = ctx.rt
        B, half, ld = ctx.dims
        df = df.contiguous()
        gw = rt.grad_of(ctx.w)
        L.check(rt.lib.jen1_time_features_bwd(ctx.t.data_ptr(), ctx.isf, ctx.w.data_ptr(), df.data_ptr(), gw.data_ptr(), B, half, ld, rt.stream()),
                "jen1_time_features_bwd")
        return None, None, None


class CfgLossFn(Function):
    """network output rows [nrep B, T, ld] -> per-sample losses [B]: CFG combine + unbiased-std rescale (model.py:362-369), l2 / l1
    against the target rows, mean over (C, T) (gdm.py:268-272); the backward writes the gradient of the rows directly."""

    @staticmethod
    def forward(ctx, net, tgt, rt: TrainRuntime, B: int, C: int, nrep: int, scale: float, scale_cfg: bool, phi: float, l1: bool):
        assert net.is_contiguous() and net.shape[0] == nrep * B
        T, ld = net.shape[1], net.shape[2]
        loss = torch.empty((B,), dtype=torch.float32, device=net.device)
        a = (B, C, T, ld, nrep, float(scale), 1 if scale_cfg else 0, float(phi), 1 if l1 else 0, rt.dt_of(net))
        L.check(rt.lib.jen1_cfg_loss_forward(net.data_ptr(), tgt.data_ptr(), loss.data_ptr(), *a, rt.stream()), "jen1_cfg_loss_forward")
        ctx.rt, ctx.a = rt, a
        ctx.save_for_backward(net, tgt)
        return loss

    @staticmethod
    def backward(ctx, g):
        net, tgt = ctx.saved_tensors
        rt = ctx.rt
        g = g.to(torch.float32).contiguous()
        dnet = torch.empty_like(net)
        L.check(rt.lib.jen1_cfg_loss_backward(net.data_ptr(), tgt.data_ptr(), g.data_ptr(), dnet.data_ptr(), *ctx.a, rt.stream()), "jen1_cfg_loss_backward")
        return dnet, None, None, None, None, None, None, None, None, None


# =====================================================================================================================
# the differentiable UNet (mirrors UNet1d.forward model.py:225-265 and UNetCFG1d.forward model.py:299-376)
# =====================================================================================================================
class TrainGraph:
    """Differentiable forward of a ``UNetCFG1d`` (jen1_amd/model.py) on the HIP training kernels.

    ``module`` supplies the parameters under the reference's ``state_dict`` names; gradients land in ``param.grad``.
    Call ``invalidate()`` (or let ``FusedAdamW`` do it through ``attach_optimizer``) after every parameter update.
    """

    def __init__(self, module: torch.nn.Module, spec: UNetSpec, compute_dtype: str = "bf16", device="cuda"):
        self.rt = TrainRuntime(compute_dtype, device)
        self.compute_dtype = compute_dtype
        self.spec = spec
        self.p: Dict[str, torch.nn.Parameter] = dict(module.named_parameters())
        missing = [k for k, _ in spec.param_shapes() if k not in self.p]
        assert not missing, f"module lacks parameters: {missing[:4]}"
        self.skip_scale = 2 ** -0.5 if spec.use_skip_scale else 1.0
        self._side: Optional[torch.cuda.Stream] = None
        self._module = weakref.ref(module)
        self.exchange = None        # optim.GradExchange: told when the gradients of a top-level block are complete
        self._kv = None             # {cross-attention name: (K | V window, KvSlot)} of the pass being built (_stacked_context_kv)

    def _mark(self, h: torch.Tensor, region: str) -> torch.Tensor:
        """``h`` enters the top-level block ``region``: once the gradient with respect to ``h`` exists, every parameter
        gradient of the block has been enqueued (autograd runs a block's backward nodes before it reaches its input)"""
        ex = self.exchange
        if ex is not None and ex.active and h.requires_grad:
            ex.expect(region)
            h.register_hook(lambda g, ex=ex, region=region: ex.region_ready(region, also=self.rt.flush_weight_grads()))
        return h

    def invalidate(self) -> None:
        self.rt.invalidate()

    def per_clip_causal_ok(self, T: int, n_context: int) -> bool:
        """whether a pass over clips of T frames with ``n_context`` conditioning tokens may carry the causal flag per clip
        (CausalRows): every attention of the network must fit the one-launch kernels, which take the flag per batch element"""
        sp, rt = self.spec, self.rt
        if not rt.small_attn:
            return False
        nk = n_context + (1 if sp.use_xattn_time else 0)
        Lx = T
        lens = []
        for d in sp.downs:
            Lx = (Lx - 1) // d.factor + 1
            lens.append(Lx)
        trs = [(d.transformer, n) for d, n in zip(sp.downs, lens) if d.transformer]
        if sp.bott_tr:
            trs.append((sp.bott_tr, lens[-1]))
        trs += [(u.transformer, n) for u, n in zip(sp.ups, reversed(lens)) if u.transformer]
        return all(rt.lib.jen1_attn_small_fits(n, n, t.head_features, rt.dt) and rt.lib.jen1_attn_small_fits(n, nk, t.head_features, rt.dt)
                   for t, n in trs)

    def attach_optimizer(self, opt) -> None:
        """after every ``FusedAdamW.step``: re-pack the compute weights of the training path AND drop the inference engine's
        packed copy (``model(x, ...)`` / ``diffusion.sample`` between optimiser steps -- the reference's periodic evaluation
        under no_grad -- must see the new weights)"""
        opt.post_step_hooks.append(self.invalidate)
        m = self._module()
        if m is not None and hasattr(m, "_invalidate_engine"):
            opt.post_step_hooks.append(m._invalidate_engine)

    # ------------------------------------------------------------------ leaves
    def _to_rows(self, x_bct: torch.Tensor) -> torch.Tensor:
        """[B, C, T] float32 -> channel-last [B, T, pad8(C)] in the compute dtype"""
        B, C, T = x_bct.shape
        out = torch.zeros((B, T, pad8(C)), dtype=self.rt.tdtype, device=x_bct.device)
        out[:, :, :C] = x_bct.transpose(1, 2)
        return out

    def _time_features(self, prefix: str, t: torch.Tensor) -> torch.Tensor:
        """LearnedPositionalEmbedding + Linear (utils/module.py:58-79) in float32"""
        w = self.p[f"{prefix}.0.weights"]
        if self.rt.fused_glue:
            fp = TimeFeaturesFn.apply(t, w, self.rt)
        else:
            x = t.to(torch.float32)[:, None]
            freqs = x * w[None, :] * 2 * math.pi
            f = torch.cat([x, freqs.sin(), freqs.cos()], dim=-1)
            fp = torch.zeros((f.shape[0], pad8(f.shape[1])), dtype=torch.float32, device=f.device)
            fp = torch.cat([f, fp[:, f.shape[1]:]], dim=-1)
        return linear(self.rt, fp, self.p[f"{prefix}.1.weight"], self.p[f"{prefix}.1.bias"])

    def mapping(self, t: torch.Tensor) -> torch.Tensor:
        """UNet1d.get_mapping (model.py:204-223, :75-89), float32"""
        rt, p = self.rt, self.p
        m = gelu(rt, self._time_features("to_time.0", t))
        m = gelu(rt, linear(rt, m, p["to_mapping.0.weight"], p["to_mapping.0.bias"]))
        return gelu(rt, linear(rt, m, p["to_mapping.2.weight"], p["to_mapping.2.bias"]))

    def films(self, smap: torch.Tensor) -> Optional[dict]:
        """every ResnetBlock1d's (scale | shift) from one GEMM (FilmBankFn), by block name; None: each block projects for itself"""
        rt, p, sp = self.rt, self.p, self.spec
        if not rt.film_bank or smap.dim() != 2:
            return None
        blocks = ([sp.to_in] + [r for d in sp.downs for r in d.blocks] + [sp.bott_pre, sp.bott_post]
                  + [r for u in sp.ups for r in u.blocks] + [sp.to_out])
        ws = [p[f"{r.name}.to_scale_shift.to_scale_shift.1.weight"] for r in blocks]
        bs = [p[f"{r.name}.to_scale_shift.to_scale_shift.1.bias"] for r in blocks]
        if any(w.shape[0] % 8 or w.shape[1] != ws[0].shape[1] for w in ws) or smap.shape[1] != pad8(ws[0].shape[1]):
            return None
        return {r.name: fs for r, fs in zip(blocks, film_bank(rt, smap, ws, bs))}

    def res_block(self, r: ResSpec, x: torch.Tensor, smap: torch.Tensor, causal: bool, films: Optional[dict] = None, skip: bool = False):
        """ResnetBlock1d.forward (blocks.py:219-231); ``smap`` = SiLU(mapping) in the compute dtype.  ``skip``: -> (output, alias of
        x): x is also a skip connection of the U-Net; the gradient that comes back through the alias is added inside the GroupNorm
        backward kernel together with the residual branch's (no accumulation launches)."""
        rt, p, n = self.rt, self.p, r.name
        xs = x
        # x feeds the first norm AND the residual / shortcut: forked, its two gradients meet inside the GroupNorm backward kernel
        if rt.fork_norms and x.requires_grad and skip and rt.fork_skips:
            h, x, xs = group_norm(rt, x, p[f"{n}.block1.groupnorm.weight"], p[f"{n}.block1.groupnorm.bias"], r.c_in, r.groups, 1e-5, None, True, fork=2)
        elif rt.fork_norms and x.requires_grad:
            h, x = group_norm(rt, x, p[f"{n}.block1.groupnorm.weight"], p[f"{n}.block1.groupnorm.bias"], r.c_in, r.groups, 1e-5, None, True, fork=True)
        else:
            h = group_norm(rt, x, p[f"{n}.block1.groupnorm.weight"], p[f"{n}.block1.groupnorm.bias"], r.c_in, r.groups, 1e-5, None, True)
        h = conv1d_same(rt, h, p[f"{n}.block1.project.conv.weight"], p[f"{n}.block1.project.conv.bias"], 1, causal)
        if films is not None:
            film, slot = films[n]
        else:
            film, slot = linear(rt, smap, p[f"{n}.to_scale_shift.to_scale_shift.1.weight"], p[f"{n}.to_scale_shift.to_scale_shift.1.bias"]), None
        h = group_norm(rt, h, p[f"{n}.block2.groupnorm.weight"], p[f"{n}.block2.groupnorm.bias"], r.c_out, r.groups, 1e-5, film, True, slot)
        if r.has_shortcut:
            x = conv1d_same(rt, x, p[f"{n}.to_out.conv.weight"], p[f"{n}.to_out.conv.bias"], 1, causal)
        # h + x (blocks.py:231) in the epilogue of the second conv
        y = conv1d_same(rt, h, p[f"{n}.block2.project.conv.weight"], p[f"{n}.block2.project.conv.bias"], 1, causal, residual=x)
        return (y, xs) if skip else y

    def attention(self, n: str, x: torch.Tensor, context, context_mask: Optional[torch.Tensor],
                  heads: int, causal: bool, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Attention.forward (blocks.py:415-437): the padding mask multiplies K and V (:431-434); ``residual`` is added by to_out's GEMM.
        ``context``: a tensor, or a one-element list holding it -- the text context feeds the norm_context of all 13 cross-attentions;
        handed on as an alias from one LayerNorm to the next, its 13 gradients are summed inside the LayerNorm backward kernels"""
        rt, p = self.rt, self.p
        box = context if isinstance(context, list) else None
        kv_row = None
        if box is not None:
            context = box[0]
            kv_row = box[1] if len(box) > 1 else None
        pre = self._kv.get(n) if (box is not None and self._kv) else None      # (K | V of this layer from the pass's stacked projection)
        # x feeds the norm(s) AND (as ``residual``) the sum after to_out: forked through the LayerNorms, so its gradients meet
        # inside their backward kernels instead of in accumulation launches
        fork = rt.fork_norms and residual is x and x.requires_grad
        if (fork and pre is None and context is None and rt.dual_norms and x.is_contiguous() and x.shape[-1] == p[f"{n}.norm.weight"].shape[0]
                and x.shape[-1] % 8 == 0 and x.shape[-1] <= 1024):
            # self-attention: both LayerNorms of x in one launch (forward and backward)
            xn, cn, x = DualLayerNormFn.apply(x, p[f"{n}.norm.weight"], p[f"{n}.norm.bias"], p[f"{n}.norm_context.weight"],
                                              p[f"{n}.norm_context.bias"], rt, 1e-5)
            residual = x
        elif fork:
            xn, x = layer_norm(rt, x, p[f"{n}.norm.weight"], p[f"{n}.norm.bias"], fork=True)
            if pre is not None:
                cn = None
            elif context is None:
                cn, x = layer_norm(rt, x, p[f"{n}.norm_context.weight"], p[f"{n}.norm_context.bias"], fork=True)
            elif box is not None and rt.fork_skips and context.requires_grad:
                cn, box[0] = layer_norm(rt, context, p[f"{n}.norm_context.weight"], p[f"{n}.norm_context.bias"], fork=True)
            else:
                cn = layer_norm(rt, context, p[f"{n}.norm_context.weight"], p[f"{n}.norm_context.bias"])
            residual = x
        else:
            ctx = x if context is None else context
            xn = layer_norm(rt, x, p[f"{n}.norm.weight"], p[f"{n}.norm.bias"])
            cn = None if pre is not None else layer_norm(rt, ctx, p[f"{n}.norm_context.weight"], p[f"{n}.norm_context.bias"])
        q = linear(rt, xn, p[f"{n}.to_q.weight"])
        kv, slot = pre if pre is not None else (linear(rt, cn, p[f"{n}.to_kv.weight"]), None)
        # the padding mask multiplies K and V (blocks.py:431-434): inside the attention kernels when the shape fits them
        o = attention_core(rt, q, kv, heads, causal, context_mask, kv_row, slot)
        return linear(rt, o, p[f"{n}.attention.to_out.weight"], p[f"{n}.attention.to_out.bias"], residual=residual)

    def transformer(self, t: TransformerSpec, x: torch.Tensor, embedding, embedding_mask, causal: bool, skip: bool = False):
        """Transformer1d.forward (blocks.py:528-537): the SAME 1x1 conv before and after the blocks.  ``skip``: -> (output, alias of x)
        as in ``res_block``"""
        rt, p, n = self.rt, self.p, t.name
        w, b = p[f"{n}.conv1d.conv.weight"], p[f"{n}.conv1d.conv.bias"]
        xs = x
        if skip and rt.fork_skips and rt.fork_norms and x.requires_grad:
            h, xs = group_norm(rt, x, p[f"{n}.group_norm.weight"], p[f"{n}.group_norm.bias"], t.channels, 32, 1e-6, None, False, fork=True)
        else:
            h = group_norm(rt, x, p[f"{n}.group_norm.weight"], p[f"{n}.group_norm.bias"], t.channels, 32, 1e-6, None, False)
        h = conv1d_same(rt, h, w, b, 1, causal)
        for l in range(t.num_layers):
            bn = f"{n}.blocks.{l}"
            h = self.attention(f"{bn}.attention", h, None, None, t.heads, causal, residual=h)            # (+ h: blocks.py:486-488)
            h = self.attention(f"{bn}.cross_attention", h, embedding, embedding_mask, t.heads, False, residual=h)
            if rt.fork_norms:
                f, h = linear(rt, h, p[f"{bn}.feed_forward.0.weight"], p[f"{bn}.feed_forward.0.bias"], fork=True)
            else:
                f = linear(rt, h, p[f"{bn}.feed_forward.0.weight"], p[f"{bn}.feed_forward.0.bias"])
            h = linear(rt, gelu(rt, f), p[f"{bn}.feed_forward.2.weight"], p[f"{bn}.feed_forward.2.bias"], residual=h)
        y = conv1d_same(rt, h, w, b, 1, causal)
        return (y, xs) if skip else y

    @staticmethod
    def _crop_pair(a: torch.Tensor, b: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """crop (utils/module.py:186-204) on the length axis of channel-last tensors"""
        la, lb = a.shape[1], b.shape[1]
        if la == lb:
            return a, b
        if la > lb:
            d = la - lb
            return a[:, d // 2: la - (d - d // 2)], b
        d = lb - la
        return a, b[:, d // 2: lb - (d - d // 2)]

    def _cross_attention_names(self) -> List[str]:
        sp = self.spec
        trs = [d.transformer for d in sp.downs if d.transformer] + ([sp.bott_tr] if sp.bott_tr else []) + [u.transformer for u in sp.ups if u.transformer]
        return [f"{t.name}.blocks.{l}.cross_attention" for t in trs for l in range(t.num_layers)]

    def _stacked_context_kv(self, rows: Optional[torch.Tensor], b_eff: int):
        """{cross-attention name: (K | V window, KvSlot)} -- every layer's ``to_kv(norm_context(context))`` from ONE standardisation and ONE
        product over the context rows (ContextKVFn), or None when the pass keeps one LayerNorm + Linear per layer: float32 mode (the
        weight-gradient product is bf16), layer widths that are not 32-column multiples, no context"""
        rt, p = self.rt, self.p
        if not (rt.kv_grouped and rt.small_attn and rows is not None and rows.dim() == 3 and rows.dtype == torch.bfloat16 and rows.is_contiguous()):
            return None
        ex = self.exchange
        if ex is not None and ex.active:
            # the overlapped gradient exchange sends a top-level block's slice of the flat gradient as soon as the gradient of the block's
            # input exists (optim.GradExchange.region_ready); the stacked projection writes the norm_context / to_kv gradients of ALL blocks
            # at the END of the backward pass, after most of those slices have left.  The pass that carries the exchange (the last of an
            # accumulation window) therefore keeps one LayerNorm + Linear per layer; the others take the stacked form.
            return None
        names = self._cross_attention_names()
        if not names:
            return None
        ws = [p[f"{n}.to_kv.weight"] for n in names]
        K = rows.shape[-1]
        # (the data-gradient product runs jen1_big_gemm over K = sum of the widths: a multiple of 64 in bf16)
        if K % 64 or any(w.shape[1] != K or w.shape[0] % 32 for w in ws) or sum(w.shape[0] for w in ws) % 64:
            return None
        pairs = context_kv(rt, rows, ws, [p[f"{n}.norm_context.weight"] for n in names], [p[f"{n}.norm_context.bias"] for n in names], b_eff)
        return dict(zip(names, pairs))

    # ------------------------------------------------------------------ UNet1d.forward
    def unet(self, x: torch.Tensor, t: torch.Tensor, embedding: torch.Tensor, embedding_mask, ctx_channels, causal: bool) -> torch.Tensor:
        """x [B, C, T] float32 (+ ctx_channels [B, 129, T]) -> [B, out_channels, T] float32"""
        sp = self.spec
        if ctx_channels is not None:
            x = torch.cat([x, ctx_channels.to(x.dtype)], dim=1)
        h = self.unet_rows(self._to_rows(x), t, embedding, embedding_mask, causal)
        return h[:, :, :sp.out_channels].to(torch.float32).transpose(1, 2)

    def unet_rows(self, h: torch.Tensor, t: torch.Tensor, embedding: torch.Tensor, embedding_mask, causal) -> torch.Tensor:
        """the same on channel-last rows: h [B, T, pad8(C_in + C_ctx)] in the compute dtype -> [B, T, pad8(out_channels)]"""
        rt, p, sp = self.rt, self.p, self.spec
        mp = self.mapping(t)
        smap = silu(rt, mp).to(rt.tdtype)
        films = self.films(smap)
        h = self.res_block(sp.to_in, h, smap, False, films)          # Patcher / Unpatcher are never causal (blocks.py:256-259)
        # Every block output of the down path is a skip connection (blocks.py:641-643).  What goes on the skip list is not the output
        # itself but the ALIAS its next consumer hands back (res_block / transformer / the next level's down conv with skip=True): the
        # gradient that arrives through the alias from the up path is then added inside that consumer's backward kernel instead of
        # by an accumulation launch of autograd (~30 per pass).
        emb_box = list(embedding) if isinstance(embedding, (tuple, list)) else [embedding]     # (handed from one cross-attention to the next;
                                                                                             #  optionally with the K / V row map of the batch)
        self._kv = self._stacked_context_kv(emb_box[0], h.shape[0])
        skips_list: List = [[None]]
        slot = (skips_list[0], 0)                                       # where the alias of the current ``h`` belongs

        def put(alias):
            # fill once: the slot belongs to the ``h`` that was current when it was opened, and only that tensor's FIRST consumer hands
            # back its alias.  (A down level with neither blocks nor a transformer opens no slot: the next level's down conv must not
            # overwrite the previous level's skip with an alias of another tensor.)
            if slot[0][slot[1]] is None:
                slot[0][slot[1]] = alias
        for d in sp.downs:
            h = self._mark(h, d.name)
            h, al = conv1d_same(rt, h, p[f"{d.name}.downsample.conv.weight"], p[f"{d.name}.downsample.conv.bias"], d.factor, causal,
                                fork=True) if (rt.fork_skips and h.requires_grad) else \
                (conv1d_same(rt, h, p[f"{d.name}.downsample.conv.weight"], p[f"{d.name}.downsample.conv.bias"], d.factor, causal), h)
            put(al)
            skips: List = []
            first = True
            for r in d.blocks:
                if first:
                    h = self.res_block(r, h, smap, causal, films)
                else:
                    h, al = self.res_block(r, h, smap, causal, films, skip=True)
                    put(al)
                first = False
                skips.append(None)
                slot = (skips, len(skips) - 1)
            if d.transformer:
                if first:
                    h = self.transformer(d.transformer, h, emb_box, embedding_mask, causal)
                else:
                    h, al = self.transformer(d.transformer, h, emb_box, embedding_mask, causal, skip=True)
                    put(al)
                skips.append(None)
                slot = (skips, len(skips) - 1)
            skips_list.append(skips)
        h = self._mark(h, "bottleneck")
        h, al = self.res_block(sp.bott_pre, h, smap, causal, films, skip=True)
        put(al)
        if sp.bott_tr:
            h = self.transformer(sp.bott_tr, h, emb_box, embedding_mask, causal)
        h = self.res_block(sp.bott_post, h, smap, causal, films)
        for u in sp.ups:
            h = self._mark(h, u.name)
            skips = skips_list.pop()
            for r in u.blocks:
                a, sk = self._crop_pair(h, skips.pop())                 # blocks.py:732-734
                h = concat_scale(rt, a, sk, self.skip_scale)
                h = self.res_block(r, h, smap, causal, films)
            if u.transformer:
                h = self.transformer(u.transformer, h, emb_box, embedding_mask, causal)
            w, b = p[f"{u.name}.upsample.weight"], p[f"{u.name}.upsample.bias"]
            f = u.factor
            if f == 1:
                h = conv1d_zero_pad(rt, h, w, b, 1)
            else:
                h = conv_transpose1d(rt, h, w, b, f, f // 2 + f % 2, f % 2)
        h = h + skips_list.pop()[0]                                      # model.py:261
        h = self._mark(h, "to_out")
        self._kv = None
        return self.res_block(sp.to_out, h, smap, False, films)

    # ------------------------------------------------------------------ UNetCFG1d.forward
    def forward(self, x: torch.Tensor, time: torch.Tensor, *, embedding: torch.Tensor, embedding_mask: Optional[torch.Tensor] = None,
                embedding_scale: float = 1.0, embedding_mask_proba: float = 0.0, batch_cfg: bool = False, scale_cfg: bool = False,
                scale_phi: float = 0.7, features=None, channels_list: Optional[Sequence[torch.Tensor]] = None,
                causal: Optional[bool] = False, dropout_rows: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Same contract as the reference forward (model.py:299-376), differentiable."""
        assert features is None, "context_features is unused on the JEN-1 path"
        rt, p, sp = self.rt, self.p, self.spec
        self.rt.abandon_weight_grads()
        # a bool, or one flag per clip (tensor [B]): a pass that holds causal and non-causal clips side by side (CausalRows)
        rows = CausalRows(causal) if torch.is_tensor(causal) else None
        causal = rows if rows is not None else bool(causal)
        causal2 = rows.twice() if rows is not None else causal          # for the CFG pair stacked on the batch axis
        B = embedding.shape[0]
        dev = x.device
        emb = embedding.to(rt.tdtype)
        mask = embedding_mask
        if sp.use_xattn_time:
            tok = gelu(rt, self._time_features("to_time_embedding.0", time)).to(rt.tdtype)
            emb = torch.cat([emb, tok[:, None, :]], dim=1)
            if mask is not None:
                mask = torch.cat([mask.to(torch.float32), torch.ones((B, 1), device=dev)], dim=1)
        fixed = p["fixed_embedding.embedding.weight"][: emb.shape[1]].to(rt.tdtype)[None].expand(B, -1, -1)
        if embedding_mask_proba > 0.0:
            if dropout_rows is not None:
                rows = dropout_rows.to(torch.bool)
            elif embedding_mask_proba >= 1.0:
                rows = torch.ones(B, dtype=torch.bool, device=dev)
            else:   # rand_bool (utils/module.py:36-42)
                rows = torch.bernoulli(torch.full((B,), float(embedding_mask_proba), device=dev)).to(torch.bool)
            emb = torch.where(rows[:, None, None], fixed, emb)
        ctx = None
        if sp.ctx_ch0:
            assert channels_list is not None and channels_list[0] is not None, "Missing context"     # model.py:189
            ctx = channels_list[0]
        if embedding_scale != 1.0:
            if batch_cfg:
                out_all = self.unet(torch.cat([x, x], 0), torch.cat([time, time], 0), torch.cat([emb, fixed], 0).contiguous(),
                                    None if mask is None else torch.cat([mask, mask], 0),
                                    None if ctx is None else torch.cat([ctx, ctx], 0), causal2)
                out, out_masked = out_all[:B], out_all[B:]
            else:
                out = self.unet(x, time, emb.contiguous(), mask, ctx, causal)
                out_masked = self.unet(x, time, fixed.contiguous(), mask, ctx, causal)
            out_cfg = out_masked + (out - out_masked) * embedding_scale
            if scale_cfg:
                out_std = out.std(dim=1, keepdim=True)
                out_cfg_std = out_cfg.std(dim=1, keepdim=True)
                return scale_phi * (out_cfg * (out_std / out_cfg_std)) + (1 - scale_phi) * out_cfg
            return out_cfg
        return self.unet(x, time, emb.contiguous(), mask, ctx, causal)

    def diffusion_loss(self, gd, x_start: torch.Tensor, t: torch.Tensor, conditioning, noise: torch.Tensor, causal, dropout_rows=None):
        """``_diffusion_loss`` off the legacy default stream (see ``_off_default_stream``): ``training_loosses`` enters here directly, not
        through ``__call__``, and an eager pass with injected noise or ``use_graph=False`` (trainer.py) must not leave a null-stream
        backward behind for a later capture of the same parameters to trip over"""
        return self._off_default_stream(self._diffusion_loss, gd, x_start, t, conditioning, noise, causal, dropout_rows)

    def _diffusion_loss(self, gd, x_start: torch.Tensor, t: torch.Tensor, conditioning, noise: torch.Tensor, causal, dropout_rows=None):
        """``GaussianDiffusion.training_loosses`` (gdm.py:245-272) around this network with both ends fused: q_sample + concat + CFG
        pair + layout change in one launch, the context rows in one, the CFG combine + rescale + loss in one (and one each way
        back) -- per-sample losses [B].  None when the settings need the literal path (an unbatched CFG pair)."""
        rt, p, sp = self.rt, self.p, self.spec
        cfg = gd.embedding_scale != 1.0
        if not rt.fused_glue or (cfg and not gd.batch_cfg) or conditioning.get("global_cond") is not None or sp.out_channels > 256:
            return None
        self.rt.abandon_weight_grads()
        rows_c = CausalRows(causal) if torch.is_tensor(causal) else None
        causal = rows_c if rows_c is not None else bool(causal)
        nrep = 2 if cfg else 1
        if nrep == 2 and rows_c is not None:
            causal = rows_c.twice()
        B, C, T = x_start.shape
        dev = x_start.device
        x_start = x_start.to(torch.float32).contiguous()
        noise = noise.to(torch.float32).contiguous()
        ca = gd.sqrt_alphas_cumprod.to(dev)[t].to(torch.float32).contiguous()
        cb = gd.sqrt_one_minus_alphas_cumprod.to(dev)[t].to(torch.float32).contiguous()
        if gd.objective == "noise":
            ta, tb = rt.const(B, 1.0), rt.const(B, 0.0)
        elif gd.objective == "x0":
            ta, tb = rt.const(B, 0.0), rt.const(B, 1.0)
        elif gd.objective == "v":
            ta, tb = ca, -cb
        else:
            raise ValueError(f"unknown objective {gd.objective}")
        ctxc = conditioning["input_concat_cond"] if sp.ctx_ch0 else None
        if sp.ctx_ch0:
            assert ctxc is not None, "Missing context"                    # model.py:189
            ctxc = ctxc.to(torch.float32).contiguous()
        Cc = 0 if ctxc is None else ctxc.shape[1]
        ld = pad8(C + Cc)
        h = torch.empty((nrep * B, T, ld), dtype=rt.tdtype, device=dev)
        tgt = torch.empty((B, T, C), dtype=torch.float32, device=dev)
        L.check(rt.lib.jen1_train_pack_input(x_start.data_ptr(), noise.data_ptr(), ca.data_ptr(), cb.data_ptr(), None if ctxc is None else ctxc.data_ptr(),
                                             h.data_ptr(), B, C, Cc, T, ld, nrep, ta.data_ptr(), tb.data_ptr(), tgt.data_ptr(), rt.dt, rt.stream()),
                "jen1_train_pack_input")
        # context rows (model.py:315-337)
        emb = conditioning["cross_attn_cond"].to(torch.float32).contiguous()
        mask = conditioning["cross_attn_masks"]
        tok = gelu(rt, self._time_features("to_time_embedding.0", t)) if sp.use_xattn_time else None
        drop = None
        if gd.cfg_dropout_proba > 0.0:
            if dropout_rows is not None:
                drop = dropout_rows.to(torch.bool)
            elif gd.cfg_dropout_proba >= 1.0:
                drop = torch.ones(B, dtype=torch.bool, device=dev)
            else:   # rand_bool (utils/module.py:36-42)
                drop = torch.bernoulli(torch.full((B,), float(gd.cfg_dropout_proba), device=dev)).to(torch.bool)
        fixed = p["fixed_embedding.embedding.weight"]
        share = nrep == 2 and rt.share_fixed_context
        ctx_rows = ContextRowsFn.apply(tok, fixed, emb, drop, rt, 0 if share else nrep)
        if share:
            # the pair's unconditional half attends to the fixed embedding whatever the batch element (model.py:333): ONE row set, projected
            # once per layer and read by all B of them through a row map (B + 1 instead of 2 B context rows through LayerNorm and to_kv)
            ctx_rows = (ctx_rows, rt.kv_rows(B))
        if mask is not None:
            mask = mask.to(torch.float32)
            if sp.use_xattn_time:
                mask = torch.cat([mask, rt.const(B, 1.0)[:, None]], dim=1)
            if nrep == 2:
                mask = torch.cat([mask, mask], 0)
        t2 = torch.cat([t, t], 0) if nrep == 2 else t
        out = self.unet_rows(h, t2, ctx_rows, mask, causal)
        return CfgLossFn.apply(out.contiguous(), tgt, rt, B, sp.out_channels, nrep, float(gd.embedding_scale), bool(gd.scale_cfg), 0.7,
                               getattr(gd, "loss_type", "l2") == "l1")

    def _off_default_stream(self, fn, *args, **kwargs):
        """run ``fn`` (launches of a differentiable pass) on a private stream when the caller is on the legacy default stream.  A backward
        pass that ran on the null stream makes a later graph capture of the same parameters crash inside hipStreamEndCapture (ROCm 7.2 /
        torch 2.10, reproduced in tests/test_gpu_train.py), so the eager path never uses it; autograd replays each node on the stream
        its forward ran on, so moving the forward moves the backward."""
        dev = self.rt.device
        cur = torch.cuda.current_stream(dev)
        if torch.cuda.is_current_stream_capturing() or cur != torch.cuda.default_stream(dev):
            return fn(*args, **kwargs)
        if self._side is None:
            self._side = torch.cuda.Stream(dev)
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            out = fn(*args, **kwargs)
        cur.wait_stream(self._side)
        if torch.is_tensor(out):
            out.record_stream(cur)
        return out

    def __call__(self, *args, **kwargs) -> torch.Tensor:
        """``forward``, off the legacy default stream (``_off_default_stream``)"""
        return self._off_default_stream(self.forward, *args, **kwargs)


# =====================================================================================================================
# hipGraph-captured forward + backward
# =====================================================================================================================
class GraphedLossStep:
    """``(training_loosses(...) * scale).backward()`` captured once per (shape, causal) as a HIP graph.

    In eager mode a training micro-batch is ~6000 launches issued from Python at ~20 us each, i.e. host-bound; the
    replayed graph runs the same kernels back to back.  Inputs are copied into static buffers, the loss comes back in
    a static scalar, gradients are accumulated into ``param.grad`` exactly as in eager mode (so ``zero_grad`` /
    the optimiser step stay outside the graph).  Noise and the CFG-dropout rows are drawn inside the graph from
    torch's graph-safe generator, so every replay sees fresh draws.
    The scratch resets inside the kernels' entry points are kernel nodes: ``hipMemsetAsync`` nodes interleaved with
    kernel nodes were observed to replay out of order on ROCm 7.2.
    """

    def __init__(self, graph: TrainGraph, diffusion, scale: float = 1.0):
        self.graph, self.diffusion, self.scale = graph, diffusion, scale
        self._captured: Dict[tuple, tuple] = {}
        # an armed optim.GradExchange (the trainer sets it for the LAST pass of an accumulation window): with RCCL the per-region
        # all-reduces are recorded into a second variant of the captured pass -- TrainGraph's hooks fire while the backward is
        # being recorded, each region's collective becomes a graph node on the communication stream behind the kernels that
        # complete the region, and the replay overlaps the exchange with the rest of the backward pass exactly like the eager
        # hooks do (DDP, train.py:88-89).  Backends that cannot be recorded (gloo) get the blocking exchange behind the replay.
        self.exchange = None

    def _body(self, static, causal):
        cond = {"cross_attn_cond": static["emb"], "cross_attn_masks": static["mask"], "global_cond": None,
                "input_concat_cond": static["concat"]}
        if static["causal"] is not None:
            causal = static["causal"]          # one flag per clip, read by the replayed kernels: refreshed before every replay
        if static["w"] is None:
            loss = self.diffusion.training_loosses(self.graph, static["x0"], static["t"], cond, causal=causal)
            (loss * self.scale).backward()
            return loss.detach()
        # merged task sub-batches: the objective is sum_i w_i * loss_i (w_i = 1 / size of the sample's own sub-batch, i.e. the
        # sum of the per-task means of trainer.py:205-211); the per-sample losses come back for the per-task report
        per_sample = self.diffusion.training_loosses(self.graph, static["x0"], static["t"], cond, causal=causal, reduction="none")
        ((per_sample * static["w"]).sum() * self.scale).backward()
        return per_sample.detach()

    def _capture(self, key, x0, t, conditioning, causal, weights=None, exchange=None):
        params = list(self.graph.p.values())
        static = {"x0": x0.clone(), "t": t.clone(), "emb": conditioning["cross_attn_cond"].clone(),
                  "causal": causal.to(torch.int32).clone() if torch.is_tensor(causal) else None,
                  "mask": None if conditioning["cross_attn_masks"] is None else conditioning["cross_attn_masks"].clone(),
                  "concat": None if conditioning["input_concat_cond"] is None else conditioning["input_concat_cond"].clone(),
                  "w": None if weights is None else weights.to(torch.float32).clone()}
        keep = [None if p.grad is None else p.grad.clone() for p in params]       # the warm-up run must not leak into the gradients
        saved_ex, self.graph.exchange = self.graph.exchange, None                 # (no collectives in the warm-up run)
        side = torch.cuda.Stream(self.graph.rt.device)
        side.wait_stream(torch.cuda.current_stream(self.graph.rt.device))
        with torch.cuda.stream(side):
            self._body(static, causal)
        torch.cuda.current_stream(self.graph.rt.device).wait_stream(side)
        for p, k in zip(params, keep):
            if k is None:
                p.grad.zero_()
            else:
                p.grad.copy_(k)
        del keep
        self.graph.rt.refresh_all()        # packed weights are refreshed OUTSIDE the graph (once per optimiser step)
        g = torch.cuda.CUDAGraph()
        self.graph.exchange = exchange
        try:
            if exchange is not None:
                exchange.begin()
            # jen1_amd/graphs.py: the collector is kept out of the capture, and only THIS thread's calls are checked against it (with a
            # process group alive its watchdog thread queries events while this thread records)
            with capture_graph(g):
                loss = self._body(static, causal)
                if exchange is not None:
                    exchange.finish()      # recorded: the leftover regions, the join of the communication stream, the 1 / world scale
        finally:
            self.graph.exchange = saved_ex
        self._captured[key] = (g, static, loss)
        return self._captured[key]

    def __call__(self, x0: torch.Tensor, t: torch.Tensor, conditioning: Dict[str, Optional[torch.Tensor]], causal: bool,
                 sample_weights: Optional[torch.Tensor] = None) -> torch.Tensor:
        """replay (capture on first use) forward + backward of one pass; returns the loss, or with ``sample_weights`` [B] the
        per-sample losses [B] of the pass whose objective is their weighted sum"""
        assert conditioning.get("global_cond") is None
        ex = self.exchange if (self.exchange is not None and self.exchange.active and self.exchange.capturable) else None
        per_clip = torch.is_tensor(causal)
        key = (tuple(x0.shape), "per clip" if per_clip else bool(causal), conditioning["cross_attn_masks"] is None, conditioning["input_concat_cond"] is None,
               sample_weights is None, ex is not None)
        hit = self._captured.get(key)
        first = hit is None
        if first:
            hit = self._capture(key, x0, t, conditioning, causal if per_clip else bool(causal), sample_weights, ex)
            if ex is not None:
                ex.begin()                 # (recording the exchange consumed the armed state; the replay below is the real pass)
        g, static, loss = hit
        if sample_weights is not None:
            static["w"].copy_(sample_weights)
        static["x0"].copy_(x0)
        static["t"].copy_(t)
        if per_clip:
            static["causal"].copy_(causal)
        static["emb"].copy_(conditioning["cross_attn_cond"])
        if static["mask"] is not None:
            static["mask"].copy_(conditioning["cross_attn_masks"])
        if static["concat"] is not None:
            static["concat"].copy_(conditioning["input_concat_cond"])
        self.graph.rt.refresh_all()
        g.replay()
        if ex is not None:
            ex.done_in_graph()
        return loss.clone()
