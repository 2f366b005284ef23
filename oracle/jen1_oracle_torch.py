"""CPU ORACLE, multi-threaded form -- TEST INFRASTRUCTURE ONLY, NOT A PRODUCT PATH.

The same restatement of ``UNetCFG1d.forward`` as ``oracle/jen1_oracle.py`` (the numpy file is the readable
specification and the one the golden fixtures pin line by line), written on torch's CPU tensor library so that it uses
every host core: this is what ``bench.py``'s ``cpu_baseline`` leg times on the GPU box ("the
build's CPU restatement ... ``torch.set_num_threads(N)``").  It is pinned too: ``tests/test_oracle_golden.py`` checks it
against the reference's own outputs (``tests/golden/full_bench.npz``, ``tiny_unet.npz``).

Only ``tests/`` and ``bench.py``'s ``cpu_baseline`` may import this module; nothing under ``jen-1-pytorch_amd/`` does.
Reference lines (paths relative to /root/reference) are cited per function; parameters use the reference's
``state_dict`` keys, tensors its layouts ([B, C, T] for conv / norm, [B, N, C] inside attention), float32 arithmetic.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F


def _same_conv(x, w, b, stride: int, causal: bool):
    """``_Conv1d`` (jen1/model/blocks.py:34-53): (k-1) zeros all-left when causal, else floor((k-1)/2) on each side."""
    k = w.shape[-1]
    pad = k - 1
    x = F.pad(x, (pad, 0)) if causal else F.pad(x, (pad // 2, pad // 2))
    return F.conv1d(x, w, b, stride=stride)


class TorchOracleUNetCFG1d:
    """UNetCFG1d / UNet1d (jen1/model/model.py:13-376) and its blocks (jen1/model/blocks.py) on torch CPU tensors."""

    def __init__(self, params: Dict[str, np.ndarray], *, channels: int, multipliers: Sequence[int], factors: Sequence[int],
                 num_blocks: Sequence[int], attentions: Sequence[int], attention_heads: int, resnet_groups: int = 8,
                 use_skip_scale: bool = True, use_xattn_time: bool = True, **_unused):
        self.p = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for k, v in params.items()}
        self.multipliers, self.factors, self.num_blocks = list(multipliers), list(factors), list(num_blocks)
        self.attentions, self.heads, self.groups = list(attentions), attention_heads, resnet_groups
        self.skip_scale = 2 ** -0.5 if use_skip_scale else 1.0
        self.use_xattn_time = use_xattn_time
        self.L = len(self.multipliers) - 1

    # ---- leaves
    def _time_features(self, prefix: str, t):
        """LearnedPositionalEmbedding + Linear (utils/module.py:58-79)."""
        w = self.p[f"{prefix}.0.weights"]
        x = t.to(torch.float32)[:, None]
        freqs = x * w[None, :] * 2 * math.pi
        f = torch.cat([x, freqs.sin(), freqs.cos()], dim=-1)
        return F.linear(f, self.p[f"{prefix}.1.weight"], self.p[f"{prefix}.1.bias"])

    def mapping(self, t):
        """UNet1d.get_mapping (model.py:204-223, :75-89)."""
        m = F.gelu(self._time_features("to_time.0", t))
        m = F.gelu(F.linear(m, self.p["to_mapping.0.weight"], self.p["to_mapping.0.bias"]))
        return F.gelu(F.linear(m, self.p["to_mapping.2.weight"], self.p["to_mapping.2.bias"]))

    def conv_block(self, n, x, groups, scale_shift, causal):
        """ConvBlock1d.forward (blocks.py:137-145)."""
        h = F.group_norm(x, groups, self.p[f"{n}.groupnorm.weight"], self.p[f"{n}.groupnorm.bias"], 1e-5)
        if scale_shift is not None:
            h = h * (scale_shift[0] + 1) + scale_shift[1]
        return _same_conv(F.silu(h), self.p[f"{n}.project.conv.weight"], self.p[f"{n}.project.conv.bias"], 1, causal)

    def resnet_block(self, n, x, mapping, groups, causal):
        """ResnetBlock1d.forward (blocks.py:219-231) + MappingToScaleShift (:161-165)."""
        h = self.conv_block(f"{n}.block1", x, groups, None, causal)
        ss = F.linear(F.silu(mapping), self.p[f"{n}.to_scale_shift.to_scale_shift.1.weight"],
                      self.p[f"{n}.to_scale_shift.to_scale_shift.1.bias"])[:, :, None]
        c = ss.shape[1] // 2
        h = self.conv_block(f"{n}.block2", h, groups, (ss[:, :c], ss[:, c:]), causal)
        if f"{n}.to_out.conv.weight" in self.p:
            return h + _same_conv(x, self.p[f"{n}.to_out.conv.weight"], self.p[f"{n}.to_out.conv.bias"], 1, causal)
        return h + x

    def attention(self, n, x, context, context_mask, causal):
        """Attention.forward + AttentionBase.forward, math path (blocks.py:415-437, 355-380): the padding mask multiplies K
        and V (it is not a -inf logit mask, :431-434)."""
        ctx = x if context is None else context
        C = x.shape[-1]
        xn = F.layer_norm(x, (C,), self.p[f"{n}.norm.weight"], self.p[f"{n}.norm.bias"])
        cn = F.layer_norm(ctx, (ctx.shape[-1],), self.p[f"{n}.norm_context.weight"], self.p[f"{n}.norm_context.bias"])
        q = F.linear(xn, self.p[f"{n}.to_q.weight"])
        kv = F.linear(cn, self.p[f"{n}.to_kv.weight"])
        mid = kv.shape[-1] // 2
        k, v = kv[..., :mid], kv[..., mid:]
        if context_mask is not None:
            m = context_mask.to(torch.float32)[:, :, None]
            k, v = k * m, v * m
        B, N, _ = q.shape
        M, h = k.shape[1], self.heads
        d = mid // h
        qh, kh, vh = (t.reshape(B, -1, h, d).transpose(1, 2) for t in (q, k, v))
        sim = qh @ kh.transpose(-1, -2) * d ** -0.5
        if causal:
            keep = ~torch.ones((N, M), dtype=torch.bool).triu(M - N + 1)          # causal_mask (blocks.py:315-319)
            sim = sim.masked_fill(~keep, -torch.finfo(torch.float32).max)
        out = (sim.softmax(dim=-1) @ vh).transpose(1, 2).reshape(B, N, mid)
        return F.linear(out, self.p[f"{n}.attention.to_out.weight"], self.p[f"{n}.attention.to_out.bias"])

    def transformer1d(self, n, x, layers, embedding, embedding_mask, causal):
        """Transformer1d.forward (blocks.py:528-537): GN(32, eps=1e-6), the SAME 1x1 conv before and after the blocks."""
        w, b = self.p[f"{n}.conv1d.conv.weight"], self.p[f"{n}.conv1d.conv.bias"]
        h = F.group_norm(x, 32, self.p[f"{n}.group_norm.weight"], self.p[f"{n}.group_norm.bias"], 1e-6)
        h = _same_conv(h, w, b, 1, causal).transpose(1, 2)
        for l in range(layers):
            bn = f"{n}.blocks.{l}"
            h = self.attention(f"{bn}.attention", h, None, None, causal) + h
            h = self.attention(f"{bn}.cross_attention", h, embedding, embedding_mask, False) + h
            f = F.gelu(F.linear(h, self.p[f"{bn}.feed_forward.0.weight"], self.p[f"{bn}.feed_forward.0.bias"]))
            h = F.linear(f, self.p[f"{bn}.feed_forward.2.weight"], self.p[f"{bn}.feed_forward.2.bias"]) + h
        return _same_conv(h.transpose(1, 2).contiguous(), w, b, 1, causal)

    @staticmethod
    def _crop(x1, x2):
        """``crop`` (utils/module.py:186-204): centre-crop the longer tensor along T."""
        d = x1.shape[-1] - x2.shape[-1]
        if d == 0:
            return x1, x2
        start = d // 2
        end = d - start
        if d > 0:
            return x1[:, :, start: x1.shape[-1] - end], x2
        return x1, x2[:, :, start: -end]

    def unet(self, x, t, embedding, embedding_mask, ctx_channels, causal):
        """UNet1d.forward (model.py:225-265)."""
        if ctx_channels is not None:
            x = torch.cat([x, ctx_channels], dim=1)
        mp, G = self.mapping(t), self.groups
        x = self.resnet_block("to_in.block", x, mp, 1, False)
        skips_list: List = [x]
        for i in range(self.L):
            n = f"downsamples.{i}"
            x = _same_conv(x, self.p[f"{n}.downsample.conv.weight"], self.p[f"{n}.downsample.conv.bias"], self.factors[i], causal)
            skips = []
            for j in range(self.num_blocks[i]):
                x = self.resnet_block(f"{n}.blocks.{j}", x, mp, G, causal)
                skips.append(x)
            if self.attentions[i]:
                x = self.transformer1d(f"{n}.transformer", x, self.attentions[i], embedding, embedding_mask, causal)
                skips.append(x)
            skips_list.append(skips)
        x = self.resnet_block("bottleneck.pre_block", x, mp, G, causal)
        if self.attentions[-1]:
            x = self.transformer1d("bottleneck.transformer", x, self.attentions[-1], embedding, embedding_mask, causal)
        x = self.resnet_block("bottleneck.post_block", x, mp, G, causal)
        for idx, i in enumerate(reversed(range(self.L))):
            n = f"upsamples.{idx}"
            skips = skips_list.pop()
            for j in range(self.num_blocks[i] + (1 if self.attentions[i] else 0)):
                xa, sk = self._crop(x, skips.pop())                               # blocks.py:732-734
                x = self.resnet_block(f"{n}.blocks.{j}", torch.cat([xa, sk * self.skip_scale], dim=1), mp, G, causal)
            if self.attentions[i]:
                x = self.transformer1d(f"{n}.transformer", x, self.attentions[i], embedding, embedding_mask, causal)
            f = self.factors[i]
            w, b = self.p[f"{n}.upsample.weight"], self.p[f"{n}.upsample.bias"]
            x = F.conv1d(x, w, b, padding=1) if f == 1 else F.conv_transpose1d(x, w, b, stride=f, padding=f // 2 + f % 2, output_padding=f % 2)
        x = x + skips_list.pop()                                                    # model.py:261
        return self.resnet_block("to_out.block", x, mp, 1, False)

    @torch.no_grad()
    def forward(self, x, time, *, embedding, embedding_mask=None, embedding_scale: float = 1.0, batch_cfg: bool = False,
                scale_cfg: bool = False, scale_phi: float = 0.7, channels_list=None, causal: bool = False):
        """UNetCFG1d.forward (model.py:299-376) with ``embedding_mask_proba = 0`` (the CFG-dropout branch is covered by the
        numpy oracle)."""
        T_ = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a))
        x, time, emb = T_(x).float(), T_(time), T_(embedding).float()
        mask = None if embedding_mask is None else T_(embedding_mask).float()
        ctx = None if not channels_list else T_(channels_list[0]).float()
        B = emb.shape[0]
        if self.use_xattn_time:
            tok = F.gelu(self._time_features("to_time_embedding.0", time))
            emb = torch.cat([emb, tok[:, None, :]], dim=1)
            if mask is not None:
                mask = torch.cat([mask, torch.ones((B, 1))], dim=1)
        fixed = self.p["fixed_embedding.embedding.weight"][None, : emb.shape[1]].expand_as(emb)
        if embedding_scale != 1.0:
            if batch_cfg:
                cat2 = lambda a: None if a is None else torch.cat([a, a], 0)
                out_all = self.unet(cat2(x), cat2(time), torch.cat([emb, fixed], 0), cat2(mask), cat2(ctx), causal)
                out, out_masked = out_all[:B], out_all[B:]
            else:
                out = self.unet(x, time, emb, mask, ctx, causal)
                out_masked = self.unet(x, time, fixed, mask, ctx, causal)
            out_cfg = out_masked + (out - out_masked) * embedding_scale
            if scale_cfg:
                out_cfg = scale_phi * (out_cfg * (out.std(dim=1, keepdim=True) / out_cfg.std(dim=1, keepdim=True))) + (1 - scale_phi) * out_cfg
            return out_cfg.numpy()
        return self.unet(x, time, emb, mask, ctx, causal).numpy()

    __call__ = forward
