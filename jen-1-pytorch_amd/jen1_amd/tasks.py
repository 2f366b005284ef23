"""Task masks and conditioning hand-off around the denoiser.

Host-side mirror of
  * ``UnifiedMultiTaskTrainer.random_mask`` / ``get_conditioning``  (/root/reference/trainer.py:215-247, :249-278)
  * ``Jen1.get_mask`` / ``Jen1.get_conditioning``                    (/root/reference/generation.py:134-145, :152-192)
  * the tail of ``T5Conditioner.forward``: ``proj_out`` + mask       (/root/reference/jen1/conditioners.py:84-111)
with the same names, argument meaning and defects (the reference's ``random.randint`` receives float bounds;
``text_guided`` flips a coin for ``causal``).  The T5 encoder itself is out of scope: the tail takes its
``last_hidden_state``.  The projection runs on the HIP path (jen1_conv_gemm with the token mask as its row scale).
"""
from __future__ import annotations

import math
import random as _random
from typing import Dict, List, Optional, Sequence, Tuple

import torch


def random_mask(sequence: torch.Tensor, max_mask_length: int, task: str, rng=_random) -> Tuple[torch.Tensor, torch.Tensor, bool]:
    """trainer.py:215-247.  Returns (masked_sequence, mask[b,1,T] with 1 = keep, causal).  ``rng`` needs
    ``choices`` and ``randint`` (the ``random`` module by default, like the reference)."""
    b, _, sequence_length = sequence.size()
    t = task.lower()
    if t == "text_guided":
        mask = torch.zeros((1, 1, sequence_length))
        causal = rng.choices([True, False])[0]
    elif t == "music_inpaint":
        mask_length = _randint(rng, sequence_length * 0.2, sequence_length * 0.8)
        mask_start = _randint(rng, 0, sequence_length - mask_length)
        mask = torch.ones((1, 1, sequence_length))
        mask[:, :, mask_start:mask_start + mask_length] = 0
        causal = False
    elif t == "music_cont":
        mask_length = _randint(rng, sequence_length * 0.2, sequence_length * 0.8)
        mask = torch.ones((1, 1, sequence_length))
        mask[:, :, -mask_length:] = 0
        causal = True
    else:
        raise ValueError(f"unknown task {task!r}")
    mask = torch.cat([mask] * b, dim=0).to(sequence.device)
    return sequence * mask, mask, causal


def _randint(rng, lo, hi) -> int:
    """``random.randint`` with the reference's float bounds (trainer.py:225,235): Python < 3.12 accepts integral
    floats only, so the reference effectively needs T to be a multiple of 5; non-integral bounds are truncated here."""
    return rng.randint(int(lo), int(hi))


def get_mask(sample_size: int, start: float, end: float, batch_size: int, sample_rate: int = 48000) -> torch.Tensor:
    """generation.py:134-145: ones with [floor(start*sr), ceil(end*sr)) zeroed, replicated over the batch."""
    mask = torch.ones((1, 1, sample_size))
    mask[:, :, math.floor(start * sample_rate):math.ceil(end * sample_rate)] = 0
    return torch.cat([mask] * batch_size, dim=0)


def get_conditioning(cond: Dict[str, object], cross_attn_cond_ids: Sequence[str] = ("prompt",), global_cond_ids: Sequence[str] = (),
                     input_concat_ids: Sequence[str] = ("masked_input", "mask"), batch_size: Optional[int] = None) -> Dict[str, Optional[torch.Tensor]]:
    """trainer.py:249-278 (``batch_size`` None) and generation.py:152-192 (``batch_size`` given: input-concat entries
    are read as ``cond[key][0]`` and 2-D ones are expanded over the batch, as the reference does)."""
    cross_in = cross_masks = global_cond = concat = None
    if len(cross_attn_cond_ids) > 0:
        cross_in = torch.cat([cond[k][0] for k in cross_attn_cond_ids], dim=1)
        cross_masks = torch.cat([cond[k][1] for k in cross_attn_cond_ids], dim=1)
    if len(global_cond_ids) > 0:
        global_cond = torch.cat([cond[k][0] for k in global_cond_ids], dim=-1)
        if global_cond.dim() == 3:
            global_cond = global_cond.squeeze(1)
    if len(input_concat_ids) > 0:
        if batch_size is None:
            concat = torch.cat([cond[k] for k in input_concat_ids], dim=1)
        else:
            parts = []
            for k in input_concat_ids:
                t = cond[k][0]
                if t.dim() == 2:
                    t = t.unsqueeze(0).expand(batch_size, -1, -1)
                parts.append(t)
            concat = torch.cat(parts, dim=1)
    return {"cross_attn_cond": cross_in, "cross_attn_masks": cross_masks, "global_cond": global_cond, "input_concat_cond": concat}


class TextConditionerTail:
    """``embeddings = proj_out(last_hidden_state) * attention_mask[..., None]`` (conditioners.py:106-111) on the HIP
    path: one jen1_conv_gemm launch with the token mask as its row scale.  ``weight`` [out, in] / ``bias`` [out] are
    the reference's ``proj_out`` Linear parameters."""

    def __init__(self, weight: torch.Tensor, bias: torch.Tensor, dtype: str = "f32", device="cuda"):
        from . import lib as L
        from .engine import KernelCtx
        from .packing import pack_gemm_weight
        self.kc = KernelCtx(dtype, device)
        self.out_features, self.in_features = weight.shape
        assert self.in_features % 32 == 0 and self.out_features % 32 == 0
        dev = self.kc.device
        self.w = pack_gemm_weight(weight.detach().to(dev, torch.float32)[None], self.kc.tdtype)
        self.b = bias.detach().to(dev, torch.float32).contiguous()
        self._plans: Dict[Tuple[int, int], tuple] = {}

    def __call__(self, last_hidden_state: torch.Tensor, attention_mask: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        from .engine import Act, OpBuilder
        B, N, F = last_hidden_state.shape
        assert F == self.in_features
        key = (B, N)
        if key not in self._plans:
            dev, td = self.kc.device, self.kc.tdtype
            x = Act(torch.empty((B, N, F), dtype=td, device=dev), B, N, F, F)
            y = Act(torch.empty((B, N, self.out_features), dtype=torch.float32, device=dev), B, N, self.out_features, self.out_features)
            m = torch.empty((B * N,), dtype=torch.float32, device=dev)
            ob = OpBuilder(self.kc)
            ob.conv(ob.ops, src0=x, w=self.w, bias=self.b, out=y, row_scale=m, y_f32=True)
            ob.finalize_workspace()
            self._plans[key] = (ob, x, y, m)
        ob, x, y, m = self._plans[key]
        x.t.copy_(last_hidden_state)
        m.copy_(attention_mask.reshape(-1).to(torch.float32))
        ob.run()
        return y.t.clone(), attention_mask
