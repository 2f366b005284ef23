"""Synthetic inputs of the shapes BASELINE.json names.

Everything is numpy and comes from ``init_fill`` so the same arrays exist in
the build container (golden generation against the reference) and on the GPU
box.  No datasets, checkpoints or Encodec weights are available offline.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np

from .init_fill import fill_normal, fill_uniform


def latents(B: int, T: int, C: int = 128, seed: int = 0, key: str = "x") -> np.ndarray:
    """x ~ N(0,1) float32 [B, C, T]."""
    return fill_normal(f"synth.{key}.{B}.{C}.{T}", (B, C, T), seed)


def text_mask(B: int, N: int = 128) -> np.ndarray:
    """row i keeps its first min(N, 16*(i+1)) tokens (bool [B, N])."""
    m = np.zeros((B, N), dtype=bool)
    for i in range(B):
        m[i, : min(N, 16 * (i + 1))] = True
    return m


def text_embedding(B: int, N: int = 128, F: int = 1024, seed: int = 0) -> np.ndarray:
    """T5Conditioner output contract: N(0,1) features zeroed where the mask is False
    (reference jen1/conditioners.py:107-111)."""
    e = fill_normal(f"synth.emb.{B}.{N}.{F}", (B, N, F), seed)
    return (e * text_mask(B, N)[:, :, None]).astype(np.float32)


def concat_cond(B: int, T: int, task: str = "text_guided", C: int = 128, seed: int = 0) -> np.ndarray:
    """input_concat_cond [B, C+1, T] = cat([x*mask, mask]) (reference trainer.py:271).
    text_guided: all-zero mask => all-zero tensor (generation.py:98,118).
    music_inpaint / music_cont: deterministic spans of 40% of T."""
    if task == "text_guided":
        return np.zeros((B, C + 1, T), dtype=np.float32)
    mask = np.ones((1, 1, T), dtype=np.float32)
    n = int(0.4 * T)
    if task == "music_inpaint":
        s = int(0.3 * T)
        mask[:, :, s: s + n] = 0
    elif task == "music_cont":
        mask[:, :, T - n:] = 0
    else:
        raise ValueError(task)
    src = fill_normal(f"synth.clip.{B}.{C}.{T}", (B, C, T), seed)
    return np.concatenate([src * mask, np.broadcast_to(mask, (B, 1, T))], axis=1).astype(np.float32)


def conditioning(B: int, T: int, task: str = "text_guided", seed: int = 0) -> Dict[str, Optional[np.ndarray]]:
    """The dict GaussianDiffusion consumes (reference generation.py:187-192)."""
    return {
        "cross_attn_cond": text_embedding(B, seed=seed),
        "cross_attn_masks": text_mask(B),
        "global_cond": None,
        "input_concat_cond": concat_cond(B, T, task, seed=seed),
    }


def noise_list(n: int, shape, seed: int = 0, uniform: bool = False):
    """Pre-drawn per-step noises (the reference draws randn_like / rand_like inside
    the loop, gdm.py:161,218; parity runs inject them)."""
    if uniform:
        return [fill_uniform(f"synth.noise.{i}", shape, seed, 0.0, 1.0) for i in range(n)]
    return [fill_normal(f"synth.noise.{i}", shape, seed) for i in range(n)]


TRAIN8_TASKS = (("text_guided", 3, False), ("music_inpaint", 3, False), ("music_cont", 2, True))


def train8_inputs(T: int = 1500):
    """BASELINE configs[3] per-GPU shape: 8 clips as the 3 / 3 / 2 task sub-batches of one micro-batch (reference
    trainer.py:183-213), every sub-batch with its own text conditioning, its task's mask applied to ITS clips
    (trainer.py:197-203), timesteps and uniform noise (gdm.py:247).  -> [(task, x0, t, conditioning, noise, causal)];
    shared by the golden generator (tests/golden/make_golden.py fulltrain8) and the tests."""
    x_all = latents(8, T, key="clip8")
    tt = np.array([17, 801, 417, 999, 0, 250, 640, 93], dtype=np.int64)
    parts, o = [], 0
    for task, b, causal in TRAIN8_TASKS:
        x0 = x_all[o:o + b]
        cond = conditioning(b, T, task)
        keep = cond["input_concat_cond"][:, 128:129] if task != "text_guided" else np.zeros((b, 1, T), dtype=np.float32)
        cond["input_concat_cond"] = np.concatenate([x0 * keep, np.broadcast_to(keep, (b, 1, T))], axis=1).astype(np.float32)
        noise = fill_uniform(f"synth.trainnoise8.{task}", (b, 128, T), 3, 0.0, 1.0)
        parts.append((task, x0, tt[o:o + b], cond, noise, causal))
        o += b
    return parts
