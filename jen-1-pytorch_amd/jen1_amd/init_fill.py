"""Deterministic, torch-independent weight / input filler.

No weights are ever committed to this repo.  Every test, fixture and bench run
materialises tensors from ``fill(key, shape, seed)``: a counter-based Philox
stream keyed by (crc32(key), seed), so the same named tensor is bit-identical in
the build container (where the reference is imported to make golden vectors)
and on the GPU box (where only this package exists).

The value distribution mirrors PyTorch's default module init closely enough
that activations stay O(1) through the ~300 layers of the full UNet
(reference modules: nn.Conv1d / nn.Linear -> U(-1/sqrt(fan_in), 1/sqrt(fan_in));
nn.Embedding and LearnedPositionalEmbedding.weights -> N(0,1),
/root/reference/utils/module.py:24,65).  Norm affine parameters are perturbed
away from (1, 0) on purpose so a kernel that drops gamma/beta fails parity.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np

__all__ = ["fill", "fill_normal", "fill_uniform", "fill_state_dict"]


def _gen(key: str, seed: int) -> np.random.Generator:
    k = zlib.crc32(key.encode("utf-8")) & 0xFFFFFFFF
    return np.random.Generator(np.random.Philox(key=[(seed << 32) | k, 0x6A656E31]))


def fill_normal(key: str, shape, seed: int = 0, std: float = 1.0) -> np.ndarray:
    g = _gen(key, seed)
    return (g.standard_normal(size=tuple(shape), dtype=np.float64) * std).astype(np.float32)


def fill_uniform(key: str, shape, seed: int = 0, lo: float = -1.0, hi: float = 1.0) -> np.ndarray:
    g = _gen(key, seed)
    return g.uniform(lo, hi, size=tuple(shape)).astype(np.float32)


def fill(key: str, shape: Tuple[int, ...], seed: int = 1234) -> np.ndarray:
    """Parameter value for a reference ``state_dict`` key."""
    shape = tuple(int(s) for s in shape)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "weights":                       # LearnedPositionalEmbedding
        return fill_normal(key, shape, seed)
    if "fixed_embedding" in key:                # nn.Embedding
        return fill_normal(key, shape, seed)
    is_norm = ("groupnorm" in key or "group_norm" in key or ".norm" in key)
    if is_norm:
        if leaf == "weight":
            return (1.0 + 0.2 * fill_uniform(key, shape, seed)).astype(np.float32)
        return (0.1 * fill_uniform(key, shape, seed)).astype(np.float32)
    if leaf == "weight":
        if key.endswith("upsample.weight") and len(shape) == 3 and "downsample" not in key:
            # ConvTranspose1d weight is [C_in, C_out, k]; torch fan_in = C_out * k.
            # The f=1 level uses a plain Conv1d [C_out, C_in, 3]; fan_in = C_in*k.
            # Both reduce to shape[1]*shape[2].
            fan_in = shape[1] * shape[2]
        else:
            fan_in = int(np.prod(shape[1:]))
        bound = 1.0 / np.sqrt(max(fan_in, 1))
        return fill_uniform(key, shape, seed, -bound, bound)
    if leaf == "bias":
        # bias bound needs fan_in of the sibling weight; a fixed small bound is
        # enough for parity purposes and keeps this function shape-local.
        return fill_uniform(key, shape, seed, -0.05, 0.05)
    return fill_uniform(key, shape, seed, -0.05, 0.05)


def fill_state_dict(shapes: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = 1234) -> Dict[str, np.ndarray]:
    return {k: fill(k, s, seed) for k, s in shapes}
