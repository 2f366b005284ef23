"""Shared helpers for the parity tests (inputs/weights are regenerated, never stored)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jen-1-pytorch_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from jen1_amd.init_fill import fill  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
SEED = 1234


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def filled(shapes, prefix="", seed=SEED):
    """{key: array} for (key, shape) pairs; ``prefix`` is prepended for the fill key only."""
    return {k: fill(prefix + k, s, seed) for k, s in shapes}


def rel_err(a, b):
    """max-abs(a-b) / max-abs(b): the parity metric BASELINE.json states (<= 1e-3 in fp32)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
