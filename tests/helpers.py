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


# ---- measured parity numbers: every gate that matters writes what it measured -------------------------------------------------------
# ``pytest -q`` swallows prints, so the multi-step / gradient parity tests append one JSON object per case to gpurun_out/parity.jsonl
# (scratch on the GPU box, merged back by gpurun); tools/parity_summary.py turns it into profiles/rNN_parity.txt.  The bf16 gates in
# BF16_GATES are <= 2x the values measured on MI355X at the time they were set (profiles/r06_parity.txt).
PARITY_LOG = os.path.join(ROOT, "gpurun_out", "parity.jsonl")


def record_parity(test: str, case: str, mode: str, **metrics) -> None:
    import json
    try:
        os.makedirs(os.path.dirname(PARITY_LOG), exist_ok=True)
        with open(PARITY_LOG, "a") as f:
            f.write(json.dumps({"test": test, "case": case, "mode": mode, **{k: float(v) for k, v in metrics.items()}}) + "\n")
    except OSError:
        pass


# bf16 gates of the multi-step sampler and of the training pass against the reference's float32 outputs.  Keys: (test, case) -> {metric: gate}.
# Measured values: profiles/r06_parity.txt; every gate is <= 2x its measured value (bf16 storage, float32 accumulate; the reference's own
# bf16 autocast sits at 9.5e-3 max-abs / 8.5e-3 relative L2 per forward).
BF16_GATES = {
    # (test, case): {metric: gate}                                  measured on MI355X (worst of the runs in profiles/r06_parity.txt)
    ("full_ddim", "ddim10.B2.cfg"): {"l2": 4.5e-2},                 # 2.31e-2
    ("full_ddim", "ddim2.B8.cfg"): {"l2": 7.5e-2},                  # 3.81e-2
    ("full_ddim", "ddim2.B8.nocfg"): {"l2": 7.5e-2},                # 3.93e-2
    ("full_ddim", "ddim2.T9000.cont"): {"l2": 6.8e-2},              # 3.4e-2 - 3.7e-2
    ("full_ddim100", "ddim100.B2.cfg.step10"): {"l2": 1.55e-2},     # 7.9e-3 - 8.0e-3
    ("full_ddim100", "ddim100.B2.cfg.step25"): {"l2": 1.6e-2},      # 8.41e-3
    ("full_ddim100", "ddim100.B2.cfg.step50"): {"l2": 1.9e-2},      # 9.72e-3
    ("full_ddim100", "ddim100.B2.cfg.step75"): {"l2": 3.5e-2},      # 1.79e-2
    ("full_ddim100", "ddim100.B2.cfg"): {"l2": 4.5e-2},             # 2.34e-2
    ("full_ddim100", "ddim100.B8.nocfg.step10"): {"l2": 1.65e-2},   # 8.4e-3
    ("full_ddim100", "ddim100.B8.nocfg.step25"): {"l2": 1.5e-2},    # 7.8e-3
    ("full_ddim100", "ddim100.B8.nocfg.step50"): {"l2": 2.6e-2},    # 1.34e-2
    ("full_ddim100", "ddim100.B8.nocfg.step75"): {"l2": 7.8e-2},    # 4.0e-2
    ("full_ddim100", "ddim100.B8.nocfg"): {"l2": 1.0e-1},           # 6.47e-2  (100 steps, eta = 0, no CFG: the bench workload)
    ("configs3_micro_batch", "bf16"): {"loss": 1.2e-3, "norm": 1.1e-2, "samp": 0.74, "samp_rms": 6.0e-2},       # 5.9e-4 - 6.4e-4, 5.7e-3 - 6.3e-3, worst entry 0.24 - 0.50 (a tail statistic under float atomics: gate 1.5x the worst seen), rms 3.1e-2
}


def bf16_gate(test: str, case: str, metric: str) -> float:
    return BF16_GATES[(test, case)][metric]
