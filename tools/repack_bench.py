#!/usr/bin/env python3
"""jen1_repack (every compute copy of the 296.5 M parameters after an optimiser step) timed alone, graph-replayed.

    python tools/repack_bench.py            prints us per launch and the bytes it moves"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jen-1-pytorch_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from jen1_amd import graphs, synth  # noqa: E402
from jen1_amd.config import full_model_config  # noqa: E402
from jen1_amd.diffusion import GaussianDiffusion, get_beta_schedule  # noqa: E402
from jen1_amd.model import UNetCFG1d  # noqa: E402


def main():
    dev = lambda v: None if v is None else torch.from_numpy(np.ascontiguousarray(v)).cuda()   # noqa: E731
    model = UNetCFG1d(**full_model_config(), init_seed=1234, compute_dtype="bf16", device="cuda")
    model.train()
    graph = model.train_graph("bf16")
    betas, _ = get_beta_schedule("linear", 1000)
    gd = GaussianDiffusion(steps=1000, betas=betas, objective="noise", loss_type="l2", device="cuda", cfg_dropout_proba=0.2,
                           embedding_scale=0.8, batch_cfg=True, scale_cfg=True)
    B, T = 2, 300
    x0 = dev(synth.latents(B, T, key="clip"))
    cond = {k: dev(v) for k, v in synth.conditioning(B, T, "music_inpaint").items()}
    t = torch.randint(0, 1000, (B,), device="cuda")
    gd.training_loosses(graph, x0, t, cond, causal=False).backward()      # registers every compute copy
    rt = graph.rt
    torch.cuda.synchronize()
    src = sum(w.numel() * 4 for hit in rt._packed.values() if hit[2] != -1 and not hit[3].endswith("D") for w in [hit[0]()] if w is not None)
    dst = sum(hit[1].numel() * hit[1].element_size() for hit in rt._packed.values() if hit[2] != -1 and hit[0]() is not None)
    for tag in ("one 32 x 32 tile per block",):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for _ in range(3):
            rt.invalidate()
            rt.refresh_all()
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(20):
            rt.invalidate()
            rt.refresh_all()
        ev[1].record()
        torch.cuda.synchronize()
        us = ev[0].elapsed_time(ev[1]) * 1000 / 20
        print(f"{tag:28s} {us:8.1f} us per refresh (incl. the fold / bias launches)   parameters read {src / 1e6:.0f} MB, copies written {dst / 1e6:.0f} MB"
              f" -> {(src + dst) / us / 1e6:.2f} TB/s")


if __name__ == "__main__":
    main()
