#!/usr/bin/env python3
"""One training micro-batch of the full denoiser on the HIP training path, eager or as a replayed HIP graph.

    python tools/train_step.py [--dtype bf16|f32] [--batch 8] [--length 1500] [--iters 5] [--tiny] [--eager]

Prints per-iteration forward+backward and optimiser times; used under rocprofv3 for profiles/r01_train_step_*.txt.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jen-1-pytorch_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from jen1_amd import synth  # noqa: E402
from jen1_amd.config import full_model_config, tiny_model_config  # noqa: E402
from jen1_amd.diffusion import GaussianDiffusion, get_beta_schedule  # noqa: E402
from jen1_amd.model import UNetCFG1d  # noqa: E402
from jen1_amd.optim import FusedAdamW  # noqa: E402
from jen1_amd.train import GraphedLossStep  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--length", type=int, default=1500)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--tiny", action="store_true")
    ap.add_argument("--eager", action="store_true")
    a = ap.parse_args()
    dev = lambda v: None if v is None else torch.from_numpy(np.ascontiguousarray(v)).cuda()   # noqa: E731
    model = UNetCFG1d(**(tiny_model_config() if a.tiny else full_model_config()), init_seed=1234, compute_dtype=a.dtype, device="cuda")
    model.train()
    opt = FusedAdamW(model.parameters())
    graph = model.train_graph(a.dtype)
    graph.attach_optimizer(opt)
    betas, _ = get_beta_schedule("linear", 1000)
    gd = GaussianDiffusion(steps=1000, betas=betas, objective="noise", loss_type="l2", device="cuda", cfg_dropout_proba=0.2,
                           embedding_scale=0.8, batch_cfg=True, scale_cfg=True)
    B, T = a.batch, a.length
    x0 = dev(synth.latents(B, T, key="clip"))
    cond = {k: dev(v) for k, v in synth.conditioning(B, T, "music_inpaint").items()}
    t = torch.randint(0, 1000, (B,), device="cuda")
    step = None if a.eager else GraphedLossStep(graph, gd)
    for it in range(a.iters + 1):
        torch.cuda.synchronize()
        t0 = time.time()
        opt.zero_grad()
        if step is None:
            loss = gd.training_loosses(graph, x0, t, cond, causal=False)
            loss.backward()
            loss = loss.detach()
        else:
            loss = step(x0, t, cond, False)
        torch.cuda.synchronize()
        t1 = time.time()
        opt.step()
        torch.cuda.synchronize()
        t2 = time.time()
        tag = "(includes capture)" if it == 0 and step is not None else ""
        print(f"iter {it}: loss {float(loss):.5f}  fwd+bwd {1e3 * (t1 - t0):.1f} ms  optimiser {1e3 * (t2 - t1):.1f} ms  "
              f"grad norm {float(opt.grad_norm()):.4f}  peak {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB {tag}", flush=True)


if __name__ == "__main__":
    main()
