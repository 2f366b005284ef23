#!/usr/bin/env python3
"""SEANet decoder of Encodec 48 kHz on the HIP kernels, timed (synthetic weights by key name; the checkpoint is not
available offline).  Used under rocprofv3 for profiles/r01_encodec_decode_kernel_stats.txt.

    python tools/encodec_decode.py [--dtype bf16|f32] [--batch 8] [--frames 1500] [--reps 3]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "jen-1-pytorch_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from jen1_amd.encodec import SEANetDecoderHIP  # noqa: E402
from jen1_amd.init_fill import fill  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--frames", type=int, default=1500)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    sch = json.loads(str(np.load(os.path.join(ROOT, "tests", "golden", "encodec.npz"))["schema"]))
    dec = SEANetDecoderHIP({k: torch.from_numpy(fill("encodec.decoder." + k, tuple(s), 1234)) for k, s in sch}, compute_dtype=a.dtype)
    emb = torch.randn((a.batch, 128, a.frames), device="cuda")
    dec(emb)
    torch.cuda.synchronize()
    for r in range(a.reps):
        t0 = time.perf_counter()
        y = dec(emb)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"rep {r}: {tuple(emb.shape)} latents -> {tuple(y.shape)} samples in {dt * 1e3:.1f} ms "
              f"({a.batch * y.shape[-1] / 48000 / dt:.0f} audio-seconds per second)", flush=True)


if __name__ == "__main__":
    main()
