"""Replayed-pass time of the training path (bench.train_step_bench alone, without the sampling bench around it).

    python tools/train_pass_time.py [--dtype bf16] [--batch 8] [--frames 1500]

Prints the train_step dictionary bench.py puts under extra.train_step.  Environment switches of jen1_amd.train (JEN1_TRAIN_*) apply.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "jen-1-pytorch_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--frames", type=int, default=1500)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import torch
    import bench
    from jen1_amd.config import full_model_config
    cfg = full_model_config()
    out = bench.train_step_bench(cfg, args.batch, args.frames, args.dtype, torch.device("cuda:0"), reps=args.reps)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
