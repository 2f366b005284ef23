"""Root cause of the round-4 SIGABRT of `pytest -m gpu` (GPUTEST_r04.json): Python's cyclic garbage collector running INSIDE a hipGraph
capture.  Each variant leaves one kind of dead-but-uncollected object behind (a reference cycle), starts a capture, and forces
``gc.collect()`` in the middle of it.  Run every variant in its own process:  python tools/repro_gc_capture.py all"""
import ctypes
import gc
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "jen-1-pytorch_amd")]
VARIANTS = ("none", "tensor", "event", "stream", "graph", "pinned", "stepper")


class Cycle:
    def __init__(self, payload):
        self.payload, self.me = payload, self


def variant(name: str, mode: str) -> None:
    import torch
    so = os.path.join(ROOT, "tools", "native", "libabort_trace.so")
    if os.path.exists(so):
        ctypes.CDLL(so).jen1_abort_trace_install(b"")
    dev = "cuda:0"
    a = torch.randn(1 << 20, device=dev)
    gc.collect()
    gc.disable()
    if name == "tensor":
        Cycle(torch.randn(1 << 24, device=dev))
    elif name == "event":
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        Cycle(e)
    elif name == "stream":
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            a.mul_(2)
        Cycle(s)
    elif name == "graph":
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            a.mul_(2)
        g.replay()
        Cycle(g)
    elif name == "pinned":
        h = torch.empty(1 << 20, pin_memory=True)
        a.copy_(h, non_blocking=True)
        Cycle(h)
    elif name == "stepper":
        from jen1_amd import synth
        from jen1_amd.config import tiny_model_config
        from jen1_amd.diffusion import GaussianDiffusion, get_beta_schedule
        from jen1_amd.model import UNetCFG1d
        m = UNetCFG1d(**tiny_model_config(), init_seed=1234, compute_dtype="f32", device=dev)
        betas, _ = get_beta_schedule("linear", 1000)
        gd = GaussianDiffusion(steps=1000, betas=betas, objective="noise", loss_type="l2", device=dev, cfg_dropout_proba=0.0,
                               embedding_scale=0.8, batch_cfg=True, scale_cfg=True, sampling_timesteps=4)
        cond = {k: (None if v is None else torch.from_numpy(v).to(dev)) for k, v in synth.conditioning(2, 300).items()}
        gd.sample(m, (2, 128, 300), cond)
        torch.cuda.synchronize()
        del m, gd, cond
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    print(f"[{name}/{mode}] capture begins", flush=True)
    with torch.cuda.graph(g2, capture_error_mode=mode):
        a.mul_(2)
        n = gc.collect()
        a.add_(1)
    g2.replay()
    torch.cuda.synchronize()
    print(f"[{name}/{mode}] survived ({n} objects collected inside the capture)", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "all":
        for mode in ("global", "thread_local", "relaxed"):
            for v in VARIANTS:
                r = subprocess.run([sys.executable, __file__, v, mode], capture_output=True, text=True, timeout=300)
                tail = (r.stdout + r.stderr).strip().splitlines()
                keep = [l for l in tail if "jen1" in l or "survived" in l or "Error" in l or "error" in l or "hip" in l.lower()][:40]
                print(f"== {v}/{mode}: rc {r.returncode}\n   " + "\n   ".join(keep), flush=True)
    else:
        variant(sys.argv[1], sys.argv[2])
