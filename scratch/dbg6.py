import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, 'jen-1-pytorch_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from helpers import rel_err
from jen1_amd import synth
from jen1_amd.config import full_model_config
from jen1_amd.model import UNetCFG1d
from jen1_amd.diffusion import GaussianDiffusion, get_beta_schedule
dev = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()
m = UNetCFG1d(**full_model_config(), compute_dtype=sys.argv[1] if len(sys.argv) > 1 else "f32", device="cuda")
B, T, S = 2, 1500, 4
cond = {k: dev(v) for k, v in synth.conditioning(B, T).items()}
betas, _ = get_beta_schedule("linear", 1000)
shape = (B, 128, T)
init = dev(synth.noise_list(1, shape, seed=7)[0])
noises = [dev(n) for n in synth.noise_list(S, shape, seed=11)]
outs = {}
for name, ug in (("eager1", False), ("eager2", False), ("graph1", True), ("graph2", True)):
    gd = GaussianDiffusion(steps=1000, betas=betas, objective="noise", loss_type="l2", device="cuda", cfg_dropout_proba=0.0,
                           embedding_scale=0.8, batch_cfg=True, scale_cfg=True, sampling_timesteps=S)
    y = gd.sample(m, shape, cond, init_noise=init, step_noises=noises, use_graph=ug)
    torch.cuda.synchronize()
    outs[name] = y.cpu().numpy()
    print(name, "finite", bool(torch.isfinite(y).all()), "absmax", float(y.abs().max()), flush=True)
for a, b in (("eager1", "eager2"), ("graph1", "graph2"), ("eager1", "graph1")):
    print(a, b, rel_err(outs[a], outs[b]))
# single forward repeatability
x = dev(synth.latents(B, T)); t = torch.tensor([999, 9], device="cuda")
kw = dict(embedding=cond["cross_attn_cond"], embedding_mask=cond["cross_attn_masks"], embedding_scale=0.8, batch_cfg=True, scale_cfg=True,
          channels_list=[cond["input_concat_cond"]])
ys = [m(x, t, **kw).cpu().numpy() for _ in range(4)]
print("forward repeat:", [rel_err(ys[i], ys[0]) for i in range(1, 4)])
