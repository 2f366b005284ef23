import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, 'jen-1-pytorch_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from helpers import golden, rel_err
from jen1_amd import synth
from jen1_amd.config import tiny_model_config
from jen1_amd.model import UNetCFG1d
from jen1_amd.diffusion import GaussianDiffusion, get_beta_schedule
dev = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()
m = UNetCFG1d(**tiny_model_config(), compute_dtype="f32", device="cuda")
g = golden("tiny_sampler")
B, T, S = 2, 300, 10
cond = {k: dev(v) for k, v in synth.conditioning(B, T).items()}
shape = (B, 128, T)
init = dev(synth.noise_list(1, shape, seed=7)[0])
noises = [dev(n) for n in synth.noise_list(S, shape, seed=11)]
betas, _ = get_beta_schedule("linear", 1000)
def run(proba, scale, bcfg, rcfg, causal, drops=None, objective="noise", use_graph=True):
    print("run", proba, scale, bcfg, rcfg, causal, objective, use_graph, flush=True)
    gd = GaussianDiffusion(steps=1000, betas=betas, objective=objective, loss_type="l2", device="cuda", cfg_dropout_proba=proba,
                           embedding_scale=scale, batch_cfg=bcfg, scale_cfg=rcfg, sampling_timesteps=S)
    y = gd.sample(m, shape, cond, causal=causal, init_noise=init, step_noises=noises, dropout_rows=drops, use_graph=use_graph)
    torch.cuda.synchronize()
    return y.cpu().numpy()
for ug in (False, True):
    print(rel_err(run(0.0, 0.8, True, True, False, use_graph=ug), g["ddim10.cfg"]), flush=True)
    print(rel_err(run(0.0, 1.0, False, False, True, use_graph=ug)[:, :, ::3], g["ddim10.nocfg.causal"]), flush=True)
    print(rel_err(run(0.2, 0.8, True, True, False, drops=g["ddim10.dropout.rows"], use_graph=ug)[:, :, ::3], g["ddim10.dropout"]), flush=True)
    print(rel_err(run(0.0, 0.8, True, True, False, objective="x0", use_graph=ug)[:, :, ::3], g["ddim10.x0"]), flush=True)
    print(rel_err(run(0.0, 0.8, True, True, False, objective="v", use_graph=ug)[:, :, ::3], g["ddim10.v"]), flush=True)
