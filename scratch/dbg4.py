import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, 'jen-1-pytorch_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from jen1_amd import synth
from jen1_amd.config import tiny_model_config
from jen1_amd.model import UNetCFG1d
from helpers import golden, rel_err
m = UNetCFG1d(**tiny_model_config(), compute_dtype=sys.argv[1] if len(sys.argv) > 1 else "f32", device="cuda")
B, T = 2, 300
x, cond = synth.latents(B, T), synth.conditioning(B, T)
t = np.array([999, 499], dtype=np.int64)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
g = golden("tiny_unet")
for key in [k for k in g.files if k.startswith("y.s")]:
    scale_s, rest = key[3:].split(".b")
    b, r, c = rest[0] == "1", rest[3] == "1", rest[6] == "1"
    print("case", key, flush=True)
    y = m(d(x), d(t), embedding=d(cond["cross_attn_cond"]), embedding_mask=d(cond["cross_attn_masks"]), embedding_scale=float(scale_s),
          batch_cfg=b, scale_cfg=r, causal=c, channels_list=[d(cond["input_concat_cond"])])
    torch.cuda.synchronize()
    print("   err", rel_err(y.cpu().numpy()[:, :, ::3], g[key]), flush=True)
