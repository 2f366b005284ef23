import sys, os
os.environ["JEN1_DEBUG_SYNC"] = "1"
sys.path.insert(0, 'tests'); sys.path.insert(0, 'jen-1-pytorch_amd'); sys.path.insert(0, '.')
import numpy as np, torch
from jen1_amd import synth
from jen1_amd.config import tiny_model_config
from jen1_amd.model import UNetCFG1d
m = UNetCFG1d(**tiny_model_config(), compute_dtype="f32", device="cuda")
print("engine init"); m.engine(); print("engine ok", flush=True)
B, T = 2, 300
x, cond = synth.latents(B, T), synth.conditioning(B, T)
t = np.array([999, 499], dtype=np.int64)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
y = m(d(x), d(t), embedding=d(cond["cross_attn_cond"]), embedding_mask=d(cond["cross_attn_masks"]), embedding_scale=1.0,
      channels_list=[d(cond["input_concat_cond"])])
torch.cuda.synchronize(); print("done", y.shape, float(y.abs().max()))
