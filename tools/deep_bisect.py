import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/jen-1-pytorch_amd")
import numpy as np, torch
from jen1_amd import synth
from jen1_amd.config import full_model_config
from jen1_amd.model import UNetCFG1d
m = UNetCFG1d(**full_model_config(), init_seed=1234, compute_dtype=sys.argv[1], device="cuda")
B, T, nrep = int(sys.argv[2]), 1500, int(sys.argv[3])
plan = m.engine().plan(B, T, nrep, False, deep=True)
print("deep level", plan.deep_level, len(plan.deep), "phases; limit", os.environ.get("JEN1_DEEP_RUN_PHASES"), flush=True)
if os.environ.get("LABELS"):
    for i, l in enumerate(plan.deep.labels): print(i, l)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
x, cond = synth.latents(B, T), synth.conditioning(B, T)
t = np.array([(131 * i + 7) % 1000 for i in range(B)], dtype=np.int64)
m._prepare(plan, dev(x), dev(t), dev(cond["cross_attn_cond"]), dev(cond["cross_attn_masks"]), [dev(cond["input_concat_cond"])], None)
plan.run(torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
print("ok, err", plan.deep.error(), flush=True)
