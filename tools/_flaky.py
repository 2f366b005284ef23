import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "jen-1-pytorch_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from jen1_amd import synth
from jen1_amd.config import UNetSpec, tiny_model_config
from jen1_amd.init_fill import fill
from jen1_amd.model import UNetCFG1d
from oracle import jen1_oracle as O
cfg = tiny_model_config()
onet = O.OracleUNetCFG1d({k: fill(k, s, 1234) for k, s in UNetSpec(**cfg).param_shapes()}, **cfg)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
bad = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for it in range(N):
    m = UNetCFG1d(**cfg, compute_dtype="f32", device="cuda")
    for B, T in ((1, 64), (3, 95), (4, 90)):
        x, cond = synth.latents(B, T), synth.conditioning(B, T, "music_inpaint")
        t = np.array([(37 * i + 5) % 1000 for i in range(B)], dtype=np.int64)
        if it == 0:
            globals().setdefault("refs", {})[(B, T)] = onet(x, t, embedding=cond["cross_attn_cond"], embedding_mask=cond["cross_attn_masks"], embedding_scale=0.8,
                                                              batch_cfg=True, scale_cfg=True, channels_list=[cond["input_concat_cond"]], causal=False)
        ref = refs[(B, T)]
        for rep in range(3):
            y = m(dev(x), dev(t), embedding=dev(cond["cross_attn_cond"]), embedding_mask=dev(cond["cross_attn_masks"]), embedding_scale=0.8,
                  batch_cfg=True, scale_cfg=True, channels_list=[dev(cond["input_concat_cond"])], causal=False)
            torch.cuda.synchronize()
            e = float(np.abs(y.cpu().numpy() - ref).max() / np.abs(ref).max())
            if e > 1e-3:
                bad += 1
                print("BAD", it, B, T, rep, e, flush=True)
    del m
    # churn the allocator like other tests do
    junk = [torch.randn((np.random.randint(1, 64), 1024, 37), device="cuda") for _ in range(8)]
    del junk
print("bad", bad, "of", N * 9)
