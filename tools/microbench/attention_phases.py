import sys, os, ctypes
os.environ["JEN1_LIB"] = os.path.abspath("jen-1-pytorch_amd/jen1_amd/libjen1_prof.so")
sys.path.insert(0, 'tests'); sys.path.insert(0, 'jen-1-pytorch_amd')
import torch
from jen1_amd import lib as L
from jen1_amd.engine import OpBuilder, KernelCtx, Act
kc = KernelCtx("bf16")
lib = L.load()
lib.jen1_debug_set_attention_buffer.argtypes = [ctypes.c_void_p]
dbg = torch.zeros(16, dtype=torch.int64, device="cuda")
assert lib.jen1_debug_set_attention_buffer(dbg.data_ptr()) == 0
names = ["entry", "indices known", "all loads issued", "K,Q in LDS", "scores done", "V^T in LDS", "softmax done", "PV stored"]
order = [0, 1, 2, 3, 4, 7, 5, 6]
def prof(B, H, d, Nq, Nk, cross, label):
    mid = H * d
    ob = OpBuilder(kc)
    q = Act(torch.randn(B, Nq, mid, device="cuda").to(torch.bfloat16), B, Nq, mid, mid)
    out = Act(torch.zeros(B, Nq, mid, device="cuda", dtype=torch.bfloat16), B, Nq, mid, mid)
    if cross:
        kv = torch.randn(2 * B, Nk, 2 * mid, device="cuda").to(torch.bfloat16)
        kv_row = torch.arange(B, dtype=torch.int32, device="cuda")
        ext = torch.randn(100, 4 * mid, device="cuda").to(torch.bfloat16)
        extra_row = torch.arange(B, dtype=torch.int32, device="cuda")
        step = torch.zeros(1, dtype=torch.int32, device="cuda")
        ob.attention(ob.ops, q=q, q_off=0, kv_t=kv, ldkv=2 * mid, k_off=0, v_off=mid, out=out, H=H, d=d, Nk=Nk, causal=False, kv_row=kv_row,
                     kv_extra=ext, extra_row=extra_row, ld_extra=4 * mid, kx_off=0, vx_off=mid, extra_step=step)
    else:
        kv = torch.randn(B, Nk, 3 * mid, device="cuda").to(torch.bfloat16)
        ob.attention(ob.ops, q=Act(kv, B, Nq, 3 * mid, 3 * mid), q_off=0, kv_t=kv, ldkv=3 * mid, k_off=mid, v_off=2 * mid, out=out, H=H, d=d, Nk=Nk, causal=False)
    rs = []
    for it in range(5):
        dbg.zero_(); ob.run(); torch.cuda.synchronize()
        st = dbg.cpu().tolist()
        rs.append([(st[i] - st[0]) / 100.0 for i in order])
    med = [sorted(r[i] for r in rs)[2] for i in range(8)]
    print(label)
    pv = 0
    for n, v in zip(names, med):
        print(f"    {n:22s} t={v:6.2f} (+{v - pv:5.2f})"); pv = v
prof(8, 8, 64, 1, 129, True, "cross-attention L=1 Nk=129 d=64")
prof(8, 8, 64, 24, 129, True, "cross-attention L=24 Nk=129 d=64")
prof(8, 8, 64, 24, 24, False, "self-attention L=24 d=64")
prof(8, 8, 64, 1, 1, False, "self-attention L=1 d=64")
