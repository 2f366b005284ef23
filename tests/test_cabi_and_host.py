"""CPU-only tests: the C ABI of libjen1_hip.so (exports, struct layout, argument validation -- none of
which needs a GPU), the host-side logic (packing, schedule tables, generic sampler) and the loud failure
of the product path when no GPU / no library is present."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

from helpers import ROOT, SEED, golden, rel_err
from jen1_amd import lib as L
from jen1_amd.config import UNetSpec, full_model_config, tiny_model_config
from oracle import jen1_oracle as O

HEADERS = [os.path.join(ROOT, "include", h) for h in ("jen1_hip.h", "jen1_train.h", "jen1_deep.h", "jen1_long.h")]


@pytest.fixture(scope="module")
def lib():
    L.build()
    return L.load()


# ------------------------------------------------------------------ C ABI
def test_every_declared_symbol_is_exported(lib):
    src = "\n".join(open(h).read() for h in HEADERS)
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    declared = set(re.findall(r"\b(jen1_[a-z0-9_]+)\s*\(", src))
    assert declared, "no declarations parsed"
    assert declared == set(L.SYMBOLS), (declared ^ set(L.SYMBOLS))
    for name in declared:
        assert getattr(lib, name) is not None


def test_build_info_and_tile_table(lib):
    assert lib.jen1_abi_version() == 2
    assert b"gfx950" in lib.jen1_build_info()
    assert [(lib.jen1_cfg_bm(c), lib.jen1_cfg_bn(c)) for c in range(5)] == [(64, 64), (128, 64), (16, 64), (16, 32), (16, 16)]
    assert lib.jen1_cfg_bm(99) == -1


def test_struct_layout_matches_header():
    """ctypes mirrors vs the C compiler's view of include/jen1_hip.h"""
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "jen1_hip.h"
#include "jen1_train.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(jen1_conv_args), offsetof(jen1_conv_args, dtype), offsetof(jen1_conv_args, gn_eps),
         offsetof(jen1_conv_args, cfg), offsetof(jen1_conv_args, zeros), offsetof(jen1_conv_args, ln_fold), sizeof(jen1_norm_args));
  printf("%zu %zu %zu %d\n", offsetof(jen1_conv_args, nseg), offsetof(jen1_conv_args, seg), sizeof(jen1_conv_seg), JEN1_MAX_SEG);
  printf("%zu %zu %zu %zu %zu %zu\n", sizeof(jen1_gemm_operand), offsetof(jen1_gemm_operand, zdiv), sizeof(jen1_gemm_args),
         offsetof(jen1_gemm_args, c), offsetof(jen1_gemm_args, M), offsetof(jen1_gemm_args, alpha));
  printf("%zu %zu %zu %zu %zu\n", sizeof(jen1_bgemm_group), sizeof(jen1_bgemm_args), offsetof(jen1_bgemm_args, M), offsetof(jen1_bgemm_args, alpha),
         offsetof(jen1_bgemm_args, c));
  return 0;
}
'''
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "t.c"), os.path.join(d, "t")
        open(src, "w").write(prog)
        subprocess.run(["gcc", "-I", os.path.dirname(HEADERS[0]), src, "-o", exe], check=True)
        got = [int(v) for v in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    a = L.ConvArgs
    want = [C.sizeof(a), a.dtype.offset, a.gn_eps.offset, a.cfg.offset, a.zeros.offset, a.ln_fold.offset, C.sizeof(L.NormArgs),
            a.nseg.offset, a.seg.offset, C.sizeof(L.ConvSeg), L.MAX_SEG,
            C.sizeof(L.GemmOperand), L.GemmOperand.zdiv.offset, C.sizeof(L.GemmArgs), L.GemmArgs.c.offset, L.GemmArgs.M.offset,
            L.GemmArgs.alpha.offset,
            C.sizeof(L.BGemmGroup), C.sizeof(L.BGemmArgs), L.BGemmArgs.M.offset, L.BGemmArgs.alpha.offset, L.BGemmArgs.c.offset]
    assert got == want


def test_training_ops_validate_arguments_without_a_gpu(lib):
    g = L.GemmArgs()
    assert lib.jen1_train_gemm(None, None) != 0 and b"args is NULL" in lib.jen1_last_error()
    g.dtype, g.M, g.N, g.K, g.taps, g.batches, g.splitk, g.c = L.F32, 4, 4, 4, 1, 1, 2, 16
    assert lib.jen1_train_gemm(C.byref(g), None) != 0 and b"atomic" in lib.jen1_last_error()
    g.atomic = 1
    assert lib.jen1_train_gemm(C.byref(g), None) != 0 and b"float32 C" in lib.jen1_last_error()
    assert lib.jen1_gn_sums(16, 16, 1, 4, 12, 16, 8, L.F32, None) != 0 and b"divisible" in lib.jen1_last_error()
    assert lib.jen1_ln_forward(16, 16, 16, 16, 16, 4, 4096, 4096, 1e-5, L.F32, None) != 0 and b"bad shape" in lib.jen1_last_error()
    assert lib.jen1_act_forward(16, 16, 8, 7, L.F32, None) != 0
    assert lib.jen1_softmax_forward(16, 16, 7, 2, 4, 4, 4, 0, L.F32, None) != 0     # rows not a multiple of Nq
    assert lib.jen1_colsum(None, None, 1, 1, 1, L.F32, None) != 0
    assert lib.jen1_convert_clear(16, 16, 6, L.F32, None) != 0 and b"multiple of 4" in lib.jen1_last_error()


def test_argument_validation_reports_errors_without_a_gpu(lib):
    a = L.ConvArgs()
    assert lib.jen1_conv_gemm(None, None) != 0 and b"null args" in lib.jen1_last_error()
    a.x0, a.w, a.y = 16, 16, 16
    a.dtype, a.c0, a.ld0 = L.F32, 40, 40               # not a multiple of 32
    assert lib.jen1_conv_gemm(C.byref(a), None) != 0
    assert b"multiples of 32" in lib.jen1_last_error()
    a.c0, a.ld0, a.M, a.out_C, a.ps_f = 64, 64, 32, 32, 1
    a.taps, a.stride, a.B, a.L_in, a.L_out = 3, 1, 2, 10, 10
    a.ld_y, a.cfg, a.tb, a.nb, a.kc_stage, a.splitk = 32, L.CFG_S16x16, 10, 2, 2, 1     # nb*tb = 20 > BN = 16
    assert lib.jen1_conv_gemm(C.byref(a), None) != 0
    assert b"exceeds BN" in lib.jen1_last_error()
    a.tb, a.nb, a.direct, a.pro_mode = 10, 1, 1, L.PRO_GN_SILU
    assert lib.jen1_conv_gemm(C.byref(a), None) != 0          # incomplete GroupNorm prologue / direct with prologue
    # explicit K segments: only in direct mode, every segment needs a pointer and a sane pitch
    a.pro_mode, a.direct, a.nseg = L.PRO_NONE, 0, 2
    assert lib.jen1_conv_gemm(C.byref(a), None) != 0 and b"need direct mode" in lib.jen1_last_error()
    a.direct = 1
    a.seg[0].x, a.seg[0].ld, a.seg[0].kch = 16, 64, 2
    a.seg[1].x, a.seg[1].ld, a.seg[1].kch = 16, 32, 2            # pitch smaller than the channels it claims
    assert lib.jen1_conv_gemm(C.byref(a), None) != 0 and b"bad K segment 1" in lib.jen1_last_error()
    a.nseg = L.MAX_SEG + 1
    assert lib.jen1_conv_gemm(C.byref(a), None) != 0 and b"bad nseg" in lib.jen1_last_error()
    assert lib.jen1_attention(None, None, None, None, None, None, None, None, 0, 0, 0, 1, 1, 8, 1, 1, 8, 0, 8, 0, 0, 8, 0,
                              1.0, L.F32, None) != 0
    n = L.NormArgs()
    assert lib.jen1_norm_apply(C.byref(n), None) != 0


def test_product_path_fails_loudly_without_gpu_or_library():
    from jen1_amd.model import UNetCFG1d
    if not torch.cuda.is_available():
        m = UNetCFG1d(**tiny_model_config(), init_seed=1234, device="cpu")
        with pytest.raises(L.Jen1HipError):
            m.engine()
    # a missing shared library is an error, never a fallback
    code = ("import sys; sys.path.insert(0, %r); from jen1_amd import lib\n"
            "try:\n    lib.load()\nexcept lib.Jen1HipError as e:\n    print('RAISED', 'no CPU / eager fallback' in str(e))\n") % os.path.join(ROOT, "jen-1-pytorch_amd")
    env = dict(os.environ, JEN1_LIB="/nonexistent/libjen1_hip.so")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env).stdout
    assert "RAISED True" in out
    # nothing in the product package imports the oracle
    pkg = os.path.join(ROOT, "jen-1-pytorch_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                assert "oracle" not in open(os.path.join(dp, f)).read().replace("the oracle", ""), f


# ------------------------------------------------------------------ host logic
def test_level_table_matches_survey_appendix_b():
    spec = UNetSpec(**full_model_config())
    assert spec.num_params() == 296_543_106
    assert spec.level_lengths(1500) == [1500, 1500, 375, 94, 24, 12, 6, 3, 2, 1]
    assert spec.level_lengths(9000) == [9000, 9000, 2250, 563, 141, 71, 36, 18, 9, 5]
    assert spec.level_lengths(300)[:5] == [300, 300, 75, 19, 5]
    assert len(spec.res_blocks()) == 56 and len(spec.transformers()) == 13
    with pytest.raises(AssertionError):
        UNetSpec(**dict(tiny_model_config(), bogus_kwarg=1))           # reference model.py:110


def test_weight_packing_layout_and_transposed_conv():
    from jen1_amd.packing import conv_weight_to_gemm, convT_weight_to_gemm, fold_layernorm, pack_gemm_weight
    g = torch.Generator().manual_seed(0)
    w = torch.randn(2, 48, 70, generator=g)                                 # [taps][M][K], K padded to 96
    pk = pack_gemm_weight(w, torch.float32)
    assert pk.shape == (2, 3, 3, 64, 8)
    for (tap, m, c) in [(0, 0, 0), (1, 17, 33), (1, 47, 69), (0, 31, 64)]:
        lane = ((c % 32) // 8) * 16 + (m % 16)
        assert pk[tap, c // 32, m // 16, lane, c % 8] == w[tap, m, c]
    assert float(pk[:, 2, :, 16:, :].abs().max()) == 0.0                   # K = 70: chunk 2 only has k 64..69 (g = 0)
    assert float(pack_gemm_weight(torch.ones(1, 16, 33), torch.float32)[0, 1, 0].sum()) == 16.0
    # ConvTranspose1d(k=2f, s=f) == 2-tap sub-pixel GEMM (reference blocks.py:88-95)
    for f in (2, 4):
        ci, co, Ln = 6, 5, 7
        wt = torch.randn(ci, co, 2 * f, generator=g)
        x = torch.randn(2, ci, Ln, generator=g)
        ref = O.conv_transpose1d(x.numpy(), wt.numpy(), None, f, f // 2 + f % 2, f % 2)
        wg = convT_weight_to_gemm(wt, f).numpy()                              # [2][f*co][ci]
        p = f // 2 + f % 2
        xp = np.pad(x.numpy(), ((0, 0), (0, 0), (1, 1)))                      # x[q-1], x[q] with zero ends
        out = np.zeros_like(ref)
        for q in range(Ln + 1):
            y = np.einsum("mc,bc->bm", wg[0], xp[:, :, q]) + np.einsum("mc,bc->bm", wg[1], xp[:, :, q + 1])
            for r in range(f):
                t = q * f + r - p
                if 0 <= t < f * Ln:
                    out[:, :, t] = y[:, r * co:(r + 1) * co]
        assert rel_err(out, ref) < 1e-5
    # LayerNorm folding: Linear(LN(x)) == (W diag(gamma)) xhat + W beta
    W_, gam, bet = torch.randn(5, 9, generator=g), torch.rand(9, generator=g) + 0.5, torch.randn(9, generator=g)
    xx = torch.randn(4, 9, generator=g)
    wf, bf = fold_layernorm(W_, gam, bet)
    xh = (xx - xx.mean(-1, keepdim=True)) / torch.sqrt(xx.var(-1, unbiased=False, keepdim=True) + 1e-5)
    ref = torch.nn.functional.linear(torch.nn.functional.layer_norm(xx, (9,), gam, bet), W_)
    assert torch.allclose(xh @ wf.T + bf, ref, atol=1e-5)
    assert conv_weight_to_gemm(torch.zeros(3, 4, 5)).shape == (5, 3, 4)


def test_diffusion_tables_and_coefficients_match_reference_goldens():
    from jen1_amd.diffusion import GaussianDiffusion, get_beta_schedule
    g = golden("schedule")
    for name in ("linear", "cosine"):
        betas, none = get_beta_schedule(name, 1000)
        assert none is None
        np.testing.assert_allclose(betas.numpy().astype(np.float32), g[f"{name}.betas"], rtol=2e-7)
        gd = GaussianDiffusion(steps=1000, betas=betas.float(), objective="noise", loss_type="l2", device="cpu", sampling_timesteps=100)
        for attr in ("alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
                     "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"):
            np.testing.assert_allclose(getattr(gd, attr).numpy(), g[f"{name}.{attr}"], rtol=1e-6, atol=1e-12)
    with pytest.raises(NotImplementedError):
        get_beta_schedule("angle", 10)
    betas, _ = get_beta_schedule("linear", 1000)
    for S in (10, 100):
        gd = GaussianDiffusion(steps=1000, betas=betas, objective="noise", loss_type="l2", device="cpu", sampling_timesteps=S)
        pairs = gd.ddim_time_pairs()
        assert [p[0] for p in pairs] + [pairs[-1][1]] == list(g[f"ddim_times.{S}"])
        coef, times = gd.ddim_coeff_table()
        assert coef.shape == (S, 8) and times.tolist() == [p[0] for p in pairs]
        np.testing.assert_allclose(coef[:-1, 2:5].numpy(), g[f"ddim_coeffs.{S}"], rtol=2e-6, atol=1e-8)
        assert coef[-1, 5] == 1.0 and float(coef[:-1, 5].abs().sum()) == 0.0       # only the last step is "x = x0"
    with pytest.raises(AssertionError):
        GaussianDiffusion(steps=10, betas=betas[:10], objective="eps", loss_type="l2", device="cpu")


@pytest.mark.parametrize("objective", ["noise", "x0", "v"])
def test_generic_sampler_and_loss_match_oracle_on_cpu(objective):
    """the literal host restatement of ddim_sample / training_loosses (any callable model) against the numpy oracle"""
    from jen1_amd.diffusion import GaussianDiffusion, get_beta_schedule
    betas, _ = get_beta_schedule("linear", 1000)
    S, shape = 6, (2, 4, 9)
    rng = np.random.default_rng(3)
    Wm = rng.standard_normal((4, 4)).astype(np.float32) * 0.3

    def np_model(x, t, **kw):
        return np.einsum("oc,bct->bot", Wm, x).astype(np.float32) + (t[:, None, None] / 1000.0).astype(np.float32)

    def th_model(x, t, **kw):
        return torch.einsum("oc,bct->bot", torch.from_numpy(Wm), x) + (t[:, None, None] / 1000.0).float()

    cond = {"cross_attn_cond": None, "cross_attn_masks": None, "global_cond": None, "input_concat_cond": None}
    init = rng.standard_normal(shape).astype(np.float32)
    noises = [rng.standard_normal(shape).astype(np.float32) for _ in range(S)]
    og = O.OracleGaussianDiffusion(steps=1000, betas=O.get_beta_schedule("linear", 1000), objective=objective,
                                   cfg_dropout_proba=0.0, embedding_scale=1.0, sampling_timesteps=S)
    ref = og.ddim_sample(lambda x, t, **kw: np_model(x, t), shape, cond, init_noise=init, step_noises=noises)
    gd = GaussianDiffusion(steps=1000, betas=betas, objective=objective, loss_type="l2", device="cpu", cfg_dropout_proba=0.0,
                           embedding_scale=1.0, sampling_timesteps=S)
    got = gd.sample(th_model, shape, cond, init_noise=torch.from_numpy(init), step_noises=[torch.from_numpy(n) for n in noises])
    assert rel_err(got.numpy(), ref) < 1e-4
    allsteps = gd.sample(th_model, shape, cond, return_all_timesteps=True, init_noise=torch.from_numpy(init),
                         step_noises=[torch.from_numpy(n) for n in noises])
    assert allsteps.shape == (2, S + 1, 4, 9)                                  # [B, S+1, C, T] like gdm.py:224
    x0 = rng.standard_normal(shape).astype(np.float32)
    t = np.array([3, 700])
    nz = rng.random(shape).astype(np.float32)
    want = og.training_losses(lambda x, t, **kw: np_model(x, t), x0, t, cond, nz)
    have = gd.training_loosses(th_model, torch.from_numpy(x0), torch.from_numpy(t), cond, noise=torch.from_numpy(nz))
    assert abs(float(have) - float(want)) <= 1e-5 * abs(float(want))


def test_init_fill_is_deterministic_and_keyed():
    from jen1_amd.init_fill import fill
    a, b = fill("to_in.block.block1.project.conv.weight", (4, 3, 3), SEED), fill("to_in.block.block1.project.conv.weight", (4, 3, 3), SEED)
    assert np.array_equal(a, b) and a.dtype == np.float32
    assert not np.array_equal(a, fill("to_in.block.block2.project.conv.weight", (4, 3, 3), SEED))
    assert abs(float(fill("x.groupnorm.weight", (1000,), SEED).mean()) - 1.0) < 0.05       # norm gammas sit around 1


# ------------------------------------------------------------------ multi-process (gloo, world_size 2)
def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    import bench
    dist = bench.init_dist(world, rank, "cpu", backend="gloo")
    dist.barrier()
    dt, value = bench.aggregate(dist, 0.5 * (rank + 1), steps=10, world=world, device="cpu")
    dist.barrier()
    q.put((rank, dt, value))
    dist.destroy_process_group()


def test_bench_rank_aggregation_gloo_world2():
    """N>1 path of bench.py: barrier + max-over-ranks time, whole-job value = all ranks' steps / slowest rank"""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=120) for _ in ps)
    [p.join(timeout=60) for p in ps]
    assert [r[0] for r in res] == [0, 1]
    for _, dt, value in res:
        assert dt == pytest.approx(1.0) and value == pytest.approx(2 * 10 / 1.0)


def test_deep_phase_planner_host_side(lib):
    """include/jen1_deep.h host helpers (no GPU): the layers of the deepest levels of the full configuration become phase
    descriptors, a layer that cannot fit is refused with a message, linking returns the LDS size of the launch"""
    import ctypes as C
    psize = lib.jen1_deep_phase_size()
    assert psize > 0 and psize % 8 == 0

    def conv_args(B, L_in, L_out, c0, c1, M, taps, extras=(), pro=L.PRO_GN_SILU, dtype=L.BF16, stride=1, groups=8):
        a = L.ConvArgs()
        a.x0, a.x1, a.w, a.y, a.bias = 0x1000, (0x2000 if c1 else None), 0x3000, 0x4000, 0x5000
        a.gn_gamma, a.gn_beta = 0x6000, 0x7000
        a.dtype, a.B, a.L_in, a.L_out = dtype, B, L_in, L_out
        a.c0, a.c1, a.ld0, a.ld1 = c0, c1, c0, c1
        a.taps, a.stride, a.pad_left = taps, stride, taps // 2
        a.M, a.out_C, a.ps_f, a.L_y, a.y_brows, a.ld_y = M, M, 1, L_out, L_out, M
        a.pro_mode, a.gn_groups, a.gn_cpg, a.gn_count, a.gn_eps, a.src1_scale = pro, groups, (c0 + c1) // groups, (c0 + c1) // groups * L_in, 1e-5, 0.7
        for i, (c, sh) in enumerate(extras):
            a.seg[i].x, a.seg[i].ld, a.seg[i].shift, a.seg[i].kch = 0x8000 + i, c, sh, c // 32
        a.nseg = len(extras)
        return a

    bufs = []

    def add(a):
        buf = (C.c_char * psize)()
        rc = lib.jen1_deep_phase_conv(C.byref(a), 0, C.cast(buf, C.c_void_p))
        if rc == 0:
            bufs.append(buf)
        return rc
    # level 8 up path (T' = 1): conv1 over cat(x, skip) with 2048 channels, conv2 + 1x1 shortcut as extra K segments
    assert add(conv_args(8, 1, 1, 1024, 1024, 1024, 3)) == 0
    assert add(conv_args(8, 1, 1, 1024, 0, 1024, 3, extras=((1024, 0), (1024, 0)))) == 0
    # level 3 (T' = 24, C = 256), float32 and the CFG pair
    assert add(conv_args(16, 24, 24, 256, 0, 256, 3, extras=((256, 0), (256, 0)), dtype=L.F32)) == 0
    # downsampling conv into level 3: 94 raw rows of 256 channels, k = 9, stride 4
    assert add(conv_args(8, 94, 24, 256, 0, 256, 9, pro=L.PRO_NONE, stride=4)) == 0
    # refused: 150 positions of one batch element are more than four 16-column fragments
    assert add(conv_args(2, 150, 150, 128, 0, 128, 3)) != 0
    assert b"does not fit" in lib.jen1_last_error()
    # refused: LayerNorm prologue
    assert add(conv_args(2, 8, 8, 128, 0, 128, 1, pro=L.PRO_LN)) != 0
    n = len(bufs)
    host = (C.c_char * (psize * n))()
    for i, b in enumerate(bufs):
        C.memmove(C.addressof(host) + i * psize, b, psize)
    bb = lib.jen1_deep_blob_bytes()
    blobs, hdrs = (C.c_char * (bb * n))(), (C.c_int32 * (4 * n))()
    lds = lib.jen1_deep_link(C.cast(host, C.c_void_p), n, 256, C.cast(blobs, C.c_void_p), C.cast(hdrs, C.c_void_p))
    assert 0 < lds <= 160 * 1024
    assert [hdrs[4 * i + 2] for i in range(n)] == [0] * n and hdrs[0] > 0 and hdrs[1] == 0 and hdrs[5] == (hdrs[0] + 7) // 8 * 8 % 256
    # per-wave K-chunk runs of the first phase (T' = 1, non-causal k = 3 over 2048 channels): only the centre tap can touch a
    # real row, so 64 of the 192 chunks are listed: one run of 8 chunks per wave, chunks 64 + wave + 8 j at columns 32 (wave + 8 j)
    cnt = (C.c_int16 * 32).from_buffer(blobs, 1024)
    assert list(cnt) == [8] * 16 + [1] * 16
    run0 = (C.c_int32 * 4).from_buffer(blobs, 1024 + 64)
    run1 = (C.c_int32 * 4).from_buffer(blobs, 1024 + 64 + 12 * 16)
    assert list(run0) == [64, 8, 0, 0] and list(run1) == [65, 8, 32, 0]
    # the tabulated first ring round of wave 1 (12 slots: chunk index, staged offset = shift * pitch + column) behind the 8 x 12 runs:
    # its 8 chunks 65, 73, ... at columns 32, 288, ..., the four slots beyond them repeat the last one
    slot1 = (C.c_int32 * 24).from_buffer(blobs, 1024 + 64 + 8 * 12 * 16 + 1 * 96)
    assert list(slot1)[:12] == [65 + 8 * j for j in range(8)] + [121] * 4
    assert list(slot1)[12:] == [32 + 256 * j for j in range(8)] + [32 + 256 * 7] * 4


def test_torch_library_ops_are_registered_with_schemas_and_fake_impls():
    """BASELINE north_star: "through PyTorch-ROCm custom ops".  Every op of jen1_amd/ops.py has a dispatcher schema and a fake
    implementation (FakeTensor tracing needs no GPU and no extension call); UNetCFG1d.forward is the unet_cfg_forward op."""
    import torch
    from torch._subclasses.fake_tensor import FakeTensorMode
    from jen1_amd import ops
    from jen1_amd.config import tiny_model_config
    from jen1_amd.model import UNetCFG1d
    for name in ops.OPS:
        op = getattr(torch.ops.jen1, name)
        assert str(op.default._schema).startswith(f"jen1::{name}("), op.default._schema
    sch = str(torch.ops.jen1.unet_cfg_forward.default._schema)
    assert "Tensor x, Tensor time, Tensor embedding, Tensor? embedding_mask, Tensor? context" in sch and sch.endswith("-> Tensor")
    cfg = tiny_model_config()
    model = UNetCFG1d(**cfg, init_seed=None, device="cpu")
    B, T = 2, 96
    with FakeTensorMode():
        x = torch.empty((B, model.spec.in_channels, T))
        t = torch.empty((B,), dtype=torch.int64)
        emb = torch.empty((B, model.spec.ctx_max_length, model.spec.ctx_features))
        msk = torch.empty((B, model.spec.ctx_max_length))
        ctx = torch.empty((B, model.spec.ctx_ch0, T))
        y = model(x, t, embedding=emb, embedding_mask=msk, embedding_scale=0.8, batch_cfg=True, scale_cfg=True, channels_list=[ctx])
        assert tuple(y.shape) == (B, model.spec.out_channels, T) and y.dtype == torch.float32
        h = torch.empty((B, T, 64), dtype=torch.bfloat16)
        g = torch.empty((64,))
        yy, sums = torch.ops.jen1.group_norm(h, g, g, None, 64, 8, 1e-5, True)
        assert yy.shape == h.shape and tuple(sums.shape) == (B, 8, 2)
        yl, st = torch.ops.jen1.layer_norm(h, g, g, 1e-5)
        assert yl.shape == h.shape and tuple(st.shape) == (B * T, 2)
        assert torch.ops.jen1.activation(h, 0).shape == h.shape
        assert tuple(torch.ops.jen1.cfg_combine(torch.empty((2 * B, T, 128)), 128, 0.8, True, 0.7).shape) == (B, 128, T)
        # the GEMM-shaped training operators: _Conv1d (strided, causal), ConvTranspose1d, Linear and the attention core
        assert tuple(ops.conv1d_same(h, torch.empty((100, 64, 5)), torch.empty((100,)), 2, True).shape) == (B, (T - 1) // 2 + 1, 104)
        assert tuple(ops.conv_transpose1d(h, torch.empty((64, 32, 4)), None, 2, 1, 0).shape) == (B, 2 * T, 32)
        assert tuple(ops.linear(h, torch.empty((256, 64))).shape) == (B, T, 256)
        o, pr = torch.ops.jen1.attention(h, torch.empty((B, 129, 128), dtype=torch.bfloat16), 8, False, torch.empty((B, 129)))
        assert o.shape == h.shape and tuple(pr.shape) == (B * 8, T, 136)
        dx, dw, db = torch.ops.jen1.conv_backward(torch.empty((B, T, 104), dtype=torch.bfloat16), h, torch.empty((100, 64, 5)), 0, 1, 2, True, True)
        assert dx.shape == h.shape and tuple(dw.shape) == (100, 64, 5) and dw.dtype == torch.float32 and tuple(db.shape) == (100,)
    # the real call without a GPU fails loudly (no CPU path)
    import pytest
    from jen1_amd.lib import Jen1HipError
    with pytest.raises((Jen1HipError, RuntimeError)):
        model(torch.zeros((B, model.spec.in_channels, T)), torch.zeros((B,), dtype=torch.int64),
              embedding=torch.zeros((B, model.spec.ctx_max_length, model.spec.ctx_features)), channels_list=[torch.zeros((B, model.spec.ctx_ch0, T))])


def test_causal_rows_pads_follow_conv1d_padding():
    """train.CausalRows: the per-clip index-map shifts of a pass that mixes causal and non-causal clips restate _Conv1d's padding
    (blocks.py:45-50: k - 1 on the left when causal, else (k - 1) // 2), one entry longer than the batch; ``twice`` stacks the
    batch on itself for the CFG pair (model.py:349-353)"""
    import torch
    from jen1_amd.train import CausalRows
    rows = CausalRows(torch.tensor([0, 1, 1, 0], dtype=torch.bool))
    for k, nc, c in ((3, 1, 2), (5, 2, 4), (9, 4, 8)):
        neg, pos = rows.pads(k)
        assert pos.dtype == torch.int32 and neg.dtype == torch.int32
        assert pos.tolist() == [nc, c, c, nc, nc] and neg.tolist() == [-nc, -c, -c, -nc, -nc]
        assert rows.pads(k)[0] is neg          # cached per kernel size
    assert rows.twice().flags.tolist() == [0, 1, 1, 0, 0, 1, 1, 0]


def test_ctypes_mirrors_have_the_size_of_the_c_structs(tmp_path):
    """lib.GemmArgs / GemmOperand / RepackEntry against sizeof() of the structs in include/jen1_train.h as gcc lays them out (the
    mirrors are filled field by field on the host and read by the kernels' launchers)"""
    import ctypes
    import subprocess
    from jen1_amd import lib as L
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stdint.h>\n#include "jen1_hip.h"\n#include "jen1_train.h"\n'
                   'int main(void) { printf("%zu %zu %zu %zu\\n", sizeof(jen1_gemm_operand), sizeof(jen1_gemm_args), sizeof(jen1_repack_entry), sizeof(jen1_kv_layer)); return 0; }\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", f"-I{os.path.join(root, 'include')}", str(src), "-o", str(exe)], check=True)
    sizes = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    assert sizes == [ctypes.sizeof(L.GemmOperand), ctypes.sizeof(L.GemmArgs), ctypes.sizeof(L.RepackEntry), ctypes.sizeof(L.KvLayer)]


def test_capture_helper_keeps_the_collector_out_of_the_window(monkeypatch):
    """jen1_amd/graphs.py (the fix of the round-4 SIGABRT): the cyclic collector runs BEFORE the capture window, is off inside it (a dead
    hipGraph reaching torch's ~CUDAGraph inside somebody's capture aborts the process on ROCm), and comes back afterwards -- also when the
    recorded code raises, and only if it was on before.  Host logic only: torch.cuda.graph is replaced by a recorder."""
    import contextlib
    import gc
    import torch
    from jen1_amd import graphs
    seen = {}

    class Cycle:
        def __init__(self):
            self.me = self

        def __del__(self):
            seen.setdefault("collected_while_enabled", []).append(gc.isenabled())

    @contextlib.contextmanager
    def fake_graph(g, capture_error_mode=None, **kw):
        seen["mode"], seen["inside_enabled"], seen["kw"] = capture_error_mode, gc.isenabled(), kw
        yield

    monkeypatch.setattr(torch.cuda, "graph", fake_graph)
    assert gc.isenabled()
    Cycle()                                               # garbage in a reference cycle, pending
    with graphs.capture(object()):
        assert not gc.isenabled()
        assert seen["collected_while_enabled"] == [True]  # the pending cycle went BEFORE the window opened
    assert gc.isenabled()
    assert seen["mode"] == "thread_local" and seen["inside_enabled"] is False and seen["kw"] == {}
    with pytest.raises(RuntimeError):
        with graphs.capture(object(), pool=(0, 1)):
            assert seen["kw"] == {"pool": (0, 1)}
            raise RuntimeError("recorded code failed")
    assert gc.isenabled()
    gc.disable()
    try:
        with graphs.capture(object()):
            pass
        assert not gc.isenabled()                         # it was off before: it stays off
    finally:
        gc.enable()


def test_long_phase_host_helpers_without_a_gpu(lib):
    """include/jen1_long.h host helpers (no GPU): geometry of the sample-resident launches at the bench shapes, a descriptor for a
    level-0 ConvBlock1d (GroupNorm + FiLM + SiLU prologue over [x, skip], k = 3, 1x1 shortcut as extra K segments), and the refusals that
    send a plan back to one launch per layer"""
    mb, tl, tb = C.c_int(0), C.c_int(0), C.c_int(0)
    geo = lambda M, L_out, G: (lib.jen1_long_geometry(M, L_out, G, C.byref(mb), C.byref(tl), C.byref(tb)), mb.value, tl.value, tb.value)
    assert geo(128, 1500, 32) == (0, 1, 32, 47)            # B = 8: 32 workgroups per sample, 47 positions per tile at T = 1500
    assert geo(128, 375, 32) == (0, 1, 32, 12)
    assert geo(256, 94, 32) == (0, 2, 16, 6)               # 256 output channels: two M blocks x 16 position tiles
    assert geo(512, 376, 32) == (0, 4, 8, 47)              # ConvTranspose1d(k = 8, s = 4) as a sub-pixel GEMM of 4 x 128 rows
    assert geo(128, 1500, 16) == (0, 1, 16, 94)            # CFG pair: 16 samples
    assert geo(128, 9000, 128) == (0, 1, 128, 71)          # T = 9000, CFG pair of one clip
    assert geo(64, 300, 64)[0] != 0 and b"multiple of 128" in lib.jen1_last_error()       # the tiny configuration: 64 channels
    assert geo(128, 9000, 32)[0] != 0                      # 282 positions per tile: more than 96
    buf = (C.c_char * 512)()
    a = L.ConvArgs()
    fake = 1 << 20                                         # (descriptors only store the pointers)
    a.x0, a.x1, a.w, a.bias, a.y = fake, fake + 4096, fake + 8192, fake + 12288, fake + 16384
    a.dtype, a.B, a.L_in, a.L_out = L.BF16, 8, 1500, 1500
    a.c0, a.c1, a.ld0, a.ld1 = 128, 128, 128, 128
    a.taps, a.stride, a.pad_left = 3, 1, 1
    a.M, a.out_C, a.ps_f, a.ps_off, a.L_y, a.y_brows, a.ld_y = 128, 128, 1, 0, 1500, 1500, 128
    a.pro_mode, a.gn_groups, a.gn_cpg, a.gn_count, a.gn_eps, a.src1_scale = L.PRO_GN_SILU, 8, 32, 32 * 1500, 1e-5, 2 ** -0.5
    a.gn_gamma, a.gn_beta = fake + 20480, fake + 24576
    a.live_mask = 1
    st = fake + 32768
    rc = lib.jen1_long_phase_conv(C.byref(a), 32, st, 32, 1, 8, st + 4096, 32, 1, 8, 1, fake + 65536, C.cast(buf, C.c_void_p))
    assert rc == 0, lib.jen1_last_error()
    assert lib.jen1_long_phase_units(C.cast(buf, C.c_void_p)) == 32
    lds = lib.jen1_long_phase_lds(C.cast(buf, C.c_void_p))
    assert 49 * 264 * 2 <= lds <= 49 * 264 * 2 + 4096      # 47 + 2 halo rows x (256 + 8) bf16 + the affine tables
    a.out_C = a.M = 192                                    # not a multiple of 128
    assert lib.jen1_long_phase_conv(C.byref(a), 32, st, 32, 1, 8, st + 4096, 32, 1, 8, 1, None, C.cast(buf, C.c_void_p)) != 0
    a.out_C = a.M = 128
    a.y_f32 = 1
    assert lib.jen1_long_phase_conv(C.byref(a), 32, st, 32, 1, 8, st + 4096, 32, 1, 8, 1, None, C.cast(buf, C.c_void_p)) != 0
    # launch arguments are checked before anything is enqueued
    assert lib.jen1_long_run(None, 1, 8, None, None, 256, 1 << 15, L.BF16, 0, None) != 0
    assert lib.jen1_long_run(fake, 1, 6, fake, None, 256, 1 << 15, L.BF16, 1, None) != 0     # XCD-local stores: a multiple of 8 samples
