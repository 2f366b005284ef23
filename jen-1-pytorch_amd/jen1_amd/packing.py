"""Host-side weight packing into the MFMA fragment order jen1_conv_gemm streams.

Packed layout (DESIGN.md "weights in HBM"):  [tap][c/32][m/16][lane 0..63][8]
with lane = g*16 + i, element j  <->  W[tap][m = 16*mt + i][c = 32*kc + 8*g + j].
One wave-instruction therefore reads one contiguous 1 KiB (bf16) / 2 KiB (f32)
block.  The M tile is the fastest block index on purpose: the workgroups of a launch own
different M tiles and walk K in lockstep, so at any instant their loads form one contiguous
span that spreads over all HBM channels (with M-tile-major blocks they sat 32 KiB apart and
camped on a few channels).

All functions take the reference's parameter tensors (reference ``state_dict``
layouts) and return device tensors in the compute dtype.
"""
from __future__ import annotations

import torch


def _ceil_to(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def pack_gemm_weight(w_tmk: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """w_tmk: [taps][M][K] float32 (already in GEMM form) -> packed [taps][Kp/32][M/16][64][8].
    M must be a multiple of 16; K is zero-padded to a multiple of 32."""
    taps, M, K = w_tmk.shape
    assert M % 16 == 0, M
    Kp = _ceil_to(K, 32)
    if Kp != K:
        w_tmk = torch.nn.functional.pad(w_tmk, (0, Kp - K))
    w = w_tmk.reshape(taps, M // 16, 16, Kp // 32, 4, 8)          # [tap, mt, i, kc, g, j]
    w = w.permute(0, 3, 1, 4, 2, 5).contiguous()                   # [tap, kc, mt, g, i, j]
    return w.reshape(taps, Kp // 32, M // 16, 64, 8).to(dtype).contiguous()


def conv_weight_to_gemm(w: torch.Tensor) -> torch.Tensor:
    """nn.Conv1d weight [C_out][C_in][k] -> [k][C_out][C_in] (reference blocks.py:42)."""
    return w.permute(2, 0, 1).contiguous()


def convT_weight_to_gemm(w: torch.Tensor, f: int) -> torch.Tensor:
    """nn.ConvTranspose1d weight [C_in][C_out][2f] (reference blocks.py:88-95) as a 2-tap
    sub-pixel GEMM: out[q*f + r - p] = x[q] . W[:, :, r] + x[q-1] . W[:, :, r+f].
    Returns [2][f*C_out][C_in] with tap 0 <-> x[q-1], tap 1 <-> x[q]; row m = r*C_out + co."""
    ci, co, k = w.shape
    assert k == 2 * f
    w_r = w.permute(2, 1, 0)                       # [j][co][ci]
    tap1 = w_r[:f].reshape(f * co, ci)             # j = r       <-> x[q]
    tap0 = w_r[f:].reshape(f * co, ci)             # j = r + f   <-> x[q-1]
    return torch.stack([tap0, tap1], dim=0).contiguous()


def fold_layernorm(w: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor):
    """Linear(LayerNorm(x)) = (W diag(gamma)) xhat + W beta  (reference blocks.py:427-429):
    returns (W', b') so the kernel's LN prologue only standardises."""
    return w * gamma[None, :], w @ beta
