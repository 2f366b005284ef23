// Fused implicit-GEMM 1-D convolution / linear kernel for gfx950 (MI355X, CDNA4).
//
// One kernel family covers every GEMM-shaped op of the JEN-1 denoiser
// (reference jen1/model/blocks.py: _Conv1d :34-53, Upsample1d :69-95,
// ConvBlock1d :137-145, ResnetBlock1d :219-231, Attention projections :427-429,
// FeedForward :440-446, MappingToScaleShift :148-165):
//
//   D[m][n] = sum_{tap, c} W[tap][m][c] * pro(X[n -> (b, q*stride + tap - pad_left)][c])
//
//   * A operand = weights, pre-packed on the host into MFMA fragment order
//     [tap][m/16][c/32][lane 0..63][8] so that every wave-instruction streams one
//     contiguous 1 KiB (bf16) / 2 KiB (f32) block from HBM -- the step is
//     weight-streaming bound.
//   * B operand = channel-last activations [B][L][C]; the tile (+ conv halo) is staged
//     ONCE in LDS with the GroupNorm(+FiLM)+SiLU / LayerNorm prologue applied, then
//     every tap reads a row-shifted view of it (no F.pad copy, no cat copy, no
//     separate norm/activation pass).
//   * MFMA: v_mfma_f32_16x16x32_bf16 (bf16 mode) or 8 x v_mfma_f32_16x16x4_f32
//     (float32 parity mode, exact fp32 FMA chain).  64-lane wavefronts, 4 waves / WG.
//   * epilogue: bias, GELU, residual, row mask, sub-pixel (transposed-conv) row
//     mapping with crop, and the statistics of the NEXT norm layer (GroupNorm
//     fine-group sums, LayerNorm row sums) via LDS + global float atomics.
//   * split-K for the deep, skinny levels (T' <= 24): partial slabs + agent-scope
//     release / ticket / acquire, last-arriving workgroup reduces and runs the
//     epilogue (cdna_hip_programming.md section 5 "in-launch split-K reduction").
#include "common.h"

namespace {

struct Layout {
  int ldsld;      // LDS row pitch in elements (stage channels + 8)
  int seg;        // staged input rows per batch element
  int tile_off, tab_a_off, tab_b_off, grp_off, row_off, stats_off, misc_off, total;
};

__host__ __device__ inline int align16(int x) { return (x + 15) & ~15; }

__host__ __device__ inline Layout make_layout(const jen1_conv_args& a, int esize) {
  Layout L;
  const int kch_total = (a.c0 + a.c1) / 32;
  const int cps = (kch_total + a.splitk - 1) / a.splitk;
  const int stage_ch = 32 * (a.kc_stage < cps ? a.kc_stage : cps);
  L.ldsld = stage_ch + 8;
  L.seg = (a.tb - 1) * a.stride + a.taps;
  int off = 0;
  L.tile_off = off;
  off = align16(off + a.nb * L.seg * L.ldsld * esize);
  const bool gn = (a.pro_mode == JEN1_PRO_GN || a.pro_mode == JEN1_PRO_GN_SILU);
  L.tab_a_off = off;
  if (gn) off = align16(off + a.nb * stage_ch * 4);
  L.tab_b_off = off;
  if (gn) off = align16(off + a.nb * stage_ch * 4);
  L.grp_off = off;
  if (gn) off = align16(off + a.nb * JEN1_FINE_GROUPS * 2 * 4);
  L.row_off = off;
  if (a.pro_mode == JEN1_PRO_LN) off = align16(off + a.nb * L.seg * 2 * 4);
  L.stats_off = off;
  if (a.out_gn_stats) off = align16(off + a.nb * JEN1_FINE_GROUPS * 2 * 4);
  L.misc_off = off;
  off += 16;
  L.total = off;
  return L;
}

// ---- MFMA wrappers: both dtypes use "lane (i = l&15, g = l>>4) owns K elements 8g..8g+7" ----
__device__ __forceinline__ void mma32(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma32(f32x4& acc, const f32x8& a, const f32x8& b) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], acc, 0, 0, 0);
}

template <typename T>
struct FragOf;
template <>
struct FragOf<float> {
  typedef f32x8 type;
};
template <>
struct FragOf<bf16_t> {
  typedef bf16x8 type;
};

__device__ __forceinline__ void frag_load(f32x8& f, const float* p) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
  f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
}
__device__ __forceinline__ void frag_load(bf16x8& f, const bf16_t* p) { f = *reinterpret_cast<const bf16x8*>(p); }

template <typename T, int MF, int NF, int WM, int WN, int PF>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const jen1_conv_args a) {
  typedef typename FragOf<T>::type Frag;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BM = 16 * MF * WM;
  constexpr bool PRECISE = is_f32<T>::value;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 15, lg = lane >> 4;

  const Layout L = make_layout(a, (int)sizeof(T));
  T* tile = reinterpret_cast<T*>(smem + L.tile_off);
  float* tab_a = reinterpret_cast<float*>(smem + L.tab_a_off);
  float* tab_b = reinterpret_cast<float*>(smem + L.tab_b_off);
  float* grp = reinterpret_cast<float*>(smem + L.grp_off);
  float* rowtab = reinterpret_cast<float*>(smem + L.row_off);
  float* st_lds = reinterpret_cast<float*>(smem + L.stats_off);
  int* misc = reinterpret_cast<int*>(smem + L.misc_off);

  // ---- tile coordinates -------------------------------------------------------------------
  const int tiles_t = (a.L_out + a.tb - 1) / a.tb;
  const int bt = blockIdx.y / tiles_t, tt = blockIdx.y - bt * tiles_t;
  const int b0 = bt * a.nb, t0 = tt * a.tb;
  const int seg = L.seg, ldsld = L.ldsld;
  const int tin0 = t0 * a.stride - a.pad_left;
  const int ctot = a.c0 + a.c1;
  const int kch_total = ctot / 32;
  const int cps = (kch_total + a.splitk - 1) / a.splitk;
  const int z = blockIdx.z;
  const int kc_begin = z * cps;
  const int kc_end = (kc_begin + cps < kch_total) ? kc_begin + cps : kch_total;
  const int MT = a.M / 16;
  const int mt_base = blockIdx.x * (BM / 16) + wm * MF;
  const int n_rows = a.nb * a.tb;

  // per-lane output column mapping
  int rowbase[NF];
  int n_b[NF], n_t[NF];
  bool n_ok[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int n = (wn * NF + nf) * 16 + li;
    const int bl = n / a.tb, tl = n - bl * a.tb;
    n_ok[nf] = (n < n_rows) && (b0 + bl < a.B) && (t0 + tl < a.L_out);
    n_b[nf] = bl;
    n_t[nf] = tl;
    rowbase[nf] = n_ok[nf] ? (bl * seg + tl * a.stride) : 0;
  }

  f32x4 acc[MF][NF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const T* wbase = reinterpret_cast<const T*>(a.w);
  const bool gn = (a.pro_mode == JEN1_PRO_GN || a.pro_mode == JEN1_PRO_GN_SILU);
  const bool do_silu = (a.pro_mode == JEN1_PRO_GN_SILU || a.pro_mode == JEN1_PRO_SILU);

  // ---- one-time prologue tables: GroupNorm group statistics / LayerNorm row statistics ----
  if (gn) {
    // grp[bl][g] = (mean, rstd) of group g of batch element b0+bl, merged from fine-group sums
    const int G = a.gn_groups;
    for (int i = tid; i < a.nb * G; i += 256) {
      const int bl = i / G, g = i - bl * G;
      const int b = b0 + bl;
      float mean = 0.f, rstd = 0.f;
      if (b < a.B) {
        int lo = g * a.gn_cpg, hi = (g == G - 1) ? ctot : lo + a.gn_cpg;
        const float* st;
        int cpf;
        float sc = 1.f;
        if (lo >= a.c0) {
          st = a.gn_stats1 + (size_t)b * 64;
          lo -= a.c0; hi -= a.c0;
          cpf = a.c1 / JEN1_FINE_GROUPS;
          sc = a.src1_scale;
        } else {
          st = a.gn_stats0 + (size_t)b * 64;
          cpf = a.c0 / JEN1_FINE_GROUPS;
          if (hi > a.c0) hi = a.c0;
        }
        float s = 0.f, q = 0.f;
        const int f0 = lo / cpf, f1 = (hi + cpf - 1) / cpf;
        for (int f = f0; f < f1; ++f) {
          s += st[2 * f];
          q += st[2 * f + 1];
        }
        s *= sc;
        q *= sc * sc;
        const float inv_n = 1.0f / (float)a.gn_count;
        mean = s * inv_n;
        float var = q * inv_n - mean * mean;
        var = var < 0.f ? 0.f : var;
        rstd = PRECISE ? 1.0f / sqrtf(var + a.gn_eps) : rsqrtf(var + a.gn_eps);
      }
      grp[2 * i] = mean;
      grp[2 * i + 1] = rstd;
    }
  }
  if (a.pro_mode == JEN1_PRO_LN) {
    const float inv_c = 1.0f / (float)a.ln_C;
    for (int i = tid; i < a.nb * seg; i += 256) {
      const int bl = i / seg, r = i - bl * seg;
      const int b = b0 + bl, tin = tin0 + r;
      float mean = 0.f, rstd = 0.f;
      if (b < a.B && tin >= 0 && tin < a.L_in) {
        const float* rs = a.ln_rowstats + ((size_t)b * a.L_in + tin) * 2;
        mean = rs[0] * inv_c;
        float var = rs[1] * inv_c - mean * mean;
        var = var < 0.f ? 0.f : var;
        rstd = PRECISE ? 1.0f / sqrtf(var + a.ln_eps) : rsqrtf(var + a.ln_eps);
      }
      rowtab[2 * i] = mean;
      rowtab[2 * i + 1] = rstd;
    }
  }
  if (a.out_gn_stats) {
    for (int i = tid; i < a.nb * 64; i += 256) st_lds[i] = 0.f;
  }

  // ---- K loop over LDS stages ---------------------------------------------------------------
  const int stage_chunks = (L.ldsld - 8) / 32;
  for (int ks = kc_begin; ks < kc_end; ks += stage_chunks) {
    const int nch = (ks + stage_chunks <= kc_end) ? stage_chunks : (kc_end - ks);
    const int sch = nch * 32;          // channels in this stage
    const int cst = ks * 32;           // first channel (concat space)
    __syncthreads();                   // previous stage fully consumed; tables above visible
    if (gn) {
      // per-(batch element, channel) affine: v = x * A + B  (GroupNorm * gamma + beta, then FiLM)
      for (int i = tid; i < a.nb * sch; i += 256) {
        const int bl = i / sch, cl = i - bl * sch;
        const int c = cst + cl;
        const int b = b0 + bl;
        float A = 0.f, Bc = 0.f;
        if (b < a.B) {
          int g = c / a.gn_cpg;
          g = g < a.gn_groups ? g : a.gn_groups - 1;
          const float mean = grp[2 * (bl * a.gn_groups + g)], rstd = grp[2 * (bl * a.gn_groups + g) + 1];
          const float gam = a.gn_gamma[c], bet = a.gn_beta[c];
          const float sc = (c >= a.c0) ? a.src1_scale : 1.0f;
          A = rstd * gam;
          Bc = bet - mean * A;
          A *= sc;
          if (a.film) {
            const int fr = a.film_row ? a.film_row[b] : b;
            const float* fp = a.film + (size_t)fr * a.film_ld + a.film_off;
            const float fs = fp[c] + 1.0f, fh = fp[a.film_C + c];
            A *= fs;
            Bc = Bc * fs + fh;
          }
        }
        tab_a[i] = A;
        tab_b[i] = Bc;
      }
      __syncthreads();
    }
    // stage the activation tile (+ halo) with the prologue applied; zeros outside [0, L_in)
    {
      const int vpr = sch / 8;
      const int nvec = a.nb * seg * vpr;
      for (int v = tid; v < nvec; v += 256) {
        const int row = v / vpr, cv = v - row * vpr;
        const int bl = row / seg, r = row - bl * seg;
        const int b = b0 + bl, tin = tin0 + r;
        const int cl = cv * 8;
        int c = cst + cl;
        float x[8];
        const bool ok = (b < a.B) && (tin >= 0) && (tin < a.L_in);
        if (ok) {
          if (c < a.c0) {
            load8(reinterpret_cast<const T*>(a.x0) + ((size_t)b * a.L_in + tin) * a.ld0 + c, x);
          } else {
            load8(reinterpret_cast<const T*>(a.x1) + ((size_t)b * a.L_in + tin) * a.ld1 + (c - a.c0), x);
            if (!gn && a.src1_scale != 1.0f) {
#pragma unroll
              for (int j = 0; j < 8; ++j) x[j] *= a.src1_scale;
            }
          }
          if (gn) {
            const float* ta = tab_a + bl * sch + cl;
            const float* tb_ = tab_b + bl * sch + cl;
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = x[j] * ta[j] + tb_[j];
          } else if (a.pro_mode == JEN1_PRO_LN) {
            const float mean = rowtab[2 * row], rstd = rowtab[2 * row + 1];
            if (a.ln_gamma) {
#pragma unroll
              for (int j = 0; j < 8; ++j) x[j] = (x[j] - mean) * rstd * a.ln_gamma[c + j] + a.ln_beta[c + j];
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j) x[j] = (x[j] - mean) * rstd;
            }
          }
          if (do_silu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = PRECISE ? silu_precise(x[j]) : silu_f(x[j]);
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) x[j] = 0.f;
        }
        store8(tile + (size_t)row * ldsld + cl, x);
      }
    }
    __syncthreads();

    // ---- MFMA loop: taps x chunks, A fragments streamed from global with a PF-deep ring ----
    const int iters = a.taps * nch;
    int ptap = 0, pk = 0;   // prefetch cursor
    auto load_a = [&](Frag(&dst)[MF], int tap, int kcl) {
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        int mt = mt_base + mf;
        mt = mt < MT ? mt : MT - 1;
        const T* p = wbase + ((size_t)((size_t)tap * MT + mt) * kch_total + (ks + kcl)) * 512 + lane * 8;
        frag_load(dst[mf], p);
      }
    };
    auto advance = [&](int& tap, int& kcl) {
      if (++kcl == nch) {
        kcl = 0;
        if (tap + 1 < a.taps) ++tap; else kcl = nch - 1;   // clamp at the last fragment
      }
    };
    Frag cur[PF][MF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      load_a(cur[u], ptap, pk);
      advance(ptap, pk);
    }
    int ctap = 0, ck = 0;
    for (int it = 0; it < iters; it += PF) {
      Frag nxt[PF][MF];
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        load_a(nxt[u], ptap, pk);
        advance(ptap, pk);
      }
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        if (it + u < iters) {
          Frag bfr[NF];
#pragma unroll
          for (int nf = 0; nf < NF; ++nf)
            frag_load(bfr[nf], tile + (size_t)(rowbase[nf] + ctap) * ldsld + ck * 32 + lg * 8);
#pragma unroll
          for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) mma32(acc[mf][nf], cur[u][mf], bfr[nf]);
          if (++ck == nch) { ck = 0; ++ctap; }
        }
      }
#pragma unroll
      for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) cur[u][mf] = nxt[u][mf];
    }
  }

  // ---- split-K: publish partial slab, last arriver reduces ---------------------------------
  if (a.splitk > 1) {
    const unsigned tile_id = blockIdx.y * gridDim.x + blockIdx.x;
    float* slab = a.slab + ((size_t)tile_id * a.splitk) * (size_t)(MF * NF * 1024);
    float* mine = slab + (size_t)z * (MF * NF * 1024);
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
        *reinterpret_cast<float4*>(mine + (size_t)(mf * NF + nf) * 1024 + tid * 4) =
            make_float4(acc[mf][nf][0], acc[mf][nf][1], acc[mf][nf][2], acc[mf][nf][3]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      misc[0] = (int)__hip_atomic_fetch_add(a.counters + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int ticket = misc[0];
    if (ticket != a.splitk - 1) return;
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(a.counters + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    for (int zz = 0; zz < a.splitk; ++zz) {
      if (zz == z) continue;
      const float* other = slab + (size_t)zz * (MF * NF * 1024);
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          const float4 o = *reinterpret_cast<const float4*>(other + (size_t)(mf * NF + nf) * 1024 + tid * 4);
          acc[mf][nf][0] += o.x; acc[mf][nf][1] += o.y; acc[mf][nf][2] += o.z; acc[mf][nf][3] += o.w;
        }
    }
  }

  // ---- epilogue -------------------------------------------------------------------------------
  T* yT = reinterpret_cast<T*>(a.y);
  float* yF = reinterpret_cast<float*>(a.y);
  const T* res = reinterpret_cast<const T*>(a.residual);
  float rs_sum[NF], rs_sq[NF];
  int yrow_n[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) { rs_sum[nf] = 0.f; rs_sq[nf] = 0.f; yrow_n[nf] = -1; }

#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    const int mt = mt_base + mf;
    const int m = mt * 16 + lg * 4;
    const bool m_ok = mt < MT;
    const int ph = m_ok ? m / a.out_C : 0;
    const int co = m - ph * a.out_C;
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    if (m_ok && a.bias) {
      const float4 bb = *reinterpret_cast<const float4*>(a.bias + co);
      bias4[0] = bb.x; bias4[1] = bb.y; bias4[2] = bb.z; bias4[3] = bb.w;
    }
    float gs[2] = {0.f, 0.f}, gq[2] = {0.f, 0.f};   // nb == 1 path: per channel-pair sums over nf
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const int b = b0 + n_b[nf];
      const int ty = (t0 + n_t[nf]) * a.ps_f + ph - a.ps_off;
      const bool ok = m_ok && n_ok[nf] && ty >= 0 && ty < a.L_y;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[mf][nf][r] + bias4[r];
      if (a.act == JEN1_ACT_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
      }
      if (ok) {
        const size_t yrow = (size_t)b * a.y_brows + a.y_row0 + ty;
        yrow_n[nf] = (int)yrow;
        if (res) {
          float rr[4];
          load4(res + yrow * a.ld_res + co, rr);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += rr[r];
        }
        if (a.row_scale) {
          const float s = a.row_scale[yrow];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= s;
        }
        if (a.y_f32) store4(yF + yrow * a.ld_y + co, v);
        else store4(yT + yrow * a.ld_y + co, v);
        rs_sum[nf] += (v[0] + v[1]) + (v[2] + v[3]);
        rs_sq[nf] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        if (a.out_gn_stats) {
          if (a.nb == 1) {
            gs[0] += v[0] + v[1]; gq[0] += v[0] * v[0] + v[1] * v[1];
            gs[1] += v[2] + v[3]; gq[1] += v[2] * v[2] + v[3] * v[3];
          } else {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
              const int fg = (co + 2 * p) / a.out_cpf;
              atomicAdd(&st_lds[(n_b[nf] * JEN1_FINE_GROUPS + fg) * 2], v[2 * p] + v[2 * p + 1]);
              atomicAdd(&st_lds[(n_b[nf] * JEN1_FINE_GROUPS + fg) * 2 + 1], v[2 * p] * v[2 * p] + v[2 * p + 1] * v[2 * p + 1]);
            }
          }
        }
      }
    }
    if (a.out_gn_stats && a.nb == 1) {
      // all 16 columns of a fragment belong to the same batch element: reduce across them
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        float s = gs[p], q = gq[p];
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
          s += __shfl_xor(s, off);
          q += __shfl_xor(q, off);
        }
        if (li == 0 && m_ok) {
          const int fg = (co + 2 * p) / a.out_cpf;
          atomicAdd(&st_lds[fg * 2], s);
          atomicAdd(&st_lds[fg * 2 + 1], q);
        }
      }
    }
  }
  if (a.out_rowstats) {
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      float s = rs_sum[nf], q = rs_sq[nf];
      s += __shfl_xor(s, 16); q += __shfl_xor(q, 16);
      s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
      if (lg == 0 && yrow_n[nf] >= 0) {
        unsafeAtomicAdd(a.out_rowstats + (size_t)yrow_n[nf] * 2, s);
        unsafeAtomicAdd(a.out_rowstats + (size_t)yrow_n[nf] * 2 + 1, q);
      }
    }
  }
  if (a.out_gn_stats) {
    __syncthreads();
    for (int i = tid; i < a.nb * 64; i += 256) {
      const int b = b0 + i / 64;
      const float v = st_lds[i];
      if (b < a.B && v != 0.f) unsafeAtomicAdd(a.out_gn_stats + (size_t)b * 64 + (i & 63), v);
    }
  }
}

template <typename T, int MF, int NF, int WM, int WN, int PF>
int launch(const jen1_conv_args& a, hipStream_t s) {
  constexpr int BM = 16 * MF * WM;
  const Layout L = make_layout(a, (int)sizeof(T));
  auto kern = conv_gemm_kernel<T, MF, NF, WM, WN, PF>;
  static bool attr_set = false;
  if (!attr_set) {
    JEN1_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const int tiles_t = (a.L_out + a.tb - 1) / a.tb;
  const int tiles_b = (a.B + a.nb - 1) / a.nb;
  dim3 grid((a.M + BM - 1) / BM, tiles_t * tiles_b, a.splitk);
  hipLaunchKernelGGL(kern, grid, dim3(256), L.total, s, a);
  JEN1_HIP(hipGetLastError());
  return 0;
}

template <typename T>
int dispatch(const jen1_conv_args& a, hipStream_t s) {
  switch (a.cfg) {
    case JEN1_CFG_128x128: return launch<T, 4, 4, 2, 2, 1>(a, s);
    case JEN1_CFG_128x64: return launch<T, 2, 4, 4, 1, 1>(a, s);
    case JEN1_CFG_64x64: return launch<T, 1, 4, 4, 1, 2>(a, s);
    case JEN1_CFG_64x32: return launch<T, 1, 2, 4, 1, 4>(a, s);
    case JEN1_CFG_64x16: return launch<T, 1, 1, 4, 1, 4>(a, s);
  }
  return jen1_set_error("jen1_conv_gemm: unknown cfg %d", a.cfg);
}

}  // namespace

extern "C" int jen1_cfg_bm(int cfg) {
  switch (cfg) {
    case JEN1_CFG_128x128: case JEN1_CFG_128x64: return 128;
    case JEN1_CFG_64x64: case JEN1_CFG_64x32: case JEN1_CFG_64x16: return 64;
  }
  return -1;
}
extern "C" int jen1_cfg_bn(int cfg) {
  switch (cfg) {
    case JEN1_CFG_128x128: return 128;
    case JEN1_CFG_128x64: case JEN1_CFG_64x64: return 64;
    case JEN1_CFG_64x32: return 32;
    case JEN1_CFG_64x16: return 16;
  }
  return -1;
}

static int validate(const jen1_conv_args& a) {
  JEN1_CHECK(a.dtype == JEN1_F32 || a.dtype == JEN1_BF16, "conv_gemm: bad dtype %d", a.dtype);
  JEN1_CHECK(a.x0 && a.w && a.y, "conv_gemm: null x0/w/y");
  JEN1_CHECK(a.c0 > 0 && a.c0 % 32 == 0 && a.c1 >= 0 && a.c1 % 32 == 0, "conv_gemm: c0/c1 must be multiples of 32 (%d,%d)", a.c0, a.c1);
  JEN1_CHECK(a.c1 == 0 || a.x1, "conv_gemm: c1 > 0 without x1");
  JEN1_CHECK(a.ld0 >= a.c0 && a.ld0 % 8 == 0 && (a.c1 == 0 || (a.ld1 >= a.c1 && a.ld1 % 8 == 0)), "conv_gemm: bad ld0/ld1");
  JEN1_CHECK(a.M > 0 && a.M % 16 == 0 && a.out_C > 0 && a.out_C % 16 == 0 && a.M == a.out_C * a.ps_f, "conv_gemm: bad M/out_C/ps_f (%d,%d,%d)", a.M, a.out_C, a.ps_f);
  JEN1_CHECK(a.taps >= 1 && a.stride >= 1 && a.B >= 1 && a.L_in >= 1 && a.L_out >= 1, "conv_gemm: bad geometry");
  JEN1_CHECK(a.ld_y % 4 == 0 && (!a.residual || a.ld_res % 4 == 0), "conv_gemm: ld_y/ld_res must be multiples of 4");
  const int bn = jen1_cfg_bn(a.cfg);
  JEN1_CHECK(bn > 0, "conv_gemm: bad cfg %d", a.cfg);
  JEN1_CHECK(a.tb >= 1 && a.nb >= 1 && a.nb * a.tb <= bn, "conv_gemm: tile nb*tb=%d*%d exceeds BN=%d", a.nb, a.tb, bn);
  JEN1_CHECK(a.kc_stage >= 1 && a.splitk >= 1, "conv_gemm: bad kc_stage/splitk");
  const int kch = (a.c0 + a.c1) / 32;
  JEN1_CHECK(a.splitk <= kch, "conv_gemm: splitk %d > chunks %d", a.splitk, kch);
  {
    const int cps = (kch + a.splitk - 1) / a.splitk;
    JEN1_CHECK((a.splitk - 1) * cps < kch, "conv_gemm: splitk %d leaves an empty K slice (chunks %d)", a.splitk, kch);
    const int st = a.kc_stage < cps ? a.kc_stage : cps;
    // a stage must not straddle the x0/x1 boundary
    JEN1_CHECK(a.c1 == 0 || ((a.c0 / 32) % st == 0 && (a.splitk == 1 || (a.c0 / 32) % cps == 0)), "conv_gemm: stage/split straddles the source boundary");
  }
  JEN1_CHECK(a.splitk == 1 || (a.slab && a.counters), "conv_gemm: split-K needs slab and counters");
  if (a.pro_mode == JEN1_PRO_GN || a.pro_mode == JEN1_PRO_GN_SILU) {
    JEN1_CHECK(a.gn_stats0 && a.gn_gamma && a.gn_beta && a.gn_groups >= 1 && a.gn_groups <= 32 && a.gn_cpg >= 1 && a.gn_count >= 1, "conv_gemm: incomplete GroupNorm prologue");
    JEN1_CHECK(a.c1 == 0 || a.gn_stats1, "conv_gemm: GroupNorm over two sources needs gn_stats1");
    JEN1_CHECK(a.c1 == 0 || a.c0 % a.gn_cpg == 0, "conv_gemm: GroupNorm group straddles the source boundary");
    JEN1_CHECK(!a.film || (a.film_ld > 0 && a.film_C > 0), "conv_gemm: bad FiLM geometry");
  }
  if (a.pro_mode == JEN1_PRO_LN) JEN1_CHECK(a.ln_rowstats && a.ln_C >= 1 && a.c1 == 0 && (!a.ln_gamma || a.ln_beta), "conv_gemm: incomplete LayerNorm prologue");
  JEN1_CHECK(!a.out_gn_stats || (a.out_cpf >= 2 && a.out_cpf % 2 == 0), "conv_gemm: out_cpf must be even");
  const jen1_conv_args& aa = a;
  const Layout L = make_layout(aa, a.dtype == JEN1_F32 ? 4 : 2);
  JEN1_CHECK(L.total <= 160 * 1024, "conv_gemm: LDS request %d B exceeds 160 KiB (tb=%d nb=%d kc_stage=%d)", L.total, a.tb, a.nb, a.kc_stage);
  return 0;
}

extern "C" int64_t jen1_conv_gemm_lds_bytes(const jen1_conv_args* args) {
  if (!args) return -1;
  return make_layout(*args, args->dtype == JEN1_F32 ? 4 : 2).total;
}

extern "C" int jen1_conv_gemm(const jen1_conv_args* args, void* stream) {
  JEN1_CHECK(args != nullptr, "conv_gemm: null args");
  if (int rc = validate(*args)) return rc;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  return args->dtype == JEN1_F32 ? dispatch<float>(*args, s) : dispatch<bf16_t>(*args, s);
}
