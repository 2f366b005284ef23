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
//     [tap][c/32][m/16][lane 0..63][8] so that every wave-instruction streams one
//     contiguous 1 KiB (bf16) / 2 KiB (f32) block from HBM, and the workgroups of a launch
//     (different m tiles, same k chunk at the same time) read one contiguous span that
//     spreads over all memory channels -- the step is weight-streaming bound
//     (SURVEY.md section 8d).
//   * B operand = channel-last activations [B][L][C]; the tile (+ conv halo) is staged
//     ONCE in LDS with the GroupNorm(+FiLM)+SiLU / LayerNorm prologue applied, then
//     every tap reads a row-shifted view of it (no F.pad copy, no cat copy, no
//     separate norm/activation pass).
//   * MFMA: v_mfma_f32_16x16x32_bf16 (bf16 mode) or 8 x v_mfma_f32_16x16x4_f32
//     (float32 parity mode, exact fp32 FMA chain).  64-lane wavefronts, 4 waves / WG.
//   * Two wave layouts: WM=4 (4 waves x 16*MF output rows, K not split) for the wide
//     levels, and WK=4 (one 16-row M tile, the 4 waves split K and reduce through LDS)
//     for the deep, skinny levels where a launch is pure weight streaming: 16-row M
//     tiles give >= 64 workgroups for C_out = 1024 without any inter-workgroup reduction.
//   * latency structure: every independent global load (weight ring, first activation
//     batch, statistics, gamma/beta) is issued before the first barrier; taps that only
//     see zero padding are skipped together with their weights (exact).
//   * epilogue: bias, GELU, residual, row mask, sub-pixel (transposed-conv) row
//     mapping with crop, and the statistics of the NEXT norm layer (GroupNorm
//     fine-group sums, LayerNorm row sums) via LDS + global float atomics.
//   * optional inter-workgroup split-K: partial slabs + agent-scope release / ticket /
//     acquire, last arriver reduces (cdna_hip_programming.md section 5).
#include "common.h"

// Tuning builds only (-DJEN1_PROFILE): workgroup (0,0,0), thread 0 records the constant-rate 100 MHz
// s_memrealtime counter at phase boundaries into args.slab (reused as a debug buffer).
#ifdef JEN1_PROFILE
#define JEN1_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0 && a.slab) \
    reinterpret_cast<unsigned long long*>(a.slab)[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define JEN1_STAMP(i) do { } while (0)
#endif

namespace {

struct Layout {
  int ldsld;      // LDS row pitch in elements (stage channels + 8)
  int seg;        // staged input rows per batch element (all taps live)
  int tile_off, gam_off, bet_off, grp_off, fine_off, row_off, stats_off, red_off, misc_off, total;
};

__host__ __device__ inline int align16(int x) { return (x + 15) & ~15; }

__host__ __device__ inline Layout make_layout(const jen1_conv_args& a, int esize, int red_floats) {
  Layout L;
  const int kch_total = (a.c0 + a.c1) / 32;
  const int cps = (kch_total + a.splitk - 1) / a.splitk;
  const int stage_ch = 32 * (a.kc_stage < cps ? a.kc_stage : cps);
  L.ldsld = stage_ch + 8;
  L.seg = (a.tb - 1) * a.stride + a.taps;
  int off = 0;
  L.tile_off = off;
  if (!a.direct) off = align16(off + a.nb * L.seg * L.ldsld * esize);
  const bool gn = (a.pro_mode == JEN1_PRO_GN || a.pro_mode == JEN1_PRO_GN_SILU);
  const bool tabs = gn || (a.pro_mode == JEN1_PRO_LN && a.ln_gamma);
  L.gam_off = off;
  if (tabs) off = align16(off + a.nb * stage_ch * 4);
  L.bet_off = off;
  if (tabs) off = align16(off + a.nb * stage_ch * 4);
  L.grp_off = off;
  if (gn) off = align16(off + a.nb * JEN1_FINE_GROUPS * 2 * 4);
  L.fine_off = off;
  if (gn) off = align16(off + a.nb * 128 * 4);
  L.row_off = off;
  if (a.pro_mode == JEN1_PRO_LN) off = align16(off + a.nb * L.seg * 2 * 4);
  L.stats_off = off;
  if (a.out_gn_stats) off = align16(off + a.nb * JEN1_FINE_GROUPS * 2 * 4);
  L.red_off = off;
  off = align16(off + red_floats * 4);
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
__device__ __forceinline__ void frag_zero(f32x8& f) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = 0.f;
}
__device__ __forceinline__ void frag_zero(bf16x8& f) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (bf16_t)0.f;
}
__device__ __forceinline__ void frag_to_float(const f32x8& f, float (&o)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = f.v[j];
}
__device__ __forceinline__ void frag_to_float(const bf16x8& f, float (&o)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (float)f[j];
}

__device__ __forceinline__ void float_to_frag(f32x8& f, const float (&o)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = o[j];
}
__device__ __forceinline__ void float_to_frag(bf16x8& f, const float (&o)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (bf16_t)o[j];
}

constexpr int VB = 4;   // activation vectors (8 channels each) per thread per staging batch

template <typename T, int MF, int NF, int WM, int WK, int PF>
__global__ __launch_bounds__(64 * WM * WK) void conv_gemm_kernel(const jen1_conv_args a) {
  typedef typename FragOf<T>::type Frag;
  constexpr int NT = 64 * WM * WK;     // threads per workgroup
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int BM = 16 * MF * WM;
  constexpr bool PRECISE = is_f32<T>::value;
  constexpr int RED_FLOATS = (WK > 1) ? (WK - 1) * WM * MF * NF * 256 : 0;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wk = wave / WM;
  const int li = lane & 15, lg = lane >> 4;

  jen1_prefetch_kernarg<sizeof(jen1_conv_args)>();
  JEN1_STAMP(0);
  const Layout L = make_layout(a, (int)sizeof(T), RED_FLOATS);
  T* tile = reinterpret_cast<T*>(smem + L.tile_off);
  float* gam_s = reinterpret_cast<float*>(smem + L.gam_off);
  float* bet_s = reinterpret_cast<float*>(smem + L.bet_off);
  float* grp = reinterpret_cast<float*>(smem + L.grp_off);
  float* fine = reinterpret_cast<float*>(smem + L.fine_off);
  float* rowtab = reinterpret_cast<float*>(smem + L.row_off);
  float* st_lds = reinterpret_cast<float*>(smem + L.stats_off);
  float* red = reinterpret_cast<float*>(smem + L.red_off);
  int* misc = reinterpret_cast<int*>(smem + L.misc_off);

  // ---- tile coordinates -------------------------------------------------------------------
  const int tiles_t = a.tiles_t;
  const int bt = (int)(((float)blockIdx.y + 0.5f) * a.inv_tiles_t), tt = blockIdx.y - bt * tiles_t;
  const int b0 = bt * a.nb, t0 = tt * a.tb;
  const int ldsld = L.ldsld;
  const int ctot = a.c0 + a.c1;
  const int kch_total = ctot / 32;
  const int cps = (kch_total + a.splitk - 1) / a.splitk;
  const int z = blockIdx.z;
  const int kc_begin = z * cps;
  const int kc_end = (kc_begin + cps < kch_total) ? kc_begin + cps : kch_total;
  const int MT = a.M / 16;
  const int mt_base = blockIdx.x * (BM / 16) + wm * MF;
  const int n_rows = a.nb * a.tb;

  // live taps: a tap whose input rows are all zero padding for every position of this tile
  // contributes nothing -- skip it and its weights (exact).  At T' = 1 this drops 2/3 of a k=3 conv.
  int tap_lo = 0, tap_hi = a.taps - 1;
  {
    const int t_last = ((t0 + a.tb < a.L_out) ? t0 + a.tb : a.L_out) - 1;
    while (tap_lo < tap_hi && t_last * a.stride + tap_lo - a.pad_left < 0) ++tap_lo;
    while (tap_hi > tap_lo && t0 * a.stride + tap_hi - a.pad_left >= a.L_in) --tap_hi;
  }
  const int ntaps = tap_hi - tap_lo + 1;
  const int seg = (a.tb - 1) * a.stride + ntaps;
  const int tin0 = t0 * a.stride + tap_lo - a.pad_left;

  // per-lane output column mapping
  int rowbase[NF];
  int n_b[NF], n_t[NF];
  bool n_ok[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int n = nf * 16 + li;
    const int bl = (int)(((float)n + 0.5f) * a.inv_tb), tl = n - bl * a.tb;     // exact for n < 64
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

  const bool owner = (wk == 0);        // waves that finish the tile (hold the reduced accumulators)
  const T* res = reinterpret_cast<const T*>(a.residual);
// ---- epilogue operands (bias, residual, row mask) are requested NOW, before the main loop: with one
// wave per SIMD every load-use pair that is not overlapped costs a full memory latency at the end
  bool okk[MF][NF];
  size_t yrow[MF][NF];
  int co_m[MF];
  bool m_okk[MF];
  float bias4[MF][4];
  float rr[MF][NF][4];
  float rsc[MF][NF];
  float lnu[MF][4];          // ln_fold: row sums of the folded weights
  float2 lnrs[NF];           // ln_fold: (sum, sumsq) of the input row behind output column nf
  if (owner) {
    if (a.ln_fold) {
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const size_t irow = n_ok[nf] ? (size_t)(b0 + n_b[nf]) * a.L_in + t0 + n_t[nf] : 0;
        lnrs[nf] = *reinterpret_cast<const float2*>(a.ln_rowstats + irow * 2);
      }
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        int mt = mt_base + mf;
        mt = mt < MT ? mt : MT - 1;
        const float4 uu = *reinterpret_cast<const float4*>(a.ln_u + mt * 16 + lg * 4);
        lnu[mf][0] = uu.x; lnu[mf][1] = uu.y; lnu[mf][2] = uu.z; lnu[mf][3] = uu.w;
      }
    }
  #pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const int mt = mt_base + mf;
      const int m = mt * 16 + lg * 4;
      m_okk[mf] = mt < MT;
      const int ph = (m_okk[mf] && a.ps_f > 1) ? m / a.out_C : 0;
      co_m[mf] = m_okk[mf] ? m - ph * a.out_C : 0;
  #pragma unroll
      for (int r = 0; r < 4; ++r) bias4[mf][r] = 0.f;
      if (a.bias) {
        const float4 bb = *reinterpret_cast<const float4*>(a.bias + co_m[mf]);
        bias4[mf][0] = bb.x; bias4[mf][1] = bb.y; bias4[mf][2] = bb.z; bias4[mf][3] = bb.w;
      }
  #pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const int ty = (t0 + n_t[nf]) * a.ps_f + ph - a.ps_off;
        okk[mf][nf] = m_okk[mf] && n_ok[nf] && ty >= 0 && ty < a.L_y;
        yrow[mf][nf] = okk[mf][nf] ? (size_t)(b0 + n_b[nf]) * a.y_brows + a.y_row0 + ty : 0;
  #pragma unroll
        for (int r = 0; r < 4; ++r) rr[mf][nf][r] = 0.f;
        if (res) load4(res + yrow[mf][nf] * a.ld_res + co_m[mf], rr[mf][nf]);
        rsc[mf][nf] = a.row_scale ? a.row_scale[yrow[mf][nf]] : 1.0f;
      }
    }


  }
  const T* wbase = reinterpret_cast<const T*>(a.w);
  const bool gn = (a.pro_mode == JEN1_PRO_GN || a.pro_mode == JEN1_PRO_GN_SILU);
  const bool ln = (a.pro_mode == JEN1_PRO_LN);
  const bool do_silu = (a.pro_mode == JEN1_PRO_GN_SILU || a.pro_mode == JEN1_PRO_SILU);
  const bool tabs = gn || (ln && a.ln_gamma);
  const int stage_chunks = (ldsld - 8) / 32;

  // ---- helpers ------------------------------------------------------------------------------
  // weight fragment of this wave for (tap, chunk): clamped so the prefetch ring never runs off the end
  auto load_a = [&](Frag(&dst)[MF], int tap, int kc) {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      int mt = mt_base + mf;
      mt = mt < MT ? mt : MT - 1;
      frag_load(dst[mf], wbase + ((size_t)((size_t)tap * kch_total + kc) * MT + mt) * 512 + lane * 8);
    }
  };
  // activation batch: raw vectors (+ FiLM scale/shift vectors) of the staging loop
  struct Batch {
    Frag x[VB];
  };
  auto load_batch = [&](Batch& bt_, int v0, int sch, int cst) {
    const int vpr = sch >> 3;
    const int nvec = a.nb * seg * vpr;
#pragma unroll
    for (int u = 0; u < VB; ++u) {
      const int v = v0 + u * NT + tid;
      const int vv = v < nvec ? v : 0;
      const int row = vv / vpr, cv = vv - row * vpr;
      const int bl = row / seg, r = row - bl * seg;
      const int b = b0 + bl, tin = tin0 + r;
      const int c = cst + cv * 8;
      const bool ok = v < nvec && b < a.B && tin >= 0 && tin < a.L_in;
      const size_t grow = (size_t)(ok ? b : 0) * a.L_in + (ok ? tin : 0);
      // branch-free (clamped address + select) so the staging loads are counted, not drained
      const T* p = (c < a.c0) ? reinterpret_cast<const T*>(a.x0) + grow * a.ld0 + c
                              : reinterpret_cast<const T*>(a.x1) + grow * a.ld1 + (c - a.c0);
      frag_load(bt_.x[u], p);
      if (!ok) frag_zero(bt_.x[u]);
    }
  };
  auto store_batch = [&](const Batch& bt_, int v0, int sch, int cst) {
    const int vpr = sch >> 3;
    const int nvec = a.nb * seg * vpr;
#pragma unroll
    for (int u = 0; u < VB; ++u) {
      const int v = v0 + u * NT + tid;
      if (v >= nvec) continue;
      const int row = v / vpr, cv = v - row * vpr;
      const int bl = row / seg, r = row - bl * seg;
      const int b = b0 + bl, tin = tin0 + r;
      const int cl = cv * 8;
      const int c = cst + cl;
      float x[8];
      frag_to_float(bt_.x[u], x);
      const bool ok = (b < a.B) && (tin >= 0) && (tin < a.L_in);   // zero padding is applied AFTER the prologue
      if (ok) {
        if (gn) {
          const float sc = (c >= a.c0) ? a.src1_scale : 1.0f;
          int g0 = c / a.gn_cpg;
          g0 = g0 < a.gn_groups ? g0 : a.gn_groups - 1;
          const bool one_group = (c - g0 * a.gn_cpg + 8 <= a.gn_cpg) || (g0 == a.gn_groups - 1);
          const float* gt = gam_s + bl * sch + cl;     // gamma * (1 + film scale)
          const float* bt2 = bet_s + bl * sch + cl;    // beta * (1 + film scale) + film shift
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            int g = g0;
            if (!one_group) {
              g = (c + j) / a.gn_cpg;
              g = g < a.gn_groups ? g : a.gn_groups - 1;
            }
            const float mean = grp[2 * (bl * a.gn_groups + g)], rstd = grp[2 * (bl * a.gn_groups + g) + 1];
            const float A = rstd * gt[j];
            x[j] = x[j] * (A * sc) + (bt2[j] - mean * A);
          }
        } else {
          if (c >= a.c0 && a.src1_scale != 1.0f) {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] *= a.src1_scale;
          }
          if (ln) {
            const float mean = rowtab[2 * row], rstd = rowtab[2 * row + 1];
            if (a.ln_gamma) {
#pragma unroll
              for (int j = 0; j < 8; ++j) x[j] = (x[j] - mean) * rstd * gam_s[bl * sch + cl + j] + bet_s[bl * sch + cl + j];
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j) x[j] = (x[j] - mean) * rstd;
            }
          }
        }
        if (do_silu) {
#pragma unroll
          for (int j = 0; j < 8; ++j) x[j] = PRECISE ? silu_precise(x[j]) : silu_f(x[j]);
        }
      }
      store8(tile + (size_t)row * ldsld + cl, x);
    }
  };

  JEN1_STAMP(1);
  // ==== phase 0: issue every independent global load =========================================
  // (a) weight ring of the first stage
  Frag ring[PF][MF];
  int p_tap = 0, p_j = 0;                 // prefetch cursor: (live tap index, this wave's chunk index)
  int nch = (kc_begin + stage_chunks <= kc_end) ? stage_chunks : (kc_end - kc_begin);
  int nmy = (nch - wk + WK - 1) / WK;     // chunks of this wave in the stage: kcl = wk + WK*j
  nmy = nmy > 0 ? nmy : 0;
  auto ring_fill = [&](int ks, int nmy_) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int jj = p_j < nmy_ ? p_j : (nmy_ > 0 ? nmy_ - 1 : 0);
      load_a(ring[u], tap_lo + p_tap, ks + wk + WK * jj < kc_end ? ks + wk + WK * jj : kc_end - 1);
      if (++p_j >= nmy_) { p_j = 0; if (p_tap + 1 < ntaps) ++p_tap; else p_j = nmy_ > 0 ? nmy_ - 1 : 0; }
    }
  };
  ring_fill(kc_begin, nmy);
  // (b) first activation batch of the first stage
  Batch cur;
  load_batch(cur, 0, nch * 32, kc_begin * 32);
  // (c) small tables: GroupNorm group statistics, LayerNorm row statistics.  The fine-group sums are
  // fetched with ONE load per thread (all in flight together) and merged from LDS: a per-group loop of
  // dependent global loads would cost one memory latency per fine group.
  if (gn) {
    const int G = a.gn_groups;
    for (int i = tid; i < a.nb * 64; i += NT) {
      const int bl = i >> 6, src = (i >> 5) & 1, f = i & 31;
      const int b = b0 + bl;
      const float* st = src ? a.gn_stats1 : a.gn_stats0;
      float2 v = make_float2(0.f, 0.f);
      if (st && b < a.B) v = *reinterpret_cast<const float2*>(st + (size_t)b * 64 + 2 * f);
      fine[2 * i] = v.x;
      fine[2 * i + 1] = v.y;
    }
    __syncthreads();
    for (int i = tid; i < a.nb * G; i += NT) {
      const int bl = i / G, g = i - bl * G;
      const int b = b0 + bl;
      float mean = 0.f, rstd = 0.f;
      if (b < a.B) {
        int lo = g * a.gn_cpg, hi = (g == G - 1) ? ctot : lo + a.gn_cpg;
        int src = 0, cpf;
        float sc = 1.f;
        if (lo >= a.c0) {
          src = 1;
          lo -= a.c0; hi -= a.c0;
          cpf = a.c1 / JEN1_FINE_GROUPS;
          sc = a.src1_scale;
        } else {
          cpf = a.c0 / JEN1_FINE_GROUPS;
          if (hi > a.c0) hi = a.c0;
        }
        float s = 0.f, q = 0.f;
        const int f0 = lo / cpf, f1 = (hi + cpf - 1) / cpf;
        for (int f = f0; f < f1; ++f) {
          s += fine[2 * ((bl * 2 + src) * 32 + f)];
          q += fine[2 * ((bl * 2 + src) * 32 + f) + 1];
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
  if (ln) {
    const float inv_c = 1.0f / (float)a.ln_C;
    for (int i = tid; i < a.nb * seg; i += NT) {
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
    for (int i = tid; i < a.nb * 64; i += NT) st_lds[i] = 0.f;
  }

  JEN1_STAMP(2);
  // ==== K loop over LDS stages ===============================================================
  bool first_stage = true;
  for (int ks = kc_begin; ks < kc_end; ks += stage_chunks) {
    nch = (ks + stage_chunks <= kc_end) ? stage_chunks : (kc_end - ks);
    nmy = (nch - wk + WK - 1) / WK;
    nmy = nmy > 0 ? nmy : 0;
    const int sch = nch * 32;
    const int cst = ks * 32;
    if (!first_stage) {
      __syncthreads();                 // previous stage fully consumed
      p_tap = 0; p_j = 0;
      ring_fill(ks, nmy);
      load_batch(cur, 0, sch, cst);
    }
    if (tabs) {
      const float* gsrc = gn ? a.gn_gamma : a.ln_gamma;
      const float* bsrc = gn ? a.gn_beta : a.ln_beta;
#pragma unroll 4
      for (int i = tid; i < a.nb * sch; i += NT) {
        const int bl = i / sch, cl = i - bl * sch;
        float gv = gsrc[cst + cl], bv = bsrc[cst + cl];
        if (gn && a.film && b0 + bl < a.B) {
          const int fr = a.film_step ? a.film_step[0] : (a.film_row ? a.film_row[b0 + bl] : b0 + bl);
          const float* fp = a.film + (size_t)fr * a.film_ld + a.film_off + cst + cl;
          const float fs = fp[0] + 1.0f;
          gv *= fs;
          bv = bv * fs + fp[a.film_C];
        }
        gam_s[i] = gv;
        bet_s[i] = bv;
      }
    }
    __syncthreads();                   // tables (and st_lds zeroing) visible
    JEN1_STAMP(3);
    {
      const int nvec = a.nb * seg * (sch >> 3);
      for (int v0 = 0; v0 < nvec; v0 += NT * VB) {
        Batch nxt;
        const bool more = v0 + NT * VB < nvec;
        if (more) load_batch(nxt, v0 + NT * VB, sch, cst);
        store_batch(cur, v0, sch, cst);
        if (more) cur = nxt;
      }
    }
    __syncthreads();

    JEN1_STAMP(4);
    // ---- MFMA loop: this wave's (live tap, chunk) pairs, weights from the prefetch ring -------
    const int iters = ntaps * nmy;
    int c_tap = 0, c_j = 0;
    for (int it = 0; it < iters; it += PF) {
      Frag nxt[PF][MF];
      {
        // refill: next PF fragments (clamped at the end of the stage)
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          const int jj = p_j < nmy ? p_j : nmy - 1;
          int kc = ks + wk + WK * jj;
          kc = kc < kc_end ? kc : kc_end - 1;
          load_a(nxt[u], tap_lo + p_tap, kc);
          if (++p_j >= nmy) { p_j = 0; if (p_tap + 1 < ntaps) ++p_tap; else p_j = nmy - 1; }
        }
      }
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        if (it + u < iters) {
          const int kcl = wk + WK * c_j;
          Frag bfr[NF];
#pragma unroll
          for (int nf = 0; nf < NF; ++nf)
            frag_load(bfr[nf], tile + (size_t)(rowbase[nf] + c_tap) * ldsld + kcl * 32 + lg * 8);
#pragma unroll
          for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) mma32(acc[mf][nf], ring[u][mf], bfr[nf]);
          if (++c_j == nmy) { c_j = 0; ++c_tap; }
        }
      }
#pragma unroll
      for (int u = 0; u < PF; ++u)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) ring[u][mf] = nxt[u][mf];
    }
    first_stage = false;
  }

  JEN1_STAMP(5);
  // ==== intra-workgroup K reduction (WK > 1): waves wk > 0 hand their partials to wk == 0 =====
  if (WK > 1) {
    if (wk > 0) {
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
          *reinterpret_cast<float4*>(red + ((size_t)(((wk - 1) * WM + wm) * MF + mf) * NF + nf) * 256 + lane * 4) =
              make_float4(acc[mf][nf][0], acc[mf][nf][1], acc[mf][nf][2], acc[mf][nf][3]);
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int w2 = 1; w2 < WK; ++w2)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) {
            const float4 o = *reinterpret_cast<const float4*>(red + ((size_t)(((w2 - 1) * WM + wm) * MF + mf) * NF + nf) * 256 + lane * 4);
            acc[mf][nf][0] += o.x; acc[mf][nf][1] += o.y; acc[mf][nf][2] += o.z; acc[mf][nf][3] += o.w;
          }
    }
  }

  // ==== inter-workgroup split-K: publish partial slab, last arriver reduces ==================
  // Fence-free form of the hand-off (cdna_hip_programming.md Guideline 16, R1): the partial tile is
  // stored WRITE-THROUGH (relaxed agent-scope 8-byte atomic stores lower to `global_store ... sc1`),
  // every storing wave drains vmcnt, one lane takes a ticket; the last arriver reads the other slabs
  // with sc1 loads (L1 bypass), so neither a release nor an acquire fence is needed.
  if (a.splitk > 1) {
    typedef unsigned long long u64;
    const unsigned tile_id = blockIdx.y * gridDim.x + blockIdx.x;
    constexpr int SLAB = WM * MF * NF * 256;     // floats per (tile, split)
    float* slab = a.slab + ((size_t)tile_id * a.splitk) * (size_t)SLAB;
    float* mine = slab + (size_t)z * SLAB;
    if (owner) {
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          u64* p = reinterpret_cast<u64*>(mine + ((size_t)(wm * MF + mf) * NF + nf) * 256 + lane * 4);
          const u64 lo = ((u64)__float_as_uint(acc[mf][nf][1]) << 32) | __float_as_uint(acc[mf][nf][0]);
          const u64 hi = ((u64)__float_as_uint(acc[mf][nf][3]) << 32) | __float_as_uint(acc[mf][nf][2]);
          __hip_atomic_store(p, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(p + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
      misc[0] = (int)__hip_atomic_fetch_add(a.counters + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int ticket = misc[0];
    if (ticket != a.splitk - 1) return;
    if (tid == 0) __hip_atomic_store(a.counters + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (owner) {
      for (int zz = 0; zz < a.splitk; ++zz) {
        if (zz == z) continue;
        const float* other = slab + (size_t)zz * SLAB;
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) {
            u64* p = reinterpret_cast<u64*>(const_cast<float*>(other) + ((size_t)(wm * MF + mf) * NF + nf) * 256 + lane * 4);
            const u64 lo = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const u64 hi = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc[mf][nf][0] += __uint_as_float((unsigned)lo);
            acc[mf][nf][1] += __uint_as_float((unsigned)(lo >> 32));
            acc[mf][nf][2] += __uint_as_float((unsigned)hi);
            acc[mf][nf][3] += __uint_as_float((unsigned)(hi >> 32));
          }
      }
    }
  }

  JEN1_STAMP(6);
  // ==== epilogue (owner waves) =================================================================
  if (owner) {
    T* yT = reinterpret_cast<T*>(a.y);
    float* yF = reinterpret_cast<float*>(a.y);
    float rs_sum[NF], rs_sq[NF];
    int yrow_n[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) { rs_sum[nf] = 0.f; rs_sq[nf] = 0.f; yrow_n[nf] = -1; }

    // ---- pass 2: finish, store, statistics
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const int co = co_m[mf];
      const bool m_ok = m_okk[mf];
      float gs[2] = {0.f, 0.f}, gq[2] = {0.f, 0.f};   // nb == 1 path: per channel-pair sums over nf
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        float v[4];
        if (a.ln_fold) {
          // Linear(LayerNorm(x)) = rstd * (W'x - mean * rowsum(W')) + W beta   (blocks.py:427-429)
          const float inv_c = 1.0f / (float)a.ln_C;
          const float mean = lnrs[nf].x * inv_c;
          float var = lnrs[nf].y * inv_c - mean * mean;
          var = var < 0.f ? 0.f : var;
          const float rstd = PRECISE ? 1.0f / sqrtf(var + a.ln_eps) : rsqrtf(var + a.ln_eps);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (acc[mf][nf][r] - mean * lnu[mf][r]) * rstd + bias4[mf][r];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[mf][nf][r] + bias4[mf][r];
        }
        if (a.act == JEN1_ACT_GELU) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
        }
        if (okk[mf][nf]) {
          const size_t yr = yrow[mf][nf];
          yrow_n[nf] = (int)yr;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (v[r] + rr[mf][nf][r]) * rsc[mf][nf];
          if (a.y_f32) store4(yF + yr * a.ld_y + co, v);
          else store4(yT + yr * a.ld_y + co, v);
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
  }
  JEN1_STAMP(7);
  if (a.out_gn_stats) {
    __syncthreads();
    for (int i = tid; i < a.nb * 64; i += NT) {
      const int b = b0 + i / 64;
      const float v = st_lds[i];
      if (b < a.B && v != 0.f) unsafeAtomicAdd(a.out_gn_stats + (size_t)b * 64 + (i & 63), v);
    }
  }
  JEN1_STAMP(8);
}

template <int MF, int NF, int WM, int WK>
constexpr int red_floats() { return (WK > 1) ? (WK - 1) * WM * MF * NF * 256 : 0; }

struct CfgDesc {
  int MF, NF, WM, WK;
};
// JEN1_CFG_*: 0 = 64x64 wide, 1 = 128x64 wide, 2..4 = 16-row weight-streaming tiles (K split over the waves)
#ifndef JEN1_SWK
#define JEN1_SWK 4      // waves (= K slices) per streaming workgroup
#endif
constexpr CfgDesc kCfg[JEN1_NUM_CFG] = {{1, 4, 4, 1}, {2, 4, 4, 1}, {1, 4, 1, JEN1_SWK}, {1, 2, 1, JEN1_SWK}, {1, 1, 1, JEN1_SWK},
                                        // T* tiles (tile_gemm.hip): MF, NF with 4 waves along M
                                        {2, 4, 4, 1}, {2, 2, 4, 1}, {2, 1, 4, 1}, {4, 2, 4, 1}, {4, 1, 4, 1}, {1, 4, 4, 1}};
inline bool is_tile_cfg(int cfg) { return cfg >= JEN1_CFG_T128x64 && cfg <= JEN1_CFG_T64x64; }

inline int cfg_red_floats(int cfg) {
  const CfgDesc d = kCfg[cfg];
  return d.WK > 1 ? (d.WK - 1) * d.WM * d.MF * d.NF * 256 : 0;
}

template <typename T, int MF, int NF, int WM, int WK, int PF>
int launch(const jen1_conv_args& a, hipStream_t s) {
  constexpr int BM = 16 * MF * WM;
  const Layout L = make_layout(a, (int)sizeof(T), red_floats<MF, NF, WM, WK>());
  auto kern = conv_gemm_kernel<T, MF, NF, WM, WK, PF>;
  JEN1_MAX_LDS_ONCE(kern, 160 * 1024);
  const int tiles_t = (a.L_out + a.tb - 1) / a.tb;
  const int tiles_b = (a.B + a.nb - 1) / a.nb;
  dim3 grid((a.M + BM - 1) / BM, tiles_t * tiles_b, a.splitk);
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WK), L.total, s, a);
  JEN1_HIP(hipGetLastError());
  return 0;
}

template <typename T>
int dispatch(const jen1_conv_args& a, hipStream_t s) {
  constexpr int PFW = is_f32<T>::value ? 4 : 8;     // prefetch ring depth of the wide layouts
  constexpr int PFS = is_f32<T>::value ? 4 : 6;     // ... of the weight-streaming layouts (LDS-staged B)
  switch (a.cfg) {
    case JEN1_CFG_W64x64: return launch<T, 1, 4, 4, 1, PFW>(a, s);
    case JEN1_CFG_W128x64: return launch<T, 2, 4, 4, 1, PFW>(a, s);
    case JEN1_CFG_S16x64: return launch<T, 1, 4, 1, JEN1_SWK, PFS>(a, s);
    case JEN1_CFG_S16x32: return launch<T, 1, 2, 1, JEN1_SWK, PFS>(a, s);
    case JEN1_CFG_S16x16: return launch<T, 1, 1, 1, JEN1_SWK, PFS>(a, s);
  }
  return jen1_set_error("jen1_conv_gemm: unknown cfg %d", a.cfg);
}

}  // namespace

extern "C" int jen1_cfg_bm(int cfg) {
  if (cfg < 0 || cfg >= JEN1_NUM_CFG) return -1;
  return 16 * kCfg[cfg].MF * kCfg[cfg].WM;
}
extern "C" int jen1_cfg_bn(int cfg) {
  if (cfg < 0 || cfg >= JEN1_NUM_CFG) return -1;
  return 16 * kCfg[cfg].NF;
}

static int validate(const jen1_conv_args& a) {
  JEN1_CHECK(a.dtype == JEN1_F32 || a.dtype == JEN1_BF16, "conv_gemm: bad dtype %d", a.dtype);
  JEN1_CHECK(a.w && a.y, "conv_gemm: null w/y");
  JEN1_CHECK(a.nseg >= 0 && a.nseg <= JEN1_MAX_SEG, "conv_gemm: bad nseg %d", a.nseg);
  // (for the T* tile configurations ``seg`` lists only raw EXTRA segments behind the taps: checked by jen1_tile_gemm_launch)
  const bool tile_extras = is_tile_cfg(a.cfg) && a.nseg > 0;
  JEN1_CHECK(a.nseg == 0 || a.direct || tile_extras, "conv_gemm: explicit K segments need direct mode");
  int kch = 0;       // 32-channel chunks the K split operates on
  if (a.nseg > 0 && !tile_extras) {
    for (int s = 0; s < a.nseg; ++s) {
      const jen1_conv_seg& g = a.seg[s];
      JEN1_CHECK(g.x && g.kch >= 1 && g.ld >= 32 * g.kch && g.ld % 8 == 0, "conv_gemm: bad K segment %d (kch=%d ld=%d)", s, g.kch, g.ld);
      kch += g.kch;
    }
  } else {
    JEN1_CHECK(a.x0, "conv_gemm: null x0");
    JEN1_CHECK(a.c0 > 0 && a.c0 % 32 == 0 && a.c1 >= 0 && a.c1 % 32 == 0, "conv_gemm: c0/c1 must be multiples of 32 (%d,%d)", a.c0, a.c1);
    JEN1_CHECK(a.c1 == 0 || a.x1, "conv_gemm: c1 > 0 without x1");
    JEN1_CHECK(a.ld0 >= a.c0 && a.ld0 % 8 == 0 && (a.c1 == 0 || (a.ld1 >= a.c1 && a.ld1 % 8 == 0)), "conv_gemm: bad ld0/ld1");
    JEN1_CHECK(a.taps >= 1, "conv_gemm: bad taps");
    kch = (a.c0 + a.c1) / 32 * (a.direct ? a.taps : 1);      // direct mode splits the flat (tap, chunk) list
  }
  JEN1_CHECK(a.M > 0 && a.M % 16 == 0 && a.out_C > 0 && a.out_C % 16 == 0 && a.M == a.out_C * a.ps_f, "conv_gemm: bad M/out_C/ps_f (%d,%d,%d)", a.M, a.out_C, a.ps_f);
  JEN1_CHECK(a.stride >= 1 && a.B >= 1 && a.L_in >= 1 && a.L_out >= 1, "conv_gemm: bad geometry");
  JEN1_CHECK(a.ld_y % 4 == 0 && (!a.residual || a.ld_res % 4 == 0), "conv_gemm: ld_y/ld_res must be multiples of 4");
  const int bn = jen1_cfg_bn(a.cfg);
  JEN1_CHECK(bn > 0, "conv_gemm: bad cfg %d", a.cfg);
  JEN1_CHECK(a.tb >= 1 && a.nb >= 1 && a.nb * a.tb <= bn, "conv_gemm: tile nb*tb=%d*%d exceeds BN=%d", a.nb, a.tb, bn);
  JEN1_CHECK(a.kc_stage >= 1 && a.splitk >= 1, "conv_gemm: bad kc_stage/splitk");
  JEN1_CHECK(a.splitk <= kch, "conv_gemm: splitk %d > chunks %d", a.splitk, kch);
  {
    const int cps = (kch + a.splitk - 1) / a.splitk;
    JEN1_CHECK((a.splitk - 1) * cps < kch, "conv_gemm: splitk %d leaves an empty K slice (chunks %d)", a.splitk, kch);
    const int st = a.kc_stage < cps ? a.kc_stage : cps;
    // a stage must not straddle the x0/x1 boundary
    JEN1_CHECK(a.direct || a.c1 == 0 || ((a.c0 / 32) % st == 0 && (a.splitk == 1 || (a.c0 / 32) % cps == 0)), "conv_gemm: stage/split straddles the source boundary");
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
  JEN1_CHECK(!a.direct || a.pro_mode == JEN1_PRO_NONE, "conv_gemm: direct (no-LDS) mode takes no prologue; run jen1_norm_apply first");
  JEN1_CHECK(!a.direct || a.nseg > 0 || a.c1 == 0 || a.src1_scale == 1.0f, "conv_gemm: direct mode needs src1_scale == 1 (fold the scale into the weights)");
  JEN1_CHECK(!a.direct || (a.cfg >= JEN1_CFG_S16x64 && a.cfg <= JEN1_CFG_S16x16), "conv_gemm: direct mode needs a streaming (S16) cfg, got %d", a.cfg);
  JEN1_CHECK(!a.direct || a.ps_f <= 8, "conv_gemm: direct mode supports ps_f <= 8");
  JEN1_CHECK(a.nb * a.tb <= 64 && a.L_out / a.tb + 1 < (1 << 20), "conv_gemm: tile too large for the reciprocal index math");
  JEN1_CHECK(!a.ln_fold || (a.direct && a.ln_u && a.ln_rowstats && a.ln_C >= 1 && (a.nseg > 0 ? (a.nseg == 1 && a.seg[0].shift == 0) : (a.taps == 1 && a.pad_left == 0 && a.c1 == 0)) &&
                            a.stride == 1 && a.L_out == a.L_in && a.ps_f == 1 && a.pro_mode == JEN1_PRO_NONE),
             "conv_gemm: ln_fold needs direct mode, ln_u / ln_rowstats, one unshifted segment and no prologue");
  JEN1_CHECK(a.m_split == 0 || (a.direct && a.m_split % 16 == 0 && a.m_split > 0 && a.m_split < a.M && a.k_split >= 1 && a.k_split <= kch &&
                               a.ps_f == 1 && !a.ln_fold && !a.out_gn_stats),
             "conv_gemm: bad dual-range split (m_split=%d k_split=%d)", a.m_split, a.k_split);
  JEN1_CHECK(!is_tile_cfg(a.cfg) || (!a.direct && a.nb == 1 && (a.ps_f == 1 || a.out_C % jen1_cfg_bm(a.cfg) == 0)),
             "conv_gemm: T* tiles need nb = 1, no direct mode and, for sub-pixel outputs, out_C a multiple of BM");
  if (!a.direct && !is_tile_cfg(a.cfg)) {
    const Layout L = make_layout(a, a.dtype == JEN1_F32 ? 4 : 2, cfg_red_floats(a.cfg));
    JEN1_CHECK(L.total <= 160 * 1024, "conv_gemm: LDS request %d B exceeds 160 KiB (tb=%d nb=%d kc_stage=%d)", L.total, a.tb, a.nb, a.kc_stage);
  }
  return 0;
}

extern "C" int64_t jen1_conv_gemm_lds_bytes(const jen1_conv_args* args) {
  if (!args || args->cfg < 0 || args->cfg >= JEN1_NUM_CFG) return -1;
  return make_layout(*args, args->dtype == JEN1_F32 ? 4 : 2, cfg_red_floats(args->cfg)).total;
}

extern "C" int jen1_conv_gemm(const jen1_conv_args* args, void* stream) {
  JEN1_CHECK(args != nullptr, "conv_gemm: null args");
  if (int rc = validate(*args)) return rc;
  jen1_conv_args a = *args;
  // derived launch constants (kept out of the kernel's serial prologue)
  a.tiles_t = (a.L_out + a.tb - 1) / a.tb;
  a.inv_tiles_t = 1.0f / (float)a.tiles_t;
  a.inv_tb = 1.0f / (float)a.tb;
  if (a.direct) return jen1_stream_gemm_launch(a, stream);
  if (is_tile_cfg(a.cfg)) return jen1_tile_gemm_launch(a, stream);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  return a.dtype == JEN1_F32 ? dispatch<float>(a, s) : dispatch<bf16_t>(a, s);
}
