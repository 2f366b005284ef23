// Small-N multi-head attention core for gfx950.
//
// Replaces AttentionBase.forward's math path (reference jen1/model/blocks.py:355-380):
//   sim = q k^T * d^-0.5 ; optional causal mask (:315-319, keep j <= i + (Nk - Nq)) ;
//   softmax in float32 (:371) ; out = attn v.
// In this network Nq <= 24 (141 at T=9000) and Nk <= 141, so the score matrix is tiny
// (0.2 % of the FLOPs): one 256-thread workgroup per
// (batch element, head, 32-query chunk) keeps K, then V, for that head in LDS, computes
// the scores on the vector ALU, and normalises every row with one wavefront:
// 64-lane __shfl_xor max / sum reductions.  Padding keys are not masked here -- the
// reference zeroes their K and V rows instead (blocks.py:431-434), which the K/V
// producer does through its row_scale epilogue.
#include "common.h"

#ifdef JEN1_PROFILE
__device__ unsigned long long* g_attn_dbg = nullptr;
extern "C" int jen1_debug_set_attention_buffer(void* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_dbg), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#define AT_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && g_attn_dbg) g_attn_dbg[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define AT_STAMP(i) do { } while (0)
#endif

namespace {

constexpr int QCHUNK = 32;     // queries per workgroup: two 16-row MFMA tiles
constexpr int FMAX = 2;        // K/V vectors per thread that may carry a deferred LayerNorm finish (self-attention: Nk <= 24)
constexpr int MAXV = 9;        // 8-element vectors per thread for one K or V tile: ceil(141*128/8/256) = 9

template <typename T> struct AFrag;
template <> struct AFrag<bf16_t> { typedef bf16x8 type; };
template <> struct AFrag<float> { typedef f32x8 type; };
template <> struct AFrag<fp8_t> { typedef long type; };
__device__ __forceinline__ void amma(f32x4& acc, const long& a, const long& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void lds_frag(long& f, const fp8_t* p) { f = *reinterpret_cast<const long*>(p); }
// a raw Q / K vector into its LDS tile: a copy when the tile has the tensors' type, a conversion in JEN1_FP8 mode
__device__ __forceinline__ void stage_vec(bf16_t* dst, const bf16x8& v) { *reinterpret_cast<bf16x8*>(dst) = v; }
__device__ __forceinline__ void stage_vec(float* dst, const f32x8& v) { *reinterpret_cast<f32x8*>(dst) = v; }
__device__ __forceinline__ void stage_vec(fp8_t* dst, const bf16x8& v) {
  float x[8];
  vec_to_float(v, x);
  store8(dst, x);
}
__device__ __forceinline__ void amma(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void amma(f32x4& acc, const f32x8& a, const f32x8& b) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], acc, 0, 0, 0);
}
__device__ __forceinline__ void lds_frag(bf16x8& f, const bf16_t* p) { f = *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ void lds_frag(f32x8& f, const float* p) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
  f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
}

// reductions over the 16 lanes of a DPP row without touching LDS: xor 1, xor 2 (quad permutes), then the two mirrors
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));      // quad_perm [1,0,3,2]
  v = fmaxf(v, dpp_f<0x4E>(v));      // quad_perm [2,3,0,1]
  v = fmaxf(v, dpp_f<0x141>(v));     // row_half_mirror
  v = fmaxf(v, dpp_f<0x140>(v));     // row_mirror
  return v;
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f<0xB1>(v);
  v += dpp_f<0x4E>(v);
  v += dpp_f<0x141>(v);
  v += dpp_f<0x140>(v);
  return v;
}

// One 256-thread workgroup per (batch element, head, 32-query chunk).
//   * every Q, K and V vector of the tile (and the operands of a deferred LayerNorm finish) is requested up front;
//     with <= 128 workgroups there is one wave per SIMD, so a load-use loop would cost a memory latency per trip;
//   * Q K^T and P V run on the matrix cores (v_mfma_f32_16x16x32_bf16, or 16x16x4_f32 in the float32 parity mode:
//     exact fp32 products) from LDS tiles: Q [32][d], K [Nk][d], then V transposed [d][Nk] in K's place;
//   * softmax in float32, one wavefront per row, 64-lane __shfl_xor max / sum (blocks.py:367-371).
// ST: what the matrix cores read from LDS (T itself; fp8_t in JEN1_FP8 mode: Q K^T and P V on e4m3 operands, float32 softmax kept --
// blocks.py:355-380 -- with the probabilities stored as 256 p and the factor taken out of the accumulator)
template <typename T, typename ST>
__global__ __launch_bounds__(256) void attention_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                         const T* __restrict__ v, T* __restrict__ out,
                                                         const int32_t* __restrict__ kv_row,
                                                         const T* __restrict__ kv_extra,
                                                         const int32_t* __restrict__ extra_row,
                                                         const int32_t* __restrict__ extra_step, int ld_extra,
                                                         int kx_off, int vx_off, int H, int d, int Nq,
                                                         int Nk, int ldq, int q_off, int ldkv, int k_off, int v_off,
                                                         int ldo, int causal, float scale,
                                                         const float* __restrict__ fin_stats, const float* __restrict__ fin_u,
                                                         const float* __restrict__ fin_b, float fin_inv_c, float fin_eps,
                                                         int fin_q, int fin_kv, float inv_H, int log2_vpr) {
  typedef typename AFrag<ST>::type Frag;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr bool PRECISE = is_f32<T>::value;
  constexpr bool F8 = sizeof(ST) == 1;
  constexpr float PS = F8 ? JEN1_FP8_P_SCALE : 1.0f;
  jen1_prefetch_kernarg<184>();
  AT_STAMP(0);
  const int bh = blockIdx.x;
  const int b = (int)(((float)bh + 0.5f) * inv_H), h = bh - b * H;
  const int q0 = blockIdx.y * QCHUNK;
  const int nq = (Nq - q0 < QCHUNK) ? (Nq - q0) : QCHUNK;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int NKP = (Nk + 31) & ~31;               // keys padded to the MFMA K step of P V
  const int DP = d < 32 ? 32 : d;                // Q / K columns the matrix cores read (zero beyond d)
  const int DC = d < 16 ? 16 : d;                // V^T rows they read
  const int dq = DP + 8;                         // row pitch of Q / K (elements): 16-byte rows, banks spread
  const int vt = NKP + 8;                        // row pitch of V^T and P
  const int sp = NKP + 4;                        // row pitch of the float32 scores (float4 reads in the softmax)
  ST* q_s = reinterpret_cast<ST*>(smem);                             // [32][dq]
  ST* kv_s = q_s + QCHUNK * dq;                                     // K [NKP][dq], later V^T [d][vt]
  const int kv_elems = (NKP * dq > DC * vt) ? NKP * dq : DC * vt;
  float* s_s = reinterpret_cast<float*>(kv_s + ((kv_elems + 15) & ~15));   // [32][sp]
  ST* p_s = reinterpret_cast<ST*>(s_s + ((QCHUNK * sp + 3) & ~3));    // [32][vt]

  const int kvbase = (kv_row ? kv_row[b] : b) * Nk;
  // optional per-step last key row (the time token of the text context, model.py:315-316)
  int xr = (kv_extra && extra_row) ? extra_row[b] : -1;
  if (xr >= 0 && extra_step) xr = extra_step[0];
  const int vpr = 1 << log2_vpr;                 // 8-element vectors per row (d = 8 * vpr)
  const int nkv = Nk * vpr, nqv = nq * vpr;
  const int hd = h * d;
  AT_STAMP(1);

  // ---- issue all loads -----------------------------------------------------------------------
  // Deferred LayerNorm-folded projections (fin_*): the producer GEMM wrote raw = W' x; the row statistics of x
  // were only complete once that launch ended, so the affine finish  rstd_row (raw - mean_row u[col]) + b[col]
  // (blocks.py:427-429) is applied here, on the way into LDS.  Its operands ride along with the Q/K/V loads.
  typedef typename VecOf<T>::type Vec;
  Vec kraw[MAXV], vraw[MAXV], qraw[2];
  float2 kst[FMAX], qst[2];
  float uk[FMAX][8], bk[FMAX][8], uv[FMAX][8], bv[FMAX][8], uq[2][8], bq[2][8];
#pragma unroll
  for (int u = 0; u < MAXV; ++u) {
    const int i = tid + u * 256;
    if (i < nkv) {
      const int r = i >> log2_vpr, c = (i & (vpr - 1)) * 8;
      const bool ex = (xr >= 0 && r == Nk - 1);
      const T* kp = ex ? kv_extra + (size_t)((unsigned)xr * (unsigned)ld_extra + (unsigned)(kx_off + hd + c))
                       : k + (size_t)((unsigned)(kvbase + r) * (unsigned)ldkv + (unsigned)(k_off + hd + c));
      const T* vp = ex ? kv_extra + (size_t)((unsigned)xr * (unsigned)ld_extra + (unsigned)(vx_off + hd + c))
                       : v + (size_t)((unsigned)(kvbase + r) * (unsigned)ldkv + (unsigned)(v_off + hd + c));
      kraw[u] = *reinterpret_cast<const Vec*>(kp);
      vraw[u] = *reinterpret_cast<const Vec*>(vp);
      if (u < FMAX) {
        if (fin_kv) {
          kst[u] = *reinterpret_cast<const float2*>(fin_stats + 2 * (kvbase + r));
          load8(fin_u + k_off + hd + c, uk[u]);
          load8(fin_b + k_off + hd + c, bk[u]);
          load8(fin_u + v_off + hd + c, uv[u]);
          load8(fin_b + v_off + hd + c, bv[u]);
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {                  // QCHUNK * 128 / 8 / 256 = 2 vectors per thread at most
    const int i = tid + u * 256;
    if (i < nqv) {
      const int r = i >> log2_vpr, c = (i & (vpr - 1)) * 8;
      qraw[u] = *reinterpret_cast<const Vec*>(q + (size_t)((unsigned)(b * Nq + q0 + r) * (unsigned)ldq + (unsigned)(q_off + hd + c)));
      if (fin_q) {
        qst[u] = *reinterpret_cast<const float2*>(fin_stats + 2 * (b * Nq + q0 + r));
        load8(fin_u + q_off + hd + c, uq[u]);
        load8(fin_b + q_off + hd + c, bq[u]);
      }
    }
  }
  AT_STAMP(2);
  auto finish = [&](float (&x)[8], const float2 st, const float (&uu)[8], const float (&bb)[8]) {
    const float mean = st.x * fin_inv_c;
    float var = st.y * fin_inv_c - mean * mean;
    var = var < 0.f ? 0.f : var;
    const float rstd = PRECISE ? 1.0f / sqrtf(var + fin_eps) : rsqrtf(var + fin_eps);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (x[e] - mean * uu[e]) * rstd + bb[e];
  };
  // ---- zero the padding the matrix cores will read: Q rows >= nq, K rows >= Nk (and columns >= d of a narrow head) ----
  {
    const float z8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (d < 32) {
      for (int i = tid; i < (QCHUNK + NKP) * (dq >> 3); i += 256) store8(q_s + i * 8, z8);     // Q and K tiles are contiguous
      __syncthreads();
    } else {
      for (int i = tid; i < (QCHUNK - nq) * vpr; i += 256) store8(q_s + (nq + (i >> log2_vpr)) * dq + (i & (vpr - 1)) * 8, z8);
      for (int i = tid; i < (NKP - Nk) * vpr; i += 256) store8(kv_s + (Nk + (i >> log2_vpr)) * dq + (i & (vpr - 1)) * 8, z8);
    }
  }
  // ---- K, Q -> LDS ------------------------------------------------------------------------------
#pragma unroll
  for (int u = 0; u < MAXV; ++u) {
    const int i = tid + u * 256;
    if (i < nkv) {
      const int r = i >> log2_vpr, c = (i & (vpr - 1)) * 8;
      bool raw_copy = true;
      if (u < FMAX) {
        if (fin_kv) {
          float x[8];
          vec_to_float(kraw[u], x);
          finish(x, kst[u], uk[u], bk[u]);
          store8(kv_s + r * dq + c, x);
          raw_copy = false;
        }
      }
      if (raw_copy) stage_vec(kv_s + r * dq + c, kraw[u]);
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int i = tid + u * 256;
    if (i < nqv) {
      const int r = i >> log2_vpr, c = (i & (vpr - 1)) * 8;
      if (fin_q) {
        float x[8];
        vec_to_float(qraw[u], x);
        finish(x, qst[u], uq[u], bq[u]);
        store8(q_s + r * dq + c, x);
      } else {
        stage_vec(q_s + r * dq + c, qraw[u]);
      }
    }
  }
  __syncthreads();
  AT_STAMP(3);
  // ---- scores on the matrix cores: wave w takes key tiles w, w + 4, ... for both query tiles --------------------
  {
    const int nkt = NKP >> 4;
    const int nqt = (nq + 15) >> 4;
    for (int kt = wave; kt < nkt; kt += 4) {
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      for (int c = 0; c < DP; c += 32) {
        Frag kb;
        lds_frag(kb, kv_s + (kt * 16 + li) * dq + c + lg * 8);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          if (qt < nqt) {
            Frag qa;
            lds_frag(qa, q_s + (qt * 16 + li) * dq + c + lg * 8);
            amma(acc[qt], qa, kb);
          }
        }
      }
      // lane (li, lg) holds rows 4 lg + r (queries), column li (key)
      const int j = kt * 16 + li;
      const float NEG = -3.402823466e+38f;
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        if (qt < nqt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int qi = qt * 16 + lg * 4 + r;
            const int lim = (q0 + qi) + (Nk - Nq);      // causal: keep j <= i + (Nk - Nq)  (blocks.py:315-319)
            s_s[qi * sp + j] = (causal && j > lim) ? NEG : acc[qt][r] * scale;
          }
        }
      }
    }
  }
  __syncthreads();
  AT_STAMP(4);
  // ---- V (already in registers) replaces K in LDS, transposed: V^T [d][keys] is the B operand of P V ------------
  {
    if (d < 16) {
      for (int i = tid; i < DC * vt; i += 256) kv_s[i] = to_elem<ST>(0.f);
      __syncthreads();
    } else {
      for (int i = tid; i < d * 32; i += 256) {                    // keys >= Nk contribute nothing (NKP - Nk < 32)
        const int c = i >> 5, j = Nk + (i & 31);
        if (j < NKP) kv_s[c * vt + j] = to_elem<ST>(0.f);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < MAXV; ++u) {
    const int i = tid + u * 256;
    if (i < nkv) {
      const int r = i >> log2_vpr, c = (i & (vpr - 1)) * 8;
      float vv[8];
      vec_to_float(vraw[u], vv);
      if (u < FMAX) {
        if (fin_kv) finish(vv, kst[u], uv[u], bv[u]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) kv_s[(c + e) * vt + r] = to_elem<ST>(vv[e]);
    }
  }
  AT_STAMP(7);
  // ---- softmax in float32 (blocks.py:367-371): 16 lanes per row, four rows per wavefront at a time; every lane owns
  // a contiguous run of keys (up to three float4 reads, all issued together), reductions are four DPP steps inside
  // the 16-lane row -- no serial LDS round trips.  P is written in the operand dtype.
  {
    const int sub = lane & 15, rsel = lane >> 4;
    const int kpl = ((NKP >> 4) + 3) & ~3;                        // keys per lane: 4, 8 or 12 (NKP <= 192)
    const int j0 = sub * kpl;
#pragma unroll 1
    for (int it = 0; it < QCHUNK / 16; ++it) {
      const int r = it * 16 + wave * 4 + rsel;
      ST* pr = p_s + r * vt;
      if (r >= nq) {                                              // padding rows of the query tiles
        for (int j = sub; j < NKP; j += 16) pr[j] = to_elem<ST>(0.f);
      } else {
        const float* sr = s_s + r * sp + j0;
        float4 x[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) x[t] = (4 * t < kpl && j0 + 4 * t < NKP) ? *reinterpret_cast<const float4*>(sr + 4 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
        float e[12];
#pragma unroll
        for (int t = 0; t < 3; ++t) { e[4 * t] = x[t].x; e[4 * t + 1] = x[t].y; e[4 * t + 2] = x[t].z; e[4 * t + 3] = x[t].w; }
        float m = -3.402823466e+38f;
#pragma unroll
        for (int t = 0; t < 12; ++t) {
          const bool in = t < kpl && j0 + t < Nk;
          e[t] = in ? e[t] : -3.402823466e+38f;
          m = fmaxf(m, e[t]);
        }
        m = row16_max(m);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 12; ++t) {
          const bool in = t < kpl && j0 + t < Nk;
          e[t] = in ? (PRECISE ? expf(e[t] - m) : __expf(e[t] - m)) : 0.f;
          sum += e[t];
        }
        sum = row16_sum(sum);
        const float inv = (PRECISE ? 1.0f / sum : __builtin_amdgcn_rcpf(sum)) * PS;
#pragma unroll
        for (int t = 0; t < 12; ++t) {
          if (t < kpl && j0 + t < NKP) pr[j0 + t] = to_elem<ST>(e[t] * inv);
        }
      }
    }
  }
  __syncthreads();
  AT_STAMP(5);
  // ---- out = P V on the matrix cores: tiles (query tile, 16 channels) dealt to the waves ---------------------------
  {
    const int nqt = (nq + 15) >> 4;
    const int nct = DC >> 4;
    for (int t = wave; t < nqt * nct; t += 4) {
      const int qt = t / nct, ct = t - qt * nct;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int j = 0; j < NKP; j += 32) {
        Frag pa, vb;
        lds_frag(pa, p_s + (qt * 16 + li) * vt + j + lg * 8);
        lds_frag(vb, kv_s + (ct * 16 + li) * vt + j + lg * 8);
        amma(acc, pa, vb);
      }
      // lane holds rows 4 lg + r (queries), column li (channel)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = qt * 16 + lg * 4 + r;
        if (qi < nq && ct * 16 + li < d) out[(size_t)((unsigned)(b * Nq + q0 + qi) * (unsigned)ldo + (unsigned)(hd + ct * 16 + li))] = (T)(F8 ? acc[r] * (1.0f / JEN1_FP8_P_SCALE) : acc[r]);
      }
    }
  }
  AT_STAMP(6);
}

}  // namespace

extern "C" int jen1_attention_fin(const void* q, const void* k, const void* v, void* out, const int32_t* kv_row,
                                  const void* kv_extra, const int32_t* extra_row, const int32_t* extra_step, int ld_extra,
                                  int kx_off, int vx_off, int B,
                                  int H, int d, int Nq, int Nk, int ldq, int q_off, int ldkv, int k_off, int v_off,
                                  int ldo, int causal, float scale, const float* ln_rowstats, const float* ln_u, const float* ln_b,
                                  int ln_C, float ln_eps, int finish_q, int finish_kv, int dtype, void* stream) {
  JEN1_CHECK(q && k && v && out, "attention: null pointer");
  JEN1_CHECK(B >= 1 && H >= 1 && d >= 8 && d % 8 == 0 && Nq >= 1 && Nk >= 1, "attention: bad sizes (head dim must be a multiple of 8)");
  JEN1_CHECK(Nk <= 192 && (Nk * (d / 8) + 255) / 256 <= 9 && d <= 128, "attention: Nk=%d d=%d outside the small-N kernel's range", Nk, d);
  JEN1_CHECK(ldq % 8 == 0 && ldkv % 8 == 0 && q_off % 8 == 0 && k_off % 8 == 0 && v_off % 8 == 0 && (!kv_extra || (ld_extra % 8 == 0 && kx_off % 8 == 0 && vx_off % 8 == 0)), "attention: offsets / strides must be multiples of 8 elements");
  JEN1_CHECK(dtype == JEN1_F32 || dtype == JEN1_BF16 || dtype == JEN1_FP8, "attention: bad dtype");
  const bool fin = finish_q || finish_kv;
  JEN1_CHECK(!fin || (ln_rowstats && ln_u && ln_b && ln_C >= 1), "attention: a deferred LayerNorm finish needs rowstats, u, bias and ln_C");
  JEN1_CHECK(!finish_kv || (!kv_row && !kv_extra && Nk * (d / 8) <= 256 * FMAX && Nq == Nk),
             "attention: the K/V finish is for self-attention over at most %d vectors", 256 * FMAX);
  JEN1_CHECK(d == 8 || d == 16 || d == 32 || d == 64 || d == 128, "attention: head dim %d must be 8, 16, 32, 64 or 128", d);
  const size_t es = dtype == JEN1_F32 ? 4 : (dtype == JEN1_FP8 ? 1 : 2);      // LDS tiles (JEN1_FP8: q / k / v / out are bf16 in memory)
  const int DP = d < 32 ? 32 : d, DC = d < 16 ? 16 : d;
  const int NKP = (Nk + 31) & ~31, dq = DP + 8, vt = NKP + 8, sp = NKP + 4;
  const size_t kv_elems = (size_t)((NKP * dq > DC * vt) ? NKP * dq : DC * vt);
  const size_t lds = es * (size_t)QCHUNK * dq + es * ((kv_elems + 15) & ~(size_t)15) + sizeof(float) * (((size_t)QCHUNK * sp + 3) & ~(size_t)3) + es * (size_t)QCHUNK * vt;
  JEN1_CHECK(lds <= 160 * 1024, "attention: Nk=%d d=%d needs %zu B of LDS (> 160 KiB)", Nk, d, lds);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(B * H, (Nq + QCHUNK - 1) / QCHUNK);
  const float inv_c = fin ? 1.0f / (float)ln_C : 0.f;
  const float inv_H = 1.0f / (float)H;
  int log2_vpr = 0;
  while ((8 << log2_vpr) < d) ++log2_vpr;
  if (dtype == JEN1_F32) {
    auto kern = attention_kernel<float, float>;
    JEN1_MAX_LDS_ONCE(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, (const float*)q, (const float*)k, (const float*)v, (float*)out,
                       kv_row, (const float*)kv_extra, extra_row, extra_step, ld_extra, kx_off, vx_off, H, d, Nq, Nk, ldq, q_off, ldkv, k_off, v_off, ldo, causal, scale,
                       ln_rowstats, ln_u, ln_b, inv_c, ln_eps, finish_q, finish_kv, inv_H, log2_vpr);
  } else if (dtype == JEN1_FP8) {
    auto kern = attention_kernel<bf16_t, fp8_t>;
    JEN1_MAX_LDS_ONCE(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                       (bf16_t*)out, kv_row, (const bf16_t*)kv_extra, extra_row, extra_step, ld_extra, kx_off, vx_off, H, d, Nq, Nk, ldq, q_off, ldkv, k_off, v_off, ldo, causal, scale,
                       ln_rowstats, ln_u, ln_b, inv_c, ln_eps, finish_q, finish_kv, inv_H, log2_vpr);
  } else {
    auto kern = attention_kernel<bf16_t, bf16_t>;
    JEN1_MAX_LDS_ONCE(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                       (bf16_t*)out, kv_row, (const bf16_t*)kv_extra, extra_row, extra_step, ld_extra, kx_off, vx_off, H, d, Nq, Nk, ldq, q_off, ldkv, k_off, v_off, ldo, causal, scale,
                       ln_rowstats, ln_u, ln_b, inv_c, ln_eps, finish_q, finish_kv, inv_H, log2_vpr);
  }
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_attention(const void* q, const void* k, const void* v, void* out, const int32_t* kv_row,
                              const void* kv_extra, const int32_t* extra_row, const int32_t* extra_step, int ld_extra,
                              int kx_off, int vx_off, int B,
                              int H, int d, int Nq, int Nk, int ldq, int q_off, int ldkv, int k_off, int v_off,
                              int ldo, int causal, float scale, int dtype, void* stream) {
  return jen1_attention_fin(q, k, v, out, kv_row, kv_extra, extra_row, extra_step, ld_extra, kx_off, vx_off, B, H, d, Nq, Nk, ldq, q_off,
                            ldkv, k_off, v_off, ldo, causal, scale, nullptr, nullptr, nullptr, 0, 0.f, 0, 0, dtype, stream);
}
