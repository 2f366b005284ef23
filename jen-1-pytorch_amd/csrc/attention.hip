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

namespace {

constexpr int QCHUNK = 32;
constexpr int FMAX = 2;    // K/V vectors per thread that may carry a deferred LayerNorm finish (self-attention: Nk <= 24)
constexpr int MAXV = 9;    // 8-element vectors per thread for one K or V tile: ceil(141*128/8/256) = 9

// Memory-level parallelism matters more than arithmetic here: with <= 128 workgroups there is one wave
// per SIMD, and a loop that loads and immediately consumes costs one full memory latency per trip.  So
// every Q, K and V vector of the tile is requested up front (V waits in registers while the scores are
// computed on K), and only then does the kernel touch LDS.
template <typename T>
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
                                                         int fin_q, int fin_kv) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr bool PRECISE = is_f32<T>::value;
  jen1_prefetch_kernarg<176>();
  const int bh = blockIdx.x;
  const int b = bh / H, h = bh - b * H;
  const int q0 = blockIdx.y * QCHUNK;
  const int nq = (Nq - q0 < QCHUNK) ? (Nq - q0) : QCHUNK;
  const int tid = threadIdx.x;
  const int dp = d + 1;                          // padded row pitch (floats) -> conflict-free column walks
  float* kv_s = reinterpret_cast<float*>(smem);  // [Nk][dp]  K first, V later
  float* q_s = kv_s + Nk * dp;                   // [QCHUNK][dp]
  float* p_s = q_s + QCHUNK * dp;                // [QCHUNK][Nk]

  const size_t kvbase = (size_t)(kv_row ? kv_row[b] : b) * Nk;
  // optional per-step last key row (the time token of the text context, model.py:315-316)
  int xr = (kv_extra && extra_row) ? extra_row[b] : -1;
  if (xr >= 0 && extra_step) xr = extra_step[0];
  const int vpr = d >> 3;                        // 8-element vectors per row
  const int nkv = Nk * vpr, nqv = nq * vpr;

  // ---- issue all loads -----------------------------------------------------------------------
  // Deferred LayerNorm-folded projections (fin_*): the producer GEMM wrote raw = W' x; the row statistics of x
  // were only complete once that launch ended, so the affine finish  rstd_row (raw - mean_row u[col]) + b[col]
  // (blocks.py:427-429) is applied here, on the way into LDS.  Its operands ride along with the Q/K/V loads.
  float kreg[MAXV][8], qreg[2][8];
  typename VecOf<T>::type vraw[MAXV];
  float2 kst[FMAX], qst[2];
  float uk[FMAX][8], bk[FMAX][8], uv[FMAX][8], bv[FMAX][8], uq[2][8], bq[2][8];
#pragma unroll
  for (int u = 0; u < MAXV; ++u) {
    const int i = tid + u * 256;
    const int ii = i < nkv ? i : 0;
    const int r = ii / vpr, c = (ii - r * vpr) * 8;
    const bool ex = (xr >= 0 && r == Nk - 1);
    const T* kp = ex ? kv_extra + (size_t)xr * ld_extra + kx_off + h * d + c : k + (kvbase + r) * ldkv + k_off + h * d + c;
    const T* vp = ex ? kv_extra + (size_t)xr * ld_extra + vx_off + h * d + c : v + (kvbase + r) * ldkv + v_off + h * d + c;
    load8(kp, kreg[u]);
    vraw[u] = *reinterpret_cast<const typename VecOf<T>::type*>(vp);
    if (u < FMAX) {
      if (fin_kv && i < nkv) {
        kst[u] = *reinterpret_cast<const float2*>(fin_stats + 2 * (kvbase + r));
        load8(fin_u + k_off + h * d + c, uk[u]);
        load8(fin_b + k_off + h * d + c, bk[u]);
        load8(fin_u + v_off + h * d + c, uv[u]);
        load8(fin_b + v_off + h * d + c, bv[u]);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {                  // QCHUNK * 128 / 8 / 256 = 2 vectors per thread at most
    const int i = tid + u * 256;
    const int ii = i < nqv ? i : 0;
    const int r = ii / vpr, c = (ii - r * vpr) * 8;
    load8(q + ((size_t)b * Nq + q0 + r) * ldq + q_off + h * d + c, qreg[u]);
    if (fin_q && i < nqv) {
      qst[u] = *reinterpret_cast<const float2*>(fin_stats + 2 * ((size_t)b * Nq + q0 + r));
      load8(fin_u + q_off + h * d + c, uq[u]);
      load8(fin_b + q_off + h * d + c, bq[u]);
    }
  }
  auto finish = [&](float (&x)[8], const float2 st, const float (&uu)[8], const float (&bb)[8]) {
    const float mean = st.x * fin_inv_c;
    float var = st.y * fin_inv_c - mean * mean;
    var = var < 0.f ? 0.f : var;
    const float rstd = PRECISE ? 1.0f / sqrtf(var + fin_eps) : rsqrtf(var + fin_eps);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (x[e] - mean * uu[e]) * rstd + bb[e];
  };
  // ---- K, Q -> LDS ------------------------------------------------------------------------------
#pragma unroll
  for (int u = 0; u < MAXV; ++u) {
    const int i = tid + u * 256;
    if (i < nkv) {
      const int r = i / vpr, c = (i - r * vpr) * 8;
      if (u < FMAX) {
        if (fin_kv) finish(kreg[u], kst[u], uk[u], bk[u]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) kv_s[r * dp + c + e] = kreg[u][e];
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int i = tid + u * 256;
    if (i < nqv) {
      const int r = i / vpr, c = (i - r * vpr) * 8;
      if (fin_q) finish(qreg[u], qst[u], uq[u], bq[u]);
#pragma unroll
      for (int e = 0; e < 8; ++e) q_s[r * dp + c + e] = qreg[u][e];
    }
  }
  __syncthreads();
  // ---- scores: one thread = one query row x 4 keys (keys strided by njb so lanes stay conflict-free) ----
  {
    const int njb = (Nk + 3) >> 2;
    for (int i = tid; i < nq * njb; i += 256) {
      const int r = i / njb, jb = i - r * njb;
      const float* qp = q_s + r * dp;
      const int j0 = jb, j1 = jb + njb, j2 = jb + 2 * njb, j3 = jb + 3 * njb;
      const float* k0 = kv_s + j0 * dp;
      const float* k1 = kv_s + (j1 < Nk ? j1 : j0) * dp;
      const float* k2 = kv_s + (j2 < Nk ? j2 : j0) * dp;
      const float* k3 = kv_s + (j3 < Nk ? j3 : j0) * dp;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
      for (int c = 0; c < d; ++c) {
        const float qv = qp[c];
        s0 = fmaf(qv, k0[c], s0);
        s1 = fmaf(qv, k1[c], s1);
        s2 = fmaf(qv, k2[c], s2);
        s3 = fmaf(qv, k3[c], s3);
      }
      const int lim = (q0 + r) + (Nk - Nq);      // causal: keep j <= i + (Nk - Nq)  (blocks.py:315-319)
      const float NEG = -3.402823466e+38f;
      float* pr = p_s + r * Nk;
      pr[j0] = (causal && j0 > lim) ? NEG : s0 * scale;
      if (j1 < Nk) pr[j1] = (causal && j1 > lim) ? NEG : s1 * scale;
      if (j2 < Nk) pr[j2] = (causal && j2 > lim) ? NEG : s2 * scale;
      if (j3 < Nk) pr[j3] = (causal && j3 > lim) ? NEG : s3 * scale;
    }
  }
  __syncthreads();
  // ---- V (already in registers) replaces K in LDS ---------------------------------------------
#pragma unroll
  for (int u = 0; u < MAXV; ++u) {
    const int i = tid + u * 256;
    if (i < nkv) {
      const int r = i / vpr, c = (i - r * vpr) * 8;
      float vv[8];
      vec_to_float(vraw[u], vv);
      if (u < FMAX) {
        if (fin_kv) finish(vv, kst[u], uv[u], bv[u]);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) kv_s[r * dp + c + e] = vv[e];
    }
  }
  // ---- softmax: one wavefront per row, 64-lane shuffle max / sum --------------------------------
  {
    const int wave = tid >> 6, lane = tid & 63;
    for (int r = wave; r < nq; r += 4) {
      float* pr = p_s + r * Nk;
      float e0 = -3.402823466e+38f, e1 = e0, e2 = e0;            // Nk <= 192: three per lane
      if (lane < Nk) e0 = pr[lane];
      if (lane + 64 < Nk) e1 = pr[lane + 64];
      if (lane + 128 < Nk) e2 = pr[lane + 128];
      float m = fmaxf(e0, fmaxf(e1, e2));
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
      e0 = (lane < Nk) ? expf(e0 - m) : 0.f;
      e1 = (lane + 64 < Nk) ? expf(e1 - m) : 0.f;
      e2 = (lane + 128 < Nk) ? expf(e2 - m) : 0.f;
      float sum = e0 + e1 + e2;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
      const float inv = 1.0f / sum;
      if (lane < Nk) pr[lane] = e0 * inv;
      if (lane + 64 < Nk) pr[lane + 64] = e1 * inv;
      if (lane + 128 < Nk) pr[lane + 128] = e2 * inv;
    }
  }
  __syncthreads();
  // ---- out = P V ---------------------------------------------------------------------------------
  for (int i = tid; i < nq * d; i += 256) {
    const int r = i / d, c = i - r * d;
    const float* pr = p_s + r * Nk;
    float o0 = 0.f, o1 = 0.f;
    int j = 0;
    for (; j + 1 < Nk; j += 2) {
      o0 = fmaf(pr[j], kv_s[j * dp + c], o0);
      o1 = fmaf(pr[j + 1], kv_s[(j + 1) * dp + c], o1);
    }
    if (j < Nk) o0 = fmaf(pr[j], kv_s[j * dp + c], o0);
    out[((size_t)b * Nq + q0 + r) * ldo + h * d + c] = (T)(o0 + o1);
  }
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
  JEN1_CHECK(dtype == JEN1_F32 || dtype == JEN1_BF16, "attention: bad dtype");
  const bool fin = finish_q || finish_kv;
  JEN1_CHECK(!fin || (ln_rowstats && ln_u && ln_b && ln_C >= 1), "attention: a deferred LayerNorm finish needs rowstats, u, bias and ln_C");
  JEN1_CHECK(!finish_kv || (!kv_row && !kv_extra && Nk * (d / 8) <= 256 * FMAX && Nq == Nk),
             "attention: the K/V finish is for self-attention over at most %d vectors", 256 * FMAX);
  const int dp = d + 1;
  const size_t lds = sizeof(float) * ((size_t)Nk * dp + (size_t)QCHUNK * dp + (size_t)QCHUNK * Nk);
  JEN1_CHECK(lds <= 160 * 1024, "attention: Nk=%d d=%d needs %zu B of LDS (> 160 KiB)", Nk, d, lds);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(B * H, (Nq + QCHUNK - 1) / QCHUNK);
  const float inv_c = fin ? 1.0f / (float)ln_C : 0.f;
  if (dtype == JEN1_F32) {
    auto kern = attention_kernel<float>;
    static bool set = false;
    if (!set) { JEN1_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, (const float*)q, (const float*)k, (const float*)v, (float*)out,
                       kv_row, (const float*)kv_extra, extra_row, extra_step, ld_extra, kx_off, vx_off, H, d, Nq, Nk, ldq, q_off, ldkv, k_off, v_off, ldo, causal, scale,
                       ln_rowstats, ln_u, ln_b, inv_c, ln_eps, finish_q, finish_kv);
  } else {
    auto kern = attention_kernel<bf16_t>;
    static bool set = false;
    if (!set) { JEN1_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                       (bf16_t*)out, kv_row, (const bf16_t*)kv_extra, extra_row, extra_step, ld_extra, kx_off, vx_off, H, d, Nq, Nk, ldq, q_off, ldkv, k_off, v_off, ldo, causal, scale,
                       ln_rowstats, ln_u, ln_b, inv_c, ln_eps, finish_q, finish_kv);
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
