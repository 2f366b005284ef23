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

template <typename T>
__global__ __launch_bounds__(256) void attention_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                         const T* __restrict__ v, T* __restrict__ out,
                                                         const int32_t* __restrict__ kv_row,
                                                         const T* __restrict__ kv_extra,
                                                         const int32_t* __restrict__ extra_row, int ld_extra,
                                                         int kx_off, int vx_off, int H, int d, int Nq,
                                                         int Nk, int ldq, int q_off, int ldkv, int k_off, int v_off,
                                                         int ldo, int causal, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
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
  const int xr = (kv_extra && extra_row) ? extra_row[b] : -1;
  // stage Q chunk and K (as float)
  for (int i = tid; i < nq * d; i += 256) {
    const int r = i / d, c = i - r * d;
    q_s[r * dp + c] = (float)q[((size_t)b * Nq + q0 + r) * ldq + q_off + h * d + c];
  }
  for (int i = tid; i < Nk * d; i += 256) {
    const int r = i / d, c = i - r * d;
    kv_s[r * dp + c] = (xr >= 0 && r == Nk - 1) ? (float)kv_extra[(size_t)xr * ld_extra + kx_off + h * d + c]
                                                : (float)k[(kvbase + r) * ldkv + k_off + h * d + c];
  }
  __syncthreads();
  // scores
  for (int i = tid; i < nq * Nk; i += 256) {
    const int r = i / Nk, j = i - r * Nk;
    const float* qp = q_s + r * dp;
    const float* kp = kv_s + j * dp;
    float s = 0.f;
    for (int c = 0; c < d; ++c) s = fmaf(qp[c], kp[c], s);
    s *= scale;
    if (causal && j > (q0 + r) + (Nk - Nq)) s = -3.402823466e+38f;
    p_s[r * Nk + j] = s;
  }
  __syncthreads();
  // V replaces K in LDS while the softmax runs on p_s
  for (int i = tid; i < Nk * d; i += 256) {
    const int r = i / d, c = i - r * d;
    kv_s[r * dp + c] = (xr >= 0 && r == Nk - 1) ? (float)kv_extra[(size_t)xr * ld_extra + vx_off + h * d + c]
                                                : (float)v[(kvbase + r) * ldkv + v_off + h * d + c];
  }
  // one wavefront per row: shuffle max, exp, shuffle sum
  {
    const int wave = tid >> 6, lane = tid & 63;
    for (int r = wave; r < nq; r += 4) {
      float* pr = p_s + r * Nk;
      float m = -3.402823466e+38f;
      for (int j = lane; j < Nk; j += 64) m = fmaxf(m, pr[j]);
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
      float sum = 0.f;
      for (int j = lane; j < Nk; j += 64) {
        const float e = expf(pr[j] - m);
        pr[j] = e;
        sum += e;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
      const float inv = 1.0f / sum;
      for (int j = lane; j < Nk; j += 64) pr[j] *= inv;
    }
  }
  __syncthreads();
  // out = P V
  for (int i = tid; i < nq * d; i += 256) {
    const int r = i / d, c = i - r * d;
    const float* pr = p_s + r * Nk;
    float o = 0.f;
    for (int j = 0; j < Nk; ++j) o = fmaf(pr[j], kv_s[j * dp + c], o);
    out[((size_t)b * Nq + q0 + r) * ldo + h * d + c] = (T)o;
  }
}

}  // namespace

extern "C" int jen1_attention(const void* q, const void* k, const void* v, void* out, const int32_t* kv_row,
                              const void* kv_extra, const int32_t* extra_row, int ld_extra, int kx_off, int vx_off, int B,
                              int H, int d, int Nq, int Nk, int ldq, int q_off, int ldkv, int k_off, int v_off,
                              int ldo, int causal, float scale, int dtype, void* stream) {
  JEN1_CHECK(q && k && v && out, "attention: null pointer");
  JEN1_CHECK(B >= 1 && H >= 1 && d >= 1 && Nq >= 1 && Nk >= 1, "attention: bad sizes");
  JEN1_CHECK(dtype == JEN1_F32 || dtype == JEN1_BF16, "attention: bad dtype");
  const int dp = d + 1;
  const size_t lds = sizeof(float) * ((size_t)Nk * dp + (size_t)QCHUNK * dp + (size_t)QCHUNK * Nk);
  JEN1_CHECK(lds <= 160 * 1024, "attention: Nk=%d d=%d needs %zu B of LDS (> 160 KiB)", Nk, d, lds);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(B * H, (Nq + QCHUNK - 1) / QCHUNK);
  if (dtype == JEN1_F32) {
    auto kern = attention_kernel<float>;
    static bool set = false;
    if (!set) { JEN1_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, (const float*)q, (const float*)k, (const float*)v, (float*)out,
                       kv_row, (const float*)kv_extra, extra_row, ld_extra, kx_off, vx_off, H, d, Nq, Nk, ldq, q_off, ldkv, k_off, v_off, ldo, causal, scale);
  } else {
    auto kern = attention_kernel<bf16_t>;
    static bool set = false;
    if (!set) { JEN1_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; }
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v,
                       (bf16_t*)out, kv_row, (const bf16_t*)kv_extra, extra_row, ld_extra, kx_off, vx_off, H, d, Nq, Nk, ldq, q_off, ldkv, k_off, v_off, ldo, causal, scale);
  }
  JEN1_HIP(hipGetLastError());
  return 0;
}
