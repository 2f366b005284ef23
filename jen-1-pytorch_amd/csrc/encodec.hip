// Encodec pieces either side of the sampler (include/jen1_train.h): the residual-vector-quantizer decode that turns
// codes into the latents the denoiser works on (generation.py:145-150) and the LSTM of the SEANet decoder that turns
// sampled latents back into audio (generation.py:130).  The convolutions / transposed convolutions / GroupNorm / ELU
// of the decoder run on jen1_train_gemm and the train_ops kernels (jen1_amd/encodec.py).
#include "common.h"
#include "jen1_train.h"

namespace {

// out[b][d][t] = sum_q tables[q][codes[q][b][t]][d] : one block per (b, 64 frames); rows of the codebooks are
// read 128 floats at a time (coalesced), the transposed store goes through LDS.
__global__ __launch_bounds__(256) void rvq_decode_kernel(const long long* __restrict__ codes, const float* __restrict__ tables,
                                                         float* __restrict__ out, int n_q, int B, int T, int bins, int D) {
  __shared__ float tile[64][129];
  const int b = blockIdx.y, t0 = blockIdx.x * 64;
  for (int d0 = 0; d0 < D; d0 += 128) {
    for (int e = threadIdx.x; e < 64 * 128; e += 256) {
      const int tl = e / 128, d = d0 + e % 128, t = t0 + tl;
      float acc = 0.f;
      if (t < T && d < D) {
        for (int q = 0; q < n_q; ++q) {
          long long idx = codes[((long long)q * B + b) * T + t];
          idx = idx < 0 ? 0 : (idx >= bins ? bins - 1 : idx);
          acc += tables[((long long)q * bins + idx) * D + d];
        }
      }
      tile[tl][e % 128] = acc;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 128; e += 256) {
      const int dl = e / 64, tl = e % 64, d = d0 + dl, t = t0 + tl;
      if (t < T && d < D) out[((long long)b * D + d) * T + t] = tile[tl][dl];
    }
    __syncthreads();
  }
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// One workgroup per sequence; 1024 threads own the 4H gate rows (RPT = 4H / 1024 each).  Per step every thread adds
// W_hh^T[k][row] * h[k] over k: the [H][4H] layout makes the weight reads coalesced, and the k loop is unrolled by 8
// with all loads issued before the FMAs, so the step is bound by streaming the matrix from L2 (it is re-read every
// step: 2 MB in bf16), not by one load latency per k.  Then H threads update (c, h).  Sequential in T by nature; the
// sequences of a batch run in parallel on different CUs.
template <typename T, int RPT>
__global__ __launch_bounds__(1024) void lstm_layer_kernel(const float* __restrict__ gin, const void* whh_t_, const void* skip_, void* y_,
                                                          int Tn, int H, int ld_y) {
  extern __shared__ float smem[];
  float* h = smem;              // [H]
  float* gates = smem + H;      // [4H]
  const T* whh_t = reinterpret_cast<const T*>(whh_t_);
  const T* skip = reinterpret_cast<const T*>(skip_);
  T* y = reinterpret_cast<T*>(y_);
  const int b = blockIdx.x, tid = threadIdx.x, G = 4 * H;      // G == RPT * 1024 (checked by the launcher)
  float c = 0.f;                 // cell state of hidden unit `tid` (threads < H)
  for (int j = tid; j < H; j += 1024) h[j] = 0.f;
  __syncthreads();
  constexpr int KU = 8;
  for (int t = 0; t < Tn; ++t) {
    const float* g_in = gin + ((long long)b * Tn + t) * G;
    float acc[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) acc[r] = g_in[tid + r * 1024];
    for (int k0 = 0; k0 < H; k0 += KU) {          // H % 8 == 0 (checked by the launcher)
      T w[KU][RPT];
#pragma unroll
      for (int u = 0; u < KU; ++u)
#pragma unroll
        for (int r = 0; r < RPT; ++r) w[u][r] = whh_t[(long long)(k0 + u) * G + tid + r * 1024];
#pragma unroll
      for (int u = 0; u < KU; ++u) {
        const float hk = h[k0 + u];
#pragma unroll
        for (int r = 0; r < RPT; ++r) acc[r] += (float)w[u][r] * hk;
      }
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) gates[tid + r * 1024] = acc[r];
    __syncthreads();
    for (int j = tid; j < H; j += 1024) {     // H <= 1024: each unit belongs to one thread for the whole sequence
      const float ig = sigmoid_f(gates[j]), fg = sigmoid_f(gates[H + j]), gg = tanhf(gates[2 * H + j]), og = sigmoid_f(gates[3 * H + j]);
      c = fg * c + ig * gg;
      const float hn = og * tanhf(c);
      h[j] = hn;
      const long long o = ((long long)b * Tn + t) * ld_y + j;
      y[o] = (T)(hn + (skip != nullptr ? (float)skip[o] : 0.f));
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int jen1_rvq_decode(const int64_t* codes, const float* tables, float* out, int n_q, int B, int T, int bins, int D, void* stream) {
  JEN1_CHECK(codes && tables && out, "jen1_rvq_decode: NULL argument");
  JEN1_CHECK(n_q >= 1 && B >= 1 && T >= 1 && bins >= 1 && D >= 1 && B <= 65535, "jen1_rvq_decode: bad shape");
  hipLaunchKernelGGL(rvq_decode_kernel, dim3((T + 63) / 64, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const long long*>(codes), tables, out, n_q, B, T, bins, D);
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_lstm_layer(const float* gin, const void* whh_t, const void* skip, void* y, int B, int T, int H, int ld_y, int dtype,
                               void* stream) {
  JEN1_CHECK(dtype == JEN1_F32 || dtype == JEN1_BF16, "jen1_lstm_layer: dtype must be JEN1_F32 or JEN1_BF16");
  JEN1_CHECK(gin && whh_t && y, "jen1_lstm_layer: NULL argument");
  JEN1_CHECK(B >= 1 && T >= 1 && ld_y >= H, "jen1_lstm_layer: bad shape B=%d T=%d H=%d ld_y=%d", B, T, H, ld_y);
  JEN1_CHECK(H == 256 || H == 512 || H == 1024, "jen1_lstm_layer: H must be 256, 512 or 1024 (4H a multiple of the 1024 threads)");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t lds = sizeof(float) * 5 * H;
#define JEN1_LSTM(TT, RPT) hipLaunchKernelGGL((lstm_layer_kernel<TT, RPT>), dim3(B), dim3(1024), lds, s, gin, whh_t, skip, y, T, H, ld_y)
  const int rpt = 4 * H / 1024;
  if (dtype == JEN1_F32) { if (rpt == 1) JEN1_LSTM(float, 1); else if (rpt == 2) JEN1_LSTM(float, 2); else JEN1_LSTM(float, 4); }
  else { if (rpt == 1) JEN1_LSTM(bf16_t, 1); else if (rpt == 2) JEN1_LSTM(bf16_t, 2); else JEN1_LSTM(bf16_t, 4); }
#undef JEN1_LSTM
  JEN1_HIP(hipGetLastError());
  return 0;
}
