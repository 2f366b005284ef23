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

// ---- LSTM layer spread over NW workgroups (the fast path) -----------------------------------------------------------
// Workgroup w owns UPW = 32 hidden units = 128 gate rows; its slice of W_hh (128 x H) lives in registers for the whole
// sequence: thread (row r, k slice s) holds W_hh[row r][s * KS .. s * KS + KS).  Up to LB = 8 sequences advance together.
// Per step: h_t of all sequences is in LDS; every thread forms its partial dot products for the LB sequences (LDS
// broadcast reads), the k slices are summed through LDS, 32 x LB threads apply the gates, publish h_{t+1} of their
// units with write-through stores, and one grid barrier (relaxed agent-scope atomics, tools/microbench/gridbar.hip:
// ~1 us at 16 workgroups) separates the steps.  The spin is bounded (2^26 polls, seconds): a workgroup that never
// becomes resident turns into an error flag that the host side checks (jen1_amd/encodec.py), never into a hang.
constexpr int LSTM_UPW = 32, LSTM_LB = 8, LSTM_NS = 8;

template <typename T, int KS>    // KS = H / LSTM_NS k values per thread
__global__ __launch_bounds__(1024) void lstm_multi_kernel(const float* __restrict__ gin, const void* whh_, const void* skip_, void* y_,
                                                          float* hbuf, unsigned* ctr, int B, int Tn, int H, int ld_y) {
  __shared__ float h_s[LSTM_LB][1024];                 // h_t of the sequences of this group (H <= 1024)
  __shared__ float red[LSTM_NS][128][LSTM_LB + 1];
  __shared__ float gates[LSTM_LB][128];
  const T* whh = reinterpret_cast<const T*>(whh_);
  const T* skip = reinterpret_cast<const T*>(skip_);
  T* y = reinterpret_cast<T*>(y_);
  const int tid = threadIdx.x, NW = gridDim.x, w = blockIdx.x, grp = blockIdx.y;
  const int b0 = grp * LSTM_LB, nb = min(LSTM_LB, B - b0);
  const int rl = tid & 127, sl = tid >> 7;            // gate row within the workgroup, k slice
  const int gate = rl >> 5, u = rl & 31;
  const int grow = gate * H + w * LSTM_UPW + u;        // row of W_hh / entry of the 4H gate vector
  float wreg[KS];
#pragma unroll
  for (int j = 0; j < KS; ++j) wreg[j] = (float)whh[(long long)grow * H + sl * KS + j];
  float* hb = hbuf + (long long)grp * 2 * LSTM_LB * H;   // [2][LB][H] ping-pong of the published hidden state
  unsigned* my_ctr = ctr + grp * 32;
  unsigned epoch = 0;
  float c = 0.f;                                         // cell state of (unit u2, sequence b2) for threads < 32 * LB
  const int u2 = tid & 31, b2 = tid >> 5;
  for (int i = tid; i < LSTM_LB * 1024; i += 1024) (&h_s[0][0])[i] = 0.f;
  __syncthreads();
  for (int t = 0; t < Tn; ++t) {
    // partial dot products of this thread's k slice for every sequence
    // the input projection of this step is fetched first: its latency hides behind the dot products
    const float g_in = (sl < nb) ? gin[((long long)(b0 + sl) * Tn + t) * (4 * H) + grow] : 0.f;
    float part[LSTM_LB];
#pragma unroll
    for (int b = 0; b < LSTM_LB; ++b) part[b] = 0.f;
#pragma unroll
    for (int b = 0; b < LSTM_LB; ++b) {
      if (b < nb) {                                  // uniform: absent sequences cost nothing
#pragma unroll
        for (int j = 0; j < KS; j += 4) {
          const float4 hv = *reinterpret_cast<const float4*>(&h_s[b][sl * KS + j]);
          part[b] += wreg[j] * hv.x + wreg[j + 1] * hv.y + wreg[j + 2] * hv.z + wreg[j + 3] * hv.w;
        }
      }
    }
#pragma unroll
    for (int b = 0; b < LSTM_LB; ++b) red[sl][rl][b] = part[b];
    __syncthreads();
    {   // thread (rl, b = sl) sums the k slices and adds the input projection
      const int b = sl;
      float g = 0.f;
#pragma unroll
      for (int s2 = 0; s2 < LSTM_NS; ++s2) g += red[s2][rl][b];
      gates[b][rl] = g + g_in;
    }
    __syncthreads();
    float* hnext = hb + (long long)((t + 1) & 1) * LSTM_LB * H;
    if (tid < 32 * LSTM_LB && b2 < nb) {
      const float ig = sigmoid_f(gates[b2][u2]), fg = sigmoid_f(gates[b2][32 + u2]), gg = tanhf(gates[b2][64 + u2]), og = sigmoid_f(gates[b2][96 + u2]);
      c = fg * c + ig * gg;
      const float hn = og * tanhf(c);
      const int j = w * LSTM_UPW + u2;
      __hip_atomic_store(hnext + b2 * H + j, hn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const long long o = ((long long)(b0 + b2) * Tn + t) * ld_y + j;
      y[o] = (T)(hn + (skip != nullptr ? (float)skip[o] : 0.f));
    }
    // grid barrier over the NW workgroups of this group
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      ++epoch;
      __hip_atomic_fetch_add(my_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = epoch * NW;
      int spins = 0;
      while (__hip_atomic_load(my_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 26)) { __hip_atomic_store(my_ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
    __syncthreads();
    if (t + 1 < Tn) {
      for (int i = tid; i < nb * H; i += 1024) {
        const int b = i / H, j = i - b * H;
        h_s[b][j] = __hip_atomic_load(hnext + b * H + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
    }
  }
}

// ---- the same layer with the recurrent product on the matrix cores (bf16 weights) -----------------------------------
// Workgroup w owns the same 128 gate rows; wave v holds the MFMA A fragments of row tile v % 8 for the K half v / 8
// (8 fragments = 32 VGPRs, resident for the whole sequence).  Up to 16 sequences are the 16 MFMA columns.  h_t stays
// float32 in the exchange buffer and is split into a bf16 high and low part when it is staged into LDS (two MFMAs per
// fragment), so the recurrence sees h at ~16 mantissa bits while the weights are the bf16 ones of the bf16 mode.
constexpr int LSTM_MB = 16;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void lstm_multi_mfma_kernel(const float* __restrict__ gin, const bf16_t* __restrict__ whh,
                                                               const bf16_t* __restrict__ skip, bf16_t* __restrict__ y, float* hbuf,
                                                               unsigned* ctr, int B, int Tn, int H, int ld_y) {
  // H == 512 (checked by the launcher): K halves of 256 = 8 MFMA steps of 32
  __shared__ __attribute__((aligned(16))) bf16_t h_hi[LSTM_MB][512 + 8];
  __shared__ __attribute__((aligned(16))) bf16_t h_lo[LSTM_MB][512 + 8];
  __shared__ float part[8][64][4];
  __shared__ float gates[LSTM_MB][128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NW = gridDim.x, w = blockIdx.x, grp = blockIdx.y;
  const int b0 = grp * LSTM_MB, nb = min(LSTM_MB, B - b0);
  const int rt = wave & 7, kh = wave >> 3;
  // gate rows of tile rt: rl = rt * 16 + i  ->  gate rl / 32, unit rl % 32
  const int rl_a = rt * 16 + (lane & 15);
  const int grow_a = (rl_a >> 5) * H + w * LSTM_UPW + (rl_a & 31);
  bf16x8 wf[8];
#pragma unroll
  for (int s2 = 0; s2 < 8; ++s2)
    wf[s2] = *reinterpret_cast<const bf16x8*>(whh + (long long)grow_a * H + kh * 256 + s2 * 32 + (lane >> 4) * 8);
  float* hb = hbuf + (long long)grp * 2 * LSTM_MB * H;
  unsigned* my_ctr = ctr + grp * 32;
  unsigned epoch = 0;
  float c = 0.f;
  const int u2 = tid & 31, b2 = tid >> 5;           // unit / sequence of the update threads (tid < 32 * 16 = 512)
  for (int i = tid; i < LSTM_MB * (512 + 8); i += 1024) { (&h_hi[0][0])[i] = (bf16_t)0.f; (&h_lo[0][0])[i] = (bf16_t)0.f; }
  __syncthreads();
  for (int t = 0; t < Tn; ++t) {
    // input projection of the 4 gate rows x 1 sequence this lane will own after the reduction (waves 0..7)
    const int col = lane & 15;
    const int rl0 = rt * 16 + (lane >> 4) * 4;
    const int grow0 = (rl0 >> 5) * H + w * LSTM_UPW + (rl0 & 31);
    float4 g_in = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kh == 0 && col < nb) g_in = *reinterpret_cast<const float4*>(gin + ((long long)(b0 + col) * Tn + t) * (4 * H) + grow0);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) {
      const int k = kh * 256 + s2 * 32 + (lane >> 4) * 8;
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(&h_hi[col][k]);
      const bf16x8 bl = *reinterpret_cast<const bf16x8*>(&h_lo[col][k]);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s2], bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s2], bl, acc, 0, 0, 0);
    }
    if (kh == 1) *reinterpret_cast<float4*>(&part[rt][lane][0]) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    __syncthreads();
    if (kh == 0) {
      const float4 o = *reinterpret_cast<const float4*>(&part[rt][lane][0]);
      if (col < nb) {
        gates[col][rl0 + 0] = acc[0] + o.x + g_in.x;
        gates[col][rl0 + 1] = acc[1] + o.y + g_in.y;
        gates[col][rl0 + 2] = acc[2] + o.z + g_in.z;
        gates[col][rl0 + 3] = acc[3] + o.w + g_in.w;
      }
    }
    __syncthreads();
    float* hnext = hb + (long long)((t + 1) & 1) * LSTM_MB * H;
    if (tid < 32 * LSTM_MB && b2 < nb) {
      const float ig = sigmoid_f(gates[b2][u2]), fg = sigmoid_f(gates[b2][32 + u2]), gg = tanhf(gates[b2][64 + u2]), og = sigmoid_f(gates[b2][96 + u2]);
      c = fg * c + ig * gg;
      const float hn = og * tanhf(c);
      const int j = w * LSTM_UPW + u2;
      __hip_atomic_store(hnext + b2 * H + j, hn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const long long o = ((long long)(b0 + b2) * Tn + t) * ld_y + j;
      y[o] = (bf16_t)(hn + (skip != nullptr ? (float)skip[o] : 0.f));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      ++epoch;
      __hip_atomic_fetch_add(my_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = epoch * NW;
      int spins = 0;
      while (__hip_atomic_load(my_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 26)) { __hip_atomic_store(my_ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
    __syncthreads();
    if (t + 1 < Tn) {
      // restage h_{t+1}: 16-byte sc1 loads (nb * 512 floats = nb * 128 vectors), split into bf16 high + low
      const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(hnext, 0, LSTM_MB * 512 * 4, 0x00020000);
      for (int i = tid; i < nb * 128; i += 1024) {
        const int b = i >> 7, j = (i & 127) * 4;
        const u32x4_t q = __builtin_amdgcn_raw_buffer_load_b128(rh, (unsigned)(b * 512 + j) * 4u, 0, 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float hv = __uint_as_float(q[e]);
          const bf16_t hi = (bf16_t)hv;
          h_hi[b][j + e] = hi;
          h_lo[b][j + e] = (bf16_t)(hv - (float)hi);
        }
      }
      __syncthreads();
    }
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

extern "C" int jen1_lstm_layer_multi(const float* gin, const void* whh, const void* skip, void* y, float* hbuf, uint32_t* counters,
                                     int B, int T, int H, int ld_y, int dtype, void* stream) {
  JEN1_CHECK(dtype == JEN1_F32 || dtype == JEN1_BF16, "jen1_lstm_layer_multi: dtype must be JEN1_F32 or JEN1_BF16");
  JEN1_CHECK(gin && whh && y && hbuf && counters, "jen1_lstm_layer_multi: NULL argument");
  JEN1_CHECK(B >= 1 && T >= 1 && ld_y >= H, "jen1_lstm_layer_multi: bad shape B=%d T=%d H=%d ld_y=%d", B, T, H, ld_y);
  JEN1_CHECK(H == 256 || H == 512 || H == 1024, "jen1_lstm_layer_multi: H must be 256, 512 or 1024");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == JEN1_BF16 && H == 512 && B > 1) {
    // matrix-core variant: 16 sequences per group (hbuf / counters are sized for 8 per group by the caller: half as many groups)
    const int groups16 = (B + LSTM_MB - 1) / LSTM_MB;
    JEN1_CHECK(groups16 * (H / LSTM_UPW) <= 256, "jen1_lstm_layer_multi: too many workgroups to be co-resident");
    hipLaunchKernelGGL(lstm_multi_mfma_kernel, dim3(H / LSTM_UPW, groups16), dim3(1024), 0, s, gin, reinterpret_cast<const bf16_t*>(whh),
                       reinterpret_cast<const bf16_t*>(skip), reinterpret_cast<bf16_t*>(y), hbuf, counters, B, T, H, ld_y);
    JEN1_HIP(hipGetLastError());
    return 0;
  }
  const int groups = (B + LSTM_LB - 1) / LSTM_LB, nw = H / LSTM_UPW;
  JEN1_CHECK(groups * nw <= 256, "jen1_lstm_layer_multi: %d workgroups must be co-resident (at most 256)", groups * nw);
  const dim3 grid(nw, groups);
#define JEN1_LSTMM(TT, KS) hipLaunchKernelGGL((lstm_multi_kernel<TT, KS>), grid, dim3(1024), 0, s, gin, whh, skip, y, hbuf, counters, B, T, H, ld_y)
  if (dtype == JEN1_F32) { if (H == 256) JEN1_LSTMM(float, 32); else if (H == 512) JEN1_LSTMM(float, 64); else JEN1_LSTMM(float, 128); }
  else { if (H == 256) JEN1_LSTMM(bf16_t, 32); else if (H == 512) JEN1_LSTMM(bf16_t, 64); else JEN1_LSTMM(bf16_t, 128); }
#undef JEN1_LSTMM
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
