// Boundary / elementwise kernels of the JEN-1 denoiser step (gfx950).
//   pack_input      [B][C][T] f32 (+ context channels) -> channel-last tile + GroupNorm sums
//   unpack_output   channel-last -> [B][C][T] f32
//   row_stats       LayerNorm row sums of an external tensor (text context)
//   time_features   LearnedPositionalEmbedding -> Linear -> GELU, always float32
//   linear_f32      tiny float32 Linear (+GELU) for the mapping MLP
//   cfg_ddim_step   CFG combine + std rescale + x0/eps prediction + DDIM update, one pass
#include <stdarg.h>

#include "common.h"

thread_local char g_jen1_err[512] = {0};

int jen1_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_jen1_err, sizeof(g_jen1_err), fmt, ap);
  va_end(ap);
  return 1;
}

extern "C" const char* jen1_last_error(void) { return g_jen1_err; }
extern "C" const char* jen1_build_info(void) { return "libjen1_hip gfx950 (CDNA4) hipcc; abi 2"; }
extern "C" int jen1_abi_version(void) { return 2; }    // 2: jen1_gemm_args.map_shift_b, jen1_repack_entry.dst2 (round 3)

// A kernel, not hipMemsetAsync: memset nodes interleaved with kernel nodes in a captured graph were observed to
// run out of order on ROCm 7.2 (training step), a kernel node never is.
__global__ __launch_bounds__(256) void memset_zero_kernel(unsigned* __restrict__ p, long long nwords) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nwords; i += (long long)gridDim.x * 256) p[i] = 0u;
}

extern "C" int jen1_memset_zero(void* p, int64_t bytes, void* stream) {
  JEN1_CHECK(p && bytes >= 0, "memset_zero: bad arguments");
  JEN1_CHECK(((uintptr_t)p & 3) == 0 && (bytes & 3) == 0, "memset_zero: pointer and size must be multiples of 4");
  if (bytes == 0) return 0;
  const long long nwords = bytes / 4;
  long long blocks = (nwords + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(memset_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<unsigned*>(p), nwords);
  JEN1_HIP(hipGetLastError());
  return 0;
}

namespace {

// ------------------------------------------------------------------------------------------------
// [B][C][T] + [B][Cc][T]  ->  [nrep*B][T][ld]   (reference torch.cat, jen1/model/model.py:240, :332-349)
template <typename T>
__global__ __launch_bounds__(256) void pack_input_kernel(const float* __restrict__ x, const float* __restrict__ ctx,
                                                          T* __restrict__ y, float* __restrict__ stats, int B, int C,
                                                          int Cc, int Tn, int ld, int nrep, float* __restrict__ parts) {
  __shared__ float tile[32][33];
  __shared__ float st[64];
  const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32, b = blockIdx.z;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  if (threadIdx.x < 64) st[threadIdx.x] = 0.f;
  const int cpf = ld / JEN1_FINE_GROUPS;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = ty + 8 * i;
    const int c = c0 + r, t = t0 + tx;
    float v = 0.f;
    if (t < Tn) {
      if (c < C) v = x[((size_t)b * C + c) * Tn + t];
      else if (c < C + Cc) v = ctx[((size_t)b * Cc + (c - C)) * Tn + t];
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int t = t0 + r, c = c0 + tx;
    if (t < Tn) {
      const T v = (T)tile[tx][r];
      for (int rep = 0; rep < nrep; ++rep) y[((size_t)(rep * B + b) * Tn + t) * ld + c] = v;
    }
  }
  if (parts) {
    // fixed-order statistics: the (sum, sumsq) of this block's 32 time steps per channel, written (not accumulated); summed over the
    // time blocks and the channels of a fine group by gn_stats_from_parts_kernel
    if (threadIdx.x < 32) {
      const int r = threadIdx.x;
      float sv = 0.f, sq = 0.f;
#pragma unroll 8
      for (int j = 0; j < 32; ++j) {
        const float v = tile[r][j];
        sv += v;
        sq = __builtin_fmaf(v, v, sq);
      }
      *reinterpret_cast<float2*>(parts + (((size_t)b * gridDim.x + blockIdx.x) * ld + c0 + r) * 2) = make_float2(sv, sq);
    }
    return;
  }
  if (stats) {
    // one lane per channel of the slab walks its 32 time steps in LDS (lane r reads bank (r + j) mod 32: no conflicts), then one LDS
    // atomic pair per channel into the fine groups of this 32-channel slab -- no cross-lane traffic on the way
    if (threadIdx.x < 32) {
      const int r = threadIdx.x;
      float sv = 0.f, sq = 0.f;
#pragma unroll 8
      for (int j = 0; j < 32; ++j) {
        const float v = tile[r][j];
        sv += v;
        sq = __builtin_fmaf(v, v, sq);
      }
      if (sq != 0.f) {
        const int fg = (c0 + r) / cpf;
        atomicAdd(&st[2 * fg - 2 * (c0 / cpf)], sv);
        atomicAdd(&st[2 * fg - 2 * (c0 / cpf) + 1], sq);
      }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      const int fgl = threadIdx.x >> 1;
      const int fg = c0 / cpf + fgl;
      const float v = st[threadIdx.x];
      if (fg < JEN1_FINE_GROUPS && v != 0.f)
        for (int rep = 0; rep < nrep; ++rep) unsafeAtomicAdd(stats + (size_t)(rep * B + b) * 64 + fg * 2 + (threadIdx.x & 1), v);
    }
  }
}

// parts[B][ntb][ld][2] -> stats[nrep * B][32][2]: thread c adds the time blocks of channel c in order, then one thread per fine group adds
// its channels in order
__global__ __launch_bounds__(512) void gn_stats_from_parts_kernel(const float* __restrict__ parts, float* __restrict__ stats, int B, int ntb,
                                                                   int ld, int nrep) {
  extern __shared__ float chan[];        // [ld][2]
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < ld; c += blockDim.x) {
    float s = 0.f, q = 0.f;
    const float2* p = reinterpret_cast<const float2*>(parts) + (size_t)b * ntb * ld + c;
    // (sixteen independent loads in flight, added in index order)
    for (int i0 = 0; i0 < ntb; i0 += 16) {
      float2 v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = p[(size_t)(i0 + j < ntb ? i0 + j : 0) * ld];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        s += i0 + j < ntb ? v[j].x : 0.f;
        q += i0 + j < ntb ? v[j].y : 0.f;
      }
    }
    chan[2 * c] = s;
    chan[2 * c + 1] = q;
  }
  __syncthreads();
  const int cpf = ld / JEN1_FINE_GROUPS;
  if (threadIdx.x < JEN1_FINE_GROUPS) {
    float s = 0.f, q = 0.f;
    for (int k = 0; k < cpf; ++k) {
      s += chan[2 * (threadIdx.x * cpf + k)];
      q += chan[2 * (threadIdx.x * cpf + k) + 1];
    }
    for (int rep = 0; rep < nrep; ++rep) {
      stats[(size_t)(rep * B + b) * 64 + threadIdx.x * 2] = s;
      stats[(size_t)(rep * B + b) * 64 + threadIdx.x * 2 + 1] = q;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void unpack_output_kernel(const T* __restrict__ y, float* __restrict__ out, int C,
                                                             int Tn, int ld) {
  __shared__ float tile[32][33];
  const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32, b = blockIdx.z;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int t = t0 + r, c = c0 + tx;
    tile[r][tx] = (t < Tn && c < C) ? (float)y[((size_t)b * Tn + t) * ld + c] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, t = t0 + tx;
    if (c < C && t < Tn) out[((size_t)b * C + c) * Tn + t] = tile[tx][r];
  }
}

// one wavefront per row: (sum, sumsq) over C columns  (LayerNorm statistics, blocks.py:401,427)
template <typename T>
__global__ __launch_bounds__(256) void row_stats_kernel(const T* __restrict__ x, float* __restrict__ stats, int rows,
                                                         int C, int ld) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float s = 0.f, q = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float v = (float)x[(size_t)row * ld + c];
    s += v;
    q += v * v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_xor(s, off);
    q += __shfl_xor(q, off);
  }
  if (lane == 0) {
    stats[2 * row] = s;
    stats[2 * row + 1] = q;
  }
}

// fine-group GroupNorm statistics in a fixed order: one workgroup per (fine group, batch element); a thread sums elements
// tid, tid + 256, ... of the group's L x cpf block, then a fixed shuffle / LDS tree
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, float* __restrict__ stats, int L, int ld) {
  __shared__ float red[8];
  const int fg = blockIdx.x, b = blockIdx.y;
  const int cpf = ld / JEN1_FINE_GROUPS;
  const T* p = x + (size_t)b * L * ld + fg * cpf;
  const int n = L * cpf;
  float s = 0.f, q = 0.f;
  for (int e = threadIdx.x; e < n; e += 256) {
    const int r = e / cpf, c = e - r * cpf;      // (integer division: a rounded float reciprocal picks the wrong row beyond ~2.8 M elements)
    const float v = (float)p[(size_t)r * ld + c];
    s += v;
    q += v * v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_xor(s, off);
    q += __shfl_xor(q, off);
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[2 * wave] = s;
    red[2 * wave + 1] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    stats[(size_t)b * 64 + fg * 2] = (red[0] + red[2]) + (red[4] + red[6]);
    stats[(size_t)b * 64 + fg * 2 + 1] = (red[1] + red[3]) + (red[5] + red[7]);
  }
}

// LearnedPositionalEmbedding + Linear + GELU (utils/module.py:58-79; model.py:84-89, :286-291).
// The phase is ((t * w) * 2) * pi evaluated left to right in float32 exactly like the reference:
// at t = 999 the argument is ~2e4 rad, one float32 ulp there is 2e-3 rad.
template <typename TT>
__global__ __launch_bounds__(256) void time_features_kernel(const TT* __restrict__ t, const float* __restrict__ freq,
                                                             const float* __restrict__ w, const float* __restrict__ bias,
                                                             float* __restrict__ out, int half, int out_features) {
  extern __shared__ float feat[];   // [2*half + 1]
  const int n = blockIdx.x;
  const float tv = (float)t[n];
  const int nin = 2 * half + 1;
  for (int i = threadIdx.x; i < half; i += 256) {
    const float f = __fmul_rn(__fmul_rn(__fmul_rn(tv, freq[i]), 2.0f), 3.14159274101257324f);
    feat[1 + i] = sinf(f);
    feat[1 + half + i] = cosf(f);
  }
  if (threadIdx.x == 0) feat[0] = tv;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int o = blockIdx.y * 4 + wave; o < out_features; o += gridDim.y * 4) {
    float s = 0.f;
    for (int i = lane; i < nin; i += 64) s = fmaf(feat[i], w[(size_t)o * nin + i], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) out[(size_t)n * out_features + o] = gelu_erf(s + bias[o]);
  }
}

// y[n][o] = act(x[n] . w[o] + bias[o]); one wavefront per output feature; rows are processed 8 at a time
// so that the loads of a pass are all in flight together (one wave per SIMD: latency, not bandwidth)
__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, int n,
                                                          int in_f, int out_f, int act) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (o >= out_f) return;
  const float* wr = w + (size_t)o * in_f;
  const float bo = bias ? bias[o] : 0.f;
  for (int r0 = 0; r0 < n; r0 += 8) {
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = lane; i < in_f; i += 64) {
      const float wv = wr[i];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = r0 + k < n ? r0 + k : n - 1;
        s[k] = fmaf(x[(size_t)r * in_f + i], wv, s[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s[k] += __shfl_xor(s[k], off);
    }
    if (lane < 8 && r0 + lane < n) {
      float v = s[0];
#pragma unroll
      for (int k = 1; k < 8; ++k) v = (lane == k) ? s[k] : v;
      v += bo;
      y[(size_t)(r0 + lane) * out_f + o] = act == JEN1_ACT_GELU ? gelu_erf(v) : v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// CFG combine + rescale (model.py:362-369) [+ model_predictions + DDIM update (gdm.py:128-140, 212-222)]
// One workgroup = 32 time steps of one batch element, all C channels.
template <typename T, bool DDIM>
__global__ __launch_bounds__(256) void cfg_step_kernel(const T* __restrict__ net, const float* __restrict__ x,
                                                        const float* __restrict__ noise, const float* __restrict__ coef,
                                                        float* __restrict__ x_out, float* __restrict__ eps_out,
                                                        float* __restrict__ x0_out, const int32_t* step_idx,
                                                        int B, int C, int Tn, int ld, int nrep,
                                                        float scale, int scale_cfg, float phi, int objective, int clip_x0,
                                                        int32_t* adv_step, unsigned* __restrict__ adv_ticket) {
  extern __shared__ float tile[];   // [C][33]
  // step_idx and adv_step may be the SAME word (jen1_cfg_ddim_step_adv: the last block advances the counter every block reads),
  // so neither is __restrict__, the counter is read with an atomic load that cannot be sunk behind the ticket below, and the
  // row of coefficients is in registers before the barrier
  float cf[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (DDIM) {
    if (step_idx) {                  // per-step rows of the coefficient / noise tables
      const int st = __hip_atomic_load(step_idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      coef += (size_t)st * 8;
      if (noise) noise += (size_t)st * B * C * Tn;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) cf[i] = coef[i];
  }
  const int t0 = blockIdx.x * 32, b = blockIdx.y;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int RPW = 8;            // rows (time steps) per wave
  constexpr int CPL = 4;            // channels per lane: C <= 256
  // phase 1: per (b, t) row -> guided output, written transposed into LDS.  All rows of the wave are
  // loaded before any is reduced (memory-level parallelism; one wave per SIMD here).
  float oc[RPW][CPL], ou[RPW][CPL];
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int r = wave + 4 * i;
    const int t = (t0 + r < Tn) ? t0 + r : Tn - 1;
    const T* pc = net + ((size_t)b * Tn + t) * ld;
    const T* pu = net + ((size_t)((nrep == 2 ? B : 0) + b) * Tn + t) * ld;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int c = lane + 64 * k;
      const int cc_ = c < C ? c : 0;
      oc[i][k] = (float)pc[cc_];
      ou[i][k] = (float)pu[cc_];
    }
  }
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int r = wave + 4 * i;
    if (t0 + r >= Tn) continue;      // wave-uniform
    float og[CPL];
    float s_c = 0.f, s_g = 0.f;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const bool cv = lane + 64 * k < C;
      og[k] = (nrep == 2) ? ou[i][k] + (oc[i][k] - ou[i][k]) * scale : oc[i][k];
      s_c += cv ? oc[i][k] : 0.f;
      s_g += cv ? og[k] : 0.f;
    }
    if (nrep == 2 && scale_cfg) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        s_c += __shfl_xor(s_c, off);
        s_g += __shfl_xor(s_g, off);
      }
      const float m_c = s_c / (float)C, m_g = s_g / (float)C;
      float v_c = 0.f, v_g = 0.f;
#pragma unroll
      for (int k = 0; k < CPL; ++k) {
        const bool cv = lane + 64 * k < C;
        const float dc = oc[i][k] - m_c, dg = og[k] - m_g;
        v_c += cv ? dc * dc : 0.f;
        v_g += cv ? dg * dg : 0.f;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        v_c += __shfl_xor(v_c, off);
        v_g += __shfl_xor(v_g, off);
      }
      const float ratio = sqrtf(v_c / (float)(C - 1)) / sqrtf(v_g / (float)(C - 1));   // unbiased std (torch.std)
#pragma unroll
      for (int k = 0; k < CPL; ++k) og[k] = phi * (og[k] * ratio) + (1.0f - phi) * og[k];
    }
#pragma unroll
    for (int k = 0; k < CPL; ++k)
      if (lane + 64 * k < C) tile[(lane + 64 * k) * 33 + r] = og[k];
  }
  __syncthreads();
  // the step counter advances here instead of in a launch of its own (jen1_cfg_ddim_step_adv): every thread of the grid has read it
  // once its block is past the barrier above, so the block that takes the last ticket may write it (and leaves the ticket at zero)
  if (adv_ticket != nullptr && threadIdx.x == 0) {
    const unsigned nblk = gridDim.x * gridDim.y;
    if (atomicAdd(adv_ticket, 1u) == nblk - 1u) {
      __hip_atomic_fetch_add(adv_step, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      adv_ticket[0] = 0u;
    }
  }
  // phase 2: [C][T]-major elementwise, 8 channels per pass with their loads issued together
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int t = t0 + tx;
  if (t >= Tn) return;
  const float sr = cf[0], srm1 = cf[1], sa_n = cf[2], cc = cf[3], sg = cf[4], last = cf[5], sa_t = cf[6], s1m_t = cf[7];
  for (int cb = 0; cb < C; cb += 64) {
    float xv[8], nv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = cb + ty + 8 * k;
      const size_t idx = ((size_t)b * C + (c < C ? c : 0)) * Tn + t;
      xv[k] = DDIM ? x[idx] : 0.f;
      nv[k] = (DDIM && noise) ? noise[idx] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = cb + ty + 8 * k;
      if (c >= C) continue;
      const size_t idx = ((size_t)b * C + c) * Tn + t;
      const float o = tile[c * 33 + tx];
      if (!DDIM) {
        x_out[idx] = o;
        continue;
      }
      float x0, eps;
      if (objective == 0) {          // 'noise'
        eps = o;
        x0 = sr * xv[k] - srm1 * eps;
        if (clip_x0) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
      } else if (objective == 1) {   // 'x0'
        x0 = o;
        if (clip_x0) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
        eps = (sr * xv[k] - x0) / srm1;
      } else {                       // 'v'
        x0 = sa_t * xv[k] - s1m_t * o;
        if (clip_x0) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
        eps = (sr * xv[k] - x0) / srm1;
      }
      float xn;
      if (last == 3.f) {
        // VDM row {alpha_t, sigma_t, alpha_next, sigma_next} (vdm/vdm.py:52-55, v objective, no clamp), evaluated as written:
        // x_pred = alpha x - sigma v; noise_pred = sigma x + alpha v; x = alpha' x_pred + sigma' noise_pred
        x0 = sr * xv[k] - srm1 * o;
        eps = srm1 * xv[k] + sr * o;
        xn = sa_n * x0 + cc * eps;
      } else if (last == 1.f) xn = x0;                                     // DDIM's final step (gdm.py:207-209)
      else if (last == 2.f) xn = x0 * sa_n + cc * xv[k] + sg * nv[k];      // DDPM row: posterior mean + sd * noise (gdm.py:144-163)
      else xn = x0 * sa_n + cc * eps + sg * nv[k];
      x_out[idx] = xn;
      if (eps_out) eps_out[idx] = eps;
      if (x0_out) x0_out[idx] = x0;
    }
  }
}


// ---- the same step for C % 8 == 0 (the latent channels of the model: 128): 16-byte row loads, no LDS crossbar -----------------------
// Half a wave owns a (b, t) row, a lane 8 consecutive channels of it; the row reductions of the std rescale are DPP row steps and one
// v_permlane16_swap; x_t and the step's noise -- which do not depend on the network output -- are requested before anything else.
template <int CTRL>
__device__ __forceinline__ float edpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float half_wave_sum(float v) {           // over the 32 lanes of a half wave, every lane gets the sum
  v += edpp<0xB1>(v);                                                // lanes ^ 1
  v += edpp<0x4E>(v);                                                // lanes ^ 2
  v += edpp<0x141>(v);                                               // row_half_mirror
  v += edpp<0x140>(v);                                               // row_mirror
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);              // lanes ^ 16
}

// (bx, by) of (nbx, nby): the block's place in the step's grid -- the kernel's own grid, or the leading blocks of step_tail_kernel's
template <typename T, bool DDIM>
__device__ __forceinline__ void cfg_step_vec_body(const T* __restrict__ net, const float* __restrict__ x,
                                                  const float* __restrict__ noise, const float* __restrict__ coef,
                                                  float* __restrict__ x_out, float* __restrict__ eps_out,
                                                  float* __restrict__ x0_out, const int32_t* step_idx,
                                                  int B, int C, int Tn, int ld, int nrep,
                                                  float scale, int scale_cfg, float phi, int objective, int clip_x0,
                                                  int32_t* adv_step, unsigned* __restrict__ adv_ticket,
                                                  T* __restrict__ rows, float* __restrict__ parts, int ld_rows,
                                                  const int bx, const int by, const int nbx, const int nby) {
  // (no implicit multiply-add fusion in here: the body is inlined into two kernels and, whichever one runs a step, the trajectory must
  // come out on the same bits; the statistics sums below fuse explicitly, as pack_input_kernel's do)
#pragma clang fp contract(off)
  extern __shared__ float tile[];   // [C][33] + one word per 8 channels
  float cf[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (DDIM) {
    if (step_idx) {                  // (see cfg_step_kernel: the counter may be advanced by this launch's last block)
      const int st = __hip_atomic_load(step_idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      coef += (size_t)st * 8;
      if (noise) noise += (size_t)st * B * C * Tn;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) cf[i] = coef[i];
  }
  const int t0 = bx * 32, b = by;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int t = t0 + tx;
  const int tcl = t < Tn ? t : Tn - 1;
  // phase-2 operands: [C][T]-major, 8 channels per 64-channel block and thread
  constexpr int NBLK = 4;           // C <= 256
  float xv[NBLK][8], nv[NBLK][8];
#pragma unroll
  for (int blk = 0; blk < NBLK; ++blk) {
    if (blk * 64 >= C) continue;     // uniform
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = blk * 64 + ty + 8 * k;
      const size_t idx = ((size_t)b * C + (c < C ? c : 0)) * Tn + tcl;
      xv[blk][k] = DDIM ? x[idx] : 0.f;
      nv[blk][k] = (DDIM && noise && cf[4] != 0.f) ? noise[idx] : 0.f;   // (a deterministic row -- sigma = 0 -- reads no noise)
    }
  }
  // phase 1: 8 rows per pass (4 waves x 2 half waves), every load of the block in flight before the first reduction
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int half = lane >> 5, c8 = (lane & 31) * 8;
  const bool cv = c8 < C;
  const int cl = cv ? c8 : 0;
  float oc[4][8], ou[4][8];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = p * 8 + wave * 2 + half;
    const int tr = (t0 + r < Tn) ? t0 + r : Tn - 1;
    load8(net + ((size_t)b * Tn + tr) * ld + cl, oc[p]);
    if (nrep == 2) load8(net + ((size_t)(B + b) * Tn + tr) * ld + cl, ou[p]);
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = p * 8 + wave * 2 + half;
    float og[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) og[j] = (nrep == 2) ? ou[p][j] + (oc[p][j] - ou[p][j]) * scale : oc[p][j];
    if (nrep == 2 && scale_cfg) {
      float s_c = 0.f, s_g = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) { s_c += oc[p][j]; s_g += og[j]; }
      s_c = half_wave_sum(cv ? s_c : 0.f);
      s_g = half_wave_sum(cv ? s_g : 0.f);
      const float m_c = s_c / (float)C, m_g = s_g / (float)C;
      float v_c = 0.f, v_g = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float dc = oc[p][j] - m_c, dg = og[j] - m_g;
        v_c += dc * dc;
        v_g += dg * dg;
      }
      v_c = half_wave_sum(cv ? v_c : 0.f);
      v_g = half_wave_sum(cv ? v_g : 0.f);
      const float ratio = sqrtf(v_c / (float)(C - 1)) / sqrtf(v_g / (float)(C - 1));   // unbiased std (torch.std)
#pragma unroll
      for (int j = 0; j < 8; ++j) og[j] = phi * (og[j] * ratio) + (1.0f - phi) * og[j];
    }
    if (cv) {
      // (lane i of a half wave writes channel rows 8 i .. 8 i + 7; the extra word per 8 channels puts the 32 lanes in 32 banks)
#pragma unroll
      for (int j = 0; j < 8; ++j) tile[(c8 + j) * 33 + (c8 >> 3) + r] = og[j];
    }
  }
  __syncthreads();
  if (adv_ticket != nullptr && threadIdx.x == 0) {
    const unsigned nblk = (unsigned)(nbx * nby);
    if (atomicAdd(adv_ticket, 1u) == nblk - 1u) {
      __hip_atomic_fetch_add(adv_step, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      adv_ticket[0] = 0u;
    }
  }
  const bool pack = DDIM && rows != nullptr;     // (uniform)
  if (t >= Tn && !pack) return;
  const float sr = cf[0], srm1 = cf[1], sa_n = cf[2], cc = cf[3], sg = cf[4], last = cf[5], sa_t = cf[6], s1m_t = cf[7];
#pragma unroll
  for (int blk = 0; blk < NBLK; ++blk) {
    if (blk * 64 >= C) continue;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = blk * 64 + ty + 8 * k;
      if (c >= C) continue;
      if (t >= Tn) {                 // (pack only: the columns past the end count as zeros in the next step's statistics)
        tile[c * 33 + (c >> 3) + tx] = 0.f;
        continue;
      }
      const size_t idx = ((size_t)b * C + c) * Tn + t;
      const float o = tile[c * 33 + (c >> 3) + tx];
      if (!DDIM) {
        x_out[idx] = o;
        continue;
      }
      const float xt = xv[blk][k];
      float x0, eps;
      if (objective == 0) {          // 'noise'
        eps = o;
        x0 = sr * xt - srm1 * eps;
        if (clip_x0) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
      } else if (objective == 1) {   // 'x0'
        x0 = o;
        if (clip_x0) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
        eps = (sr * xt - x0) / srm1;
      } else {                       // 'v'
        x0 = sa_t * xt - s1m_t * o;
        if (clip_x0) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
        eps = (sr * xt - x0) / srm1;
      }
      float xn;
      if (last == 3.f) {             // VDM row (see cfg_step_kernel)
        x0 = sr * xt - srm1 * o;
        eps = srm1 * xt + sr * o;
        xn = sa_n * x0 + cc * eps;
      } else if (last == 1.f) xn = x0;
      else if (last == 2.f) xn = x0 * sa_n + cc * xt + sg * nv[blk][k];
      else xn = x0 * sa_n + cc * eps + sg * nv[blk][k];
      x_out[idx] = xn;
      if (eps_out) eps_out[idx] = eps;
      if (x0_out) x0_out[idx] = x0;
      if (pack) tile[c * 33 + (c >> 3) + tx] = xn;       // (the element this thread read: nobody else touches it before the barrier)
    }
  }
  if (!pack) return;
  // ---- the next step's network input, written here instead of by a pack_input launch at its head (model.py:240, :332-349): the new
  // latents go back through the tile into the channel-last rows [nrep * B][T][ld_rows] (the concat-context channels behind them do not
  // change between steps), and their per-channel (sum, sumsq) over this block's 32 time steps into the partials that
  // gn_stats_from_parts_kernel adds in a fixed order -- same values, same order as pack_input_kernel
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = p * 8 + wave * 2 + half;
    if (!cv || t0 + r >= Tn) continue;
    float o8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o8[j] = tile[(c8 + j) * 33 + (c8 >> 3) + r];
    for (int rep = 0; rep < nrep; ++rep) store8(rows + ((size_t)(rep * B + b) * Tn + t0 + r) * ld_rows + c8, o8);
  }
  if (threadIdx.x < C) {
    const int c = threadIdx.x;
    float sv = 0.f, sq = 0.f;
#pragma unroll 8
    for (int j = 0; j < 32; ++j) {
      const float v = tile[c * 33 + (c >> 3) + j];
      sv += v;
      sq = __builtin_fmaf(v, v, sq);
    }
    *reinterpret_cast<float2*>(parts + (((size_t)b * nbx + bx) * ld_rows + c) * 2) = make_float2(sv, sq);
  }
}


template <typename T, bool DDIM>
__global__ __launch_bounds__(256) void cfg_step_vec_kernel(const T* __restrict__ net, const float* __restrict__ x,
                                                            const float* __restrict__ noise, const float* __restrict__ coef,
                                                            float* __restrict__ x_out, float* __restrict__ eps_out,
                                                            float* __restrict__ x0_out, const int32_t* step_idx,
                                                            int B, int C, int Tn, int ld, int nrep,
                                                            float scale, int scale_cfg, float phi, int objective, int clip_x0,
                                                            int32_t* adv_step, unsigned* __restrict__ adv_ticket,
                                                            T* __restrict__ rows, float* __restrict__ parts, int ld_rows) {
  cfg_step_vec_body<T, DDIM>(net, x, noise, coef, x_out, eps_out, x0_out, step_idx, B, C, Tn, ld, nrep, scale, scale_cfg, phi, objective,
                             clip_x0, adv_step, adv_ticket, rows, parts, ld_rows, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y);
}

// ---- the tail of a replayed sampler step as ONE launch: the first nbx * nby blocks are the CFG / DDIM step + the next step's packed input
// (above), the blocks behind them set every tensor of the next step's persistent launches to the all-ones sentinel and zero the
// statistics arena (deep_kernel.hip's poison_kernel: same table, same 8 blocks per row).  Nothing orders the two jobs -- the step
// reads the network output and the latents, the sentinel rows are the activations in between -- and one is a latency chain of a few
// loads per thread while the other streams ~60 MB of stores: side by side they last as long as the longer one.
struct TailPoisonEntry {
  unsigned long long ptr, bytes;          // bytes: a multiple of 16 (= deep_kernel.hip PoisonEntry)
};
struct TailArgs {
  const void* net; const float* x; const float* noise; const float* coef; float* x_out; int32_t* step_idx; unsigned* ticket;
  void* rows; float* parts; int ld_rows, B, C, Tn, ld, nrep; float scale; int scale_cfg; float phi; int objective, clip_x0;
  const TailPoisonEntry* tab; int n_tab, z_rows; unsigned* sync; void* zero_ptr; unsigned long long zero_bytes;
};
template <typename T>
__global__ __launch_bounds__(256) void step_tail_kernel(const TailArgs a) {
  const int nbx = (a.Tn + 31) / 32, nstep = nbx * a.B;
  if ((int)blockIdx.x < nstep) {
    cfg_step_vec_body<T, true>((const T*)a.net, a.x, a.noise, a.coef, a.x_out, nullptr, nullptr, a.step_idx, a.B, a.C, a.Tn, a.ld, a.nrep,
                               a.scale, a.scale_cfg, a.phi, a.objective, a.clip_x0, a.step_idx, a.ticket, (T*)a.rows, a.parts, a.ld_rows,
                               (int)blockIdx.x % nbx, (int)blockIdx.x / nbx, nbx, a.B);
    return;
  }
  const int pid = (int)blockIdx.x - nstep, px = pid & 7, py = pid >> 3;
  if (py >= a.n_tab) {
    uint4* z = reinterpret_cast<uint4*>(a.zero_ptr);
    const size_t n = a.zero_bytes >> 4;
    const size_t nblk = (size_t)8 * a.z_rows, blk = (size_t)(py - a.n_tab) * 8 + px;
    for (size_t i = blk * 256 + threadIdx.x; i < n; i += nblk * 256) z[i] = make_uint4(0u, 0u, 0u, 0u);
    return;
  }
  const TailPoisonEntry e = a.tab[py];
  uint4* p = reinterpret_cast<uint4*>(e.ptr);
  const size_t n = e.bytes >> 4;
  const uint4 ones = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
  for (size_t i = (size_t)px * 256 + threadIdx.x; i < n; i += (size_t)8 * 256) p[i] = ones;
  if (a.sync && pid == 0 && threadIdx.x == 0) a.sync[0] = 0u;
}

}  // namespace

extern "C" int jen1_pack_input(const float* x, const float* ctx, void* y, float* gn_stats, int B, int C, int Cc, int T,
                               int ld, int nrep, int dtype, void* stream) {
  JEN1_CHECK(x && y && (Cc == 0 || ctx), "pack_input: null pointer");
  JEN1_CHECK(ld % 32 == 0 && ld >= C + Cc && nrep >= 1, "pack_input: ld=%d must be a multiple of 32 and >= %d", ld, C + Cc);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((T + 31) / 32, ld / 32, B);
  if (dtype == JEN1_F32) hipLaunchKernelGGL(pack_input_kernel<float>, grid, dim3(256), 0, s, x, ctx, (float*)y, gn_stats, B, C, Cc, T, ld, nrep, (float*)nullptr);
  else if (dtype == JEN1_BF16) hipLaunchKernelGGL(pack_input_kernel<bf16_t>, grid, dim3(256), 0, s, x, ctx, (bf16_t*)y, gn_stats, B, C, Cc, T, ld, nrep, (float*)nullptr);
  else return jen1_set_error("pack_input: bad dtype");
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_pack_input_parts(const float* x, const float* ctx, void* y, float* parts, int B, int C, int Cc, int T, int ld, int nrep,
                                     int dtype, void* stream) {
  JEN1_CHECK(x && y && parts && (Cc == 0 || ctx), "pack_input_parts: null pointer");
  JEN1_CHECK(ld % 32 == 0 && ld >= C + Cc && nrep >= 1, "pack_input_parts: ld=%d must be a multiple of 32 and >= %d", ld, C + Cc);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((T + 31) / 32, ld / 32, B);
  if (dtype == JEN1_F32) hipLaunchKernelGGL(pack_input_kernel<float>, grid, dim3(256), 0, s, x, ctx, (float*)y, (float*)nullptr, B, C, Cc, T, ld, nrep, parts);
  else if (dtype == JEN1_BF16) hipLaunchKernelGGL(pack_input_kernel<bf16_t>, grid, dim3(256), 0, s, x, ctx, (bf16_t*)y, (float*)nullptr, B, C, Cc, T, ld, nrep, parts);
  else return jen1_set_error("pack_input_parts: bad dtype");
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_gn_stats_from_parts(const float* parts, float* gn_stats, int B, int T, int ld, int nrep, void* stream) {
  JEN1_CHECK(parts && gn_stats && B >= 1 && T >= 1 && ld % JEN1_FINE_GROUPS == 0 && ld <= 4096 && nrep >= 1, "gn_stats_from_parts: bad arguments");
  const int nt = ld >= 512 ? 512 : ((ld + 63) / 64) * 64;
  hipLaunchKernelGGL(gn_stats_from_parts_kernel, dim3(B), dim3(nt), (size_t)ld * 8, reinterpret_cast<hipStream_t>(stream), parts, gn_stats, B, (T + 31) / 32, ld, nrep);
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_unpack_output(const void* y, float* out, int B, int C, int T, int ld, int dtype, void* stream) {
  JEN1_CHECK(y && out, "unpack_output: null pointer");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((T + 31) / 32, (C + 31) / 32, B);
  if (dtype == JEN1_F32) hipLaunchKernelGGL(unpack_output_kernel<float>, grid, dim3(256), 0, s, (const float*)y, out, C, T, ld);
  else if (dtype == JEN1_BF16) hipLaunchKernelGGL(unpack_output_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)y, out, C, T, ld);
  else return jen1_set_error("unpack_output: bad dtype");
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_row_stats(const void* x, float* stats, int rows, int C, int ld, int dtype, void* stream) {
  JEN1_CHECK(x && stats && rows >= 1 && C >= 1 && ld >= C, "row_stats: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((rows + 3) / 4);
  if (dtype == JEN1_F32) hipLaunchKernelGGL(row_stats_kernel<float>, grid, dim3(256), 0, s, (const float*)x, stats, rows, C, ld);
  else if (dtype == JEN1_BF16) hipLaunchKernelGGL(row_stats_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, stats, rows, C, ld);
  else return jen1_set_error("row_stats: bad dtype");
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_gn_stats(const void* x, float* stats, int B, int L, int ld, int dtype, void* stream) {
  JEN1_CHECK(x && stats && B >= 1 && L >= 1 && ld >= JEN1_FINE_GROUPS && ld % JEN1_FINE_GROUPS == 0, "gn_stats: bad arguments (ld=%d)", ld);
  JEN1_CHECK((int64_t)L * (ld / JEN1_FINE_GROUPS) < ((int64_t)1 << 31), "gn_stats: %d rows are too many for 32-bit element indices", L);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(JEN1_FINE_GROUPS, B);
  if (dtype == JEN1_F32) hipLaunchKernelGGL(gn_stats_kernel<float>, grid, dim3(256), 0, s, (const float*)x, stats, L, ld);
  else if (dtype == JEN1_BF16) hipLaunchKernelGGL(gn_stats_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, stats, L, ld);
  else return jen1_set_error("gn_stats: bad dtype");
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_time_features(const int64_t* t, const float* freq, const float* w, const float* bias, float* out,
                                  int n, int half, int out_features, void* stream) {
  JEN1_CHECK(t && freq && w && bias && out && n >= 1 && half >= 1 && out_features >= 1, "time_features: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int gy = (out_features + 3) / 4;
  if (gy > 64) gy = 64;
  hipLaunchKernelGGL(time_features_kernel<int64_t>, dim3(n, gy), dim3(256), sizeof(float) * (2 * half + 1), s, t, freq, w, bias, out, half, out_features);
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_time_features_f32(const float* t, const float* freq, const float* w, const float* bias, float* out,
                                      int n, int half, int out_features, void* stream) {
  JEN1_CHECK(t && freq && w && bias && out && n >= 1 && half >= 1 && out_features >= 1, "time_features_f32: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int gy = (out_features + 3) / 4;
  if (gy > 64) gy = 64;
  hipLaunchKernelGGL(time_features_kernel<float>, dim3(n, gy), dim3(256), sizeof(float) * (2 * half + 1), s, t, freq, w, bias, out, half, out_features);
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_linear_f32(const float* x, const float* w, const float* bias, float* y, int n, int in_features,
                               int out_features, int act, void* stream) {
  JEN1_CHECK(x && w && y && n >= 1 && in_features >= 1 && out_features >= 1, "linear_f32: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(linear_f32_kernel, dim3((out_features + 3) / 4), dim3(256), 0, s, x, w, bias, y, n, in_features, out_features, act);
  JEN1_HIP(hipGetLastError());
  return 0;
}

template <bool DDIM>
static int launch_cfg(const void* net, const float* x, const float* noise, const float* coef, float* x_out, float* eps_out,
                      float* x0_out, const int32_t* step_idx, int B, int C, int T, int ld, int nrep, float scale, int scale_cfg, float phi,
                      int objective, int clip_x0, int dtype, void* stream, int32_t* adv_step = nullptr, unsigned* adv_ticket = nullptr,
                      void* rows = nullptr, float* parts = nullptr, int ld_rows = 0) {
  JEN1_CHECK(net && x_out, "cfg step: null pointer");
  JEN1_CHECK(nrep == 1 || nrep == 2, "cfg step: nrep must be 1 or 2");
  JEN1_CHECK(C >= 2 && C <= 256 && ld >= C, "cfg step: C must be in [2, 256]");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((T + 31) / 32, B);
  const size_t lds = sizeof(float) * ((size_t)C * 33 + C / 8 + 1);
  // rows of 8-channel vectors on 16-byte boundaries: the vector form (JEN1_CFG_STEP_SCALAR=1 keeps the general kernel: A-B switch)
  static const bool scalar_only = getenv("JEN1_CFG_STEP_SCALAR") != nullptr;
  const int esz = dtype == JEN1_F32 ? 4 : 2;
  const bool vec = !scalar_only && C % 8 == 0 && ld % 8 == 0 && ((uintptr_t)net & 15) == 0 && ((size_t)ld * esz) % 16 == 0;
  if (rows) {
    JEN1_CHECK(vec && DDIM && parts, "cfg step + pack: needs the vector form (C % 8 == 0, 16-byte rows) and the partials buffer");
    JEN1_CHECK(ld_rows >= C && ((size_t)ld_rows * esz) % 16 == 0 && ((uintptr_t)rows & 15) == 0 && ((uintptr_t)parts & 7) == 0,
               "cfg step + pack: rows must be 16-byte aligned with ld_rows >= C");
  }
  if (vec && dtype == JEN1_F32) {
    auto kern = cfg_step_vec_kernel<float, DDIM>;
    JEN1_MAX_LDS_ONCE(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, (const float*)net, x, noise, coef, x_out, eps_out, x0_out, step_idx, B, C, T, ld, nrep, scale, scale_cfg, phi, objective, clip_x0, adv_step, adv_ticket, (float*)rows, parts, ld_rows);
  } else if (vec && dtype == JEN1_BF16) {
    auto kern = cfg_step_vec_kernel<bf16_t, DDIM>;
    JEN1_MAX_LDS_ONCE(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, (const bf16_t*)net, x, noise, coef, x_out, eps_out, x0_out, step_idx, B, C, T, ld, nrep, scale, scale_cfg, phi, objective, clip_x0, adv_step, adv_ticket, (bf16_t*)rows, parts, ld_rows);
  } else if (dtype == JEN1_F32) {
    auto kern = cfg_step_kernel<float, DDIM>;
    JEN1_MAX_LDS_ONCE(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, (const float*)net, x, noise, coef, x_out, eps_out, x0_out, step_idx, B, C, T, ld, nrep, scale, scale_cfg, phi, objective, clip_x0, adv_step, adv_ticket);
  } else if (dtype == JEN1_BF16) {
    auto kern = cfg_step_kernel<bf16_t, DDIM>;
    JEN1_MAX_LDS_ONCE(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, (const bf16_t*)net, x, noise, coef, x_out, eps_out, x0_out, step_idx, B, C, T, ld, nrep, scale, scale_cfg, phi, objective, clip_x0, adv_step, adv_ticket);
  } else {
    return jen1_set_error("cfg step: bad dtype");
  }
  JEN1_HIP(hipGetLastError());
  return 0;
}

__global__ void step_advance_kernel(int32_t* step_idx) { step_idx[0] += 1; }

extern "C" int jen1_step_advance(int32_t* step_idx, void* stream) {
  JEN1_CHECK(step_idx, "step_advance: null pointer");
  hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, reinterpret_cast<hipStream_t>(stream), step_idx);
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_cfg_ddim_step(const void* net, const float* x, const float* noise, const float* coef, float* x_out,
                                  float* eps_out, float* x0_out, const int32_t* step_idx, int B, int C, int T, int ld,
                                  int nrep, float embedding_scale, int scale_cfg, float scale_phi, int objective,
                                  int clip_x0, int dtype, void* stream) {
  JEN1_CHECK(x && coef, "cfg_ddim_step: null x/coef");
  JEN1_CHECK(objective >= 0 && objective <= 2, "cfg_ddim_step: bad objective");
  return launch_cfg<true>(net, x, noise, coef, x_out, eps_out, x0_out, step_idx, B, C, T, ld, nrep, embedding_scale, scale_cfg,
                          scale_phi, objective, clip_x0, dtype, stream);
}

extern "C" int jen1_cfg_ddim_step_adv(const void* net, const float* x, const float* noise, const float* coef, float* x_out,
                                      float* eps_out, float* x0_out, int32_t* step_idx, uint32_t* ticket, int B, int C, int T, int ld,
                                      int nrep, float embedding_scale, int scale_cfg, float scale_phi, int objective,
                                      int clip_x0, int dtype, void* stream) {
  JEN1_CHECK(x && coef && step_idx && ticket, "cfg_ddim_step_adv: null x / coef / step_idx / ticket");
  JEN1_CHECK(objective >= 0 && objective <= 2, "cfg_ddim_step_adv: bad objective");
  return launch_cfg<true>(net, x, noise, coef, x_out, eps_out, x0_out, step_idx, B, C, T, ld, nrep, embedding_scale, scale_cfg,
                          scale_phi, objective, clip_x0, dtype, stream, step_idx, ticket);
}

extern "C" int jen1_cfg_ddim_step_pack(const void* net, const float* x, const float* noise, const float* coef, float* x_out,
                                       int32_t* step_idx, uint32_t* ticket, void* rows, float* parts, int ld_rows, int B, int C, int T,
                                       int ld, int nrep, float embedding_scale, int scale_cfg, float scale_phi, int objective,
                                       int clip_x0, int dtype, void* stream) {
  JEN1_CHECK(x && coef && step_idx && ticket && rows && parts, "cfg_ddim_step_pack: null x / coef / step_idx / ticket / rows / parts");
  JEN1_CHECK(objective >= 0 && objective <= 2, "cfg_ddim_step_pack: bad objective");
  return launch_cfg<true>(net, x, noise, coef, x_out, nullptr, nullptr, step_idx, B, C, T, ld, nrep, embedding_scale, scale_cfg,
                          scale_phi, objective, clip_x0, dtype, stream, step_idx, ticket, rows, parts, ld_rows);
}

extern "C" int jen1_step_tail(const void* net, const float* x, const float* noise, const float* coef, float* x_out, int32_t* step_idx,
                             uint32_t* ticket, void* rows, float* parts, int ld_rows, int B, int C, int T, int ld, int nrep,
                             float embedding_scale, int scale_cfg, float scale_phi, int objective, int clip_x0, int dtype,
                             const void* poison_table, int n_rows, uint32_t* sync, void* zero_ptr, int64_t zero_bytes, void* stream) {
  JEN1_CHECK(net && x && coef && x_out && step_idx && ticket && rows && parts, "step_tail: null net / x / coef / x_out / step_idx / ticket / rows / parts");
  JEN1_CHECK(objective >= 0 && objective <= 2 && (nrep == 1 || nrep == 2), "step_tail: bad objective / nrep");
  JEN1_CHECK(dtype == JEN1_F32 || dtype == JEN1_BF16, "step_tail: bad dtype");
  const int esz = dtype == JEN1_F32 ? 4 : 2;
  JEN1_CHECK(C >= 8 && C <= 256 && C % 8 == 0 && ld >= C && ld % 8 == 0 && ((uintptr_t)net & 15) == 0 && ((size_t)ld * esz) % 16 == 0,
             "step_tail: the network output must be 16-byte rows of C %% 8 == 0 channels (C <= 256)");
  JEN1_CHECK(ld_rows >= C && ((size_t)ld_rows * esz) % 16 == 0 && ((uintptr_t)rows & 15) == 0 && ((uintptr_t)parts & 7) == 0,
             "step_tail: rows must be 16-byte aligned with ld_rows >= C");
  JEN1_CHECK(poison_table && n_rows >= 1 && n_rows <= 60000, "step_tail: bad sentinel table");
  JEN1_CHECK(zero_ptr && zero_bytes > 0 && (zero_bytes & 15) == 0 && ((uintptr_t)zero_ptr & 15) == 0, "step_tail: the zeroed area must be 16-byte aligned and sized");
  int zrows = (int)((zero_bytes + (1 << 18) - 1) >> 18);
  zrows = zrows < 1 ? 1 : (zrows > 64 ? 64 : zrows);
  TailArgs a;
  a.net = net; a.x = x; a.noise = noise; a.coef = coef; a.x_out = x_out; a.step_idx = step_idx; a.ticket = ticket;
  a.rows = rows; a.parts = parts; a.ld_rows = ld_rows; a.B = B; a.C = C; a.Tn = T; a.ld = ld; a.nrep = nrep;
  a.scale = embedding_scale; a.scale_cfg = scale_cfg; a.phi = scale_phi; a.objective = objective; a.clip_x0 = clip_x0;
  a.tab = reinterpret_cast<const TailPoisonEntry*>(poison_table); a.n_tab = n_rows; a.z_rows = zrows; a.sync = sync;
  a.zero_ptr = zero_ptr; a.zero_bytes = (unsigned long long)zero_bytes;
  const int nstep = ((T + 31) / 32) * B;
  const dim3 grid(nstep + 8 * (n_rows + zrows));
  const size_t lds = sizeof(float) * ((size_t)C * 33 + C / 8 + 1);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == JEN1_F32) {
    auto kern = step_tail_kernel<float>;
    JEN1_MAX_LDS_ONCE(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  } else {
    auto kern = step_tail_kernel<bf16_t>;
    JEN1_MAX_LDS_ONCE(kern, 160 * 1024);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  }
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_cfg_combine(const void* net, float* out, int B, int C, int T, int ld, float embedding_scale,
                                int scale_cfg, float scale_phi, int dtype, void* stream) {
  return launch_cfg<false>(net, nullptr, nullptr, nullptr, out, nullptr, nullptr, nullptr, B, C, T, ld, 2, embedding_scale,
                           scale_cfg, scale_phi, 0, 0, dtype, stream);
}
