// Normalisation / pointwise / softmax kernels of the training path, forward and backward (include/jen1_train.h).
// Channel-last rows x[row][c]; consecutive threads take consecutive channels, so every access is coalesced.
// All of these are HBM-bound streaming kernels: one read of each input, one write of each output, float32 math.
#include <cstdlib>
#include "common.h"
#include "jen1_train.h"

namespace {

constexpr int NT = 256;

template <typename T> __device__ __forceinline__ float ldf(const T* p, long long i) { return (float)p[i]; }
template <typename T> __device__ __forceinline__ void stf(T* p, long long i, float v) { p[i] = (T)v; }

__device__ __forceinline__ float silu_grad(float f) {
  const float s = 1.0f / (1.0f + expf(-f));
  return s * (1.0f + f * (1.0f - s));
}
// bf16 mode of the one-launch GroupNorm kernels: v_exp_f32 / v_rcp_f32 (1 ulp each, against 8 mantissa bits of the tensors) instead of
// expf and an IEEE division -- on the long levels these kernels are bound by their vector arithmetic (128 workgroups = half of the
// chip's CUs, ~80 instructions per element and pass before), not by memory.  float32 mode keeps the exact forms (parity runs).
template <typename T>
__device__ __forceinline__ float sigmoid_t(float x) {
  if constexpr (sizeof(T) == 2) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
  else return 1.0f / (1.0f + expf(-x));
}
template <typename T>
__device__ __forceinline__ float silu_t(float x) {
  if constexpr (sizeof(T) == 2) return x * sigmoid_t<T>(x);
  else return silu_precise(x);
}
template <typename T>
__device__ __forceinline__ float silu_grad_t(float f) {
  const float s = sigmoid_t<T>(f);
  return s * (1.0f + f * (1.0f - s));
}
__device__ __forceinline__ float gelu_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// thread layout for per-(b, c) reductions over t: CT channels x (NT / CT) rows per block
struct RedGeom {
  int CT, RT;      // channels per block, rows in flight per block
  int rows_per_block;
};

// ---------------------------------------------------------------- GroupNorm
template <typename T>
__global__ __launch_bounds__(NT) void gn_sums_kernel(const void* x_, float* __restrict__ sums, int L, int C, int ld,
                                                      int groups, int cpg, int CT, int rows_per_block) {
  const T* x = reinterpret_cast<const T*>(x_);
  const int b = blockIdx.z;
  const int c = blockIdx.y * CT + threadIdx.x % CT;
  const int ty = threadIdx.x / CT, RT = NT / CT;
  const int t0 = blockIdx.x * rows_per_block, t1 = min(L, t0 + rows_per_block);
  float s = 0.f, ss = 0.f;
  if (c < C) {
    const T* xp = x + (long long)b * L * ld + c;
    for (int t = t0 + ty; t < t1; t += RT) {
      const float v = (float)xp[(long long)t * ld];
      s += v;
      ss += v * v;
    }
  }
  __shared__ float acc[64];   // [groups touched by this block][2]; at most 32 groups
  // groups covered by this block: c range [blockIdx.y*CT, +CT)
  const int g_lo = (blockIdx.y * CT) / cpg;
  for (int i = threadIdx.x; i < 64; i += NT) acc[i] = 0.f;
  __syncthreads();
  if (c < C) {
    const int gi = c / cpg - g_lo;
    if (gi < 32) { atomicAdd(&acc[2 * gi], s); atomicAdd(&acc[2 * gi + 1], ss); }
    else { atomicAdd(&sums[((long long)b * groups + c / cpg) * 2], s); atomicAdd(&sums[((long long)b * groups + c / cpg) * 2 + 1], ss); }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 32; i += NT) {
    const int g = g_lo + i;
    if (g < groups && (acc[2 * i] != 0.f || acc[2 * i + 1] != 0.f)) {
      atomicAdd(&sums[((long long)b * groups + g) * 2], acc[2 * i]);
      atomicAdd(&sums[((long long)b * groups + g) * 2 + 1], acc[2 * i + 1]);
    }
  }
}

struct GnDev {
  const void* x; const void* dy; const float* sums; const float* gamma; const float* beta; const void* film;
  void* y; float* P; float* Gm; float* dgamma; float* dbeta; float* dfilm;
  const void* dx_add;      // backward: added to dx (the gradient that reached the same tensor along another branch), or NULL
  const void* dx_add2;     // ... and a third branch (a ResnetBlock1d input that is also a skip connection of the U-Net), or NULL
  int B, L, C, ld, groups, cpg, film_ld, flags, film_bf16;
  float eps, inv_count;
};

// FiLM gradients of one channel: [B][2C] float32, or with flags bit1 a mirror of ``film`` (its dtype, its row stride: the slice of a
// buffer that holds the gradients of every block's FiLM projection side by side)
__device__ inline void store_dfilm(const GnDev& g, int b, int c, float d_scale, float d_shift) {
  if (g.flags & 2) {
    const long long o = (long long)b * g.film_ld + c;
    if (g.film_bf16) {
      bf16_t* d = reinterpret_cast<bf16_t*>(g.dfilm);
      d[o] = (bf16_t)d_scale; d[o + g.C] = (bf16_t)d_shift;
    } else { g.dfilm[o] = d_scale; g.dfilm[o + g.C] = d_shift; }
  } else {
    g.dfilm[(long long)b * 2 * g.C + c] = d_scale;
    g.dfilm[(long long)b * 2 * g.C + g.C + c] = d_shift;
  }
}

// (mean, rstd) of the group of channel c in batch element b
__device__ __forceinline__ void gn_moments(const GnDev& g, int b, int c, float& mean, float& rstd) {
  const float* s = g.sums + ((long long)b * g.groups + c / g.cpg) * 2;
  mean = s[0] * g.inv_count;
  const float var = fmaxf(s[1] * g.inv_count - mean * mean, 0.f);
  rstd = 1.0f / sqrtf(var + g.eps);
}

template <typename T>
__global__ __launch_bounds__(NT) void gn_apply_kernel(const GnDev g) {
  const long long total = (long long)g.B * g.L * g.C;
  const T* x = reinterpret_cast<const T*>(g.x);
  const T* film = reinterpret_cast<const T*>(g.film);
  T* y = reinterpret_cast<T*>(g.y);
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
    const int c = (int)(i % g.C);
    const long long row = i / g.C;
    const int b = (int)(row / g.L);
    float mean, rstd;
    gn_moments(g, b, c, mean, rstd);
    float n = ((float)x[row * g.ld + c] - mean) * rstd * g.gamma[c] + g.beta[c];
    if (film != nullptr) n = n * ((float)film[(long long)b * g.film_ld + c] + 1.0f) + (float)film[(long long)b * g.film_ld + g.C + c];
    if (g.flags & 1) n = silu_precise(n);
    y[row * g.ld + c] = (T)n;
  }
}

// P[b][c] = (sum dn, sum dn * xhat, sum df * n, sum df) over t
template <typename T>
__global__ __launch_bounds__(NT) void gn_bwd_sums_kernel(const GnDev g, int CT, int rows_per_block) {
  const int b = blockIdx.z;
  const int c = blockIdx.y * CT + threadIdx.x % CT;
  const int ty = threadIdx.x / CT, RT = NT / CT;
  const int t0 = blockIdx.x * rows_per_block, t1 = min(g.L, t0 + rows_per_block);
  if (c >= g.C) return;
  const T* x = reinterpret_cast<const T*>(g.x) + (long long)b * g.L * g.ld + c;
  const T* dy = reinterpret_cast<const T*>(g.dy) + (long long)b * g.L * g.ld + c;
  const T* film = reinterpret_cast<const T*>(g.film);
  float mean, rstd;
  gn_moments(g, b, c, mean, rstd);
  const float ga = g.gamma[c], be = g.beta[c];
  float sc1 = 1.0f, sh = 0.f;
  if (film != nullptr) { sc1 = (float)film[(long long)b * g.film_ld + c] + 1.0f; sh = (float)film[(long long)b * g.film_ld + g.C + c]; }
  float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
  for (int t = t0 + ty; t < t1; t += RT) {
    const float xh = ((float)x[(long long)t * g.ld] - mean) * rstd;
    const float n = xh * ga + be;
    const float f = n * sc1 + sh;
    float df = (float)dy[(long long)t * g.ld];
    if (g.flags & 1) df *= silu_grad(f);
    const float dn = df * sc1;
    p0 += dn; p1 += dn * xh; p2 += df * n; p3 += df;
  }
  float* P = g.P + ((long long)b * g.C + c) * 4;
  atomicAdd(P + 0, p0); atomicAdd(P + 1, p1); atomicAdd(P + 2, p2); atomicAdd(P + 3, p3);
}

// group means of d(xhat), FiLM gradients, parameter gradients -- no atomics:
//   blocks [0, B*G): one wave per (b, g) reduces its channels;  blocks [B*G, ..): one thread per channel sums over b
__global__ __launch_bounds__(64) void gn_bwd_finish_kernel(const GnDev g) {
  const int nbg = g.B * g.groups;
  if ((int)blockIdx.x < nbg) {
    const int b = blockIdx.x / g.groups, grp = blockIdx.x % g.groups;
    float m1 = 0.f, m2 = 0.f;
    for (int j = threadIdx.x; j < g.cpg; j += 64) {
      const int c = grp * g.cpg + j;
      const float* P = g.P + ((long long)b * g.C + c) * 4;
      const float ga = g.gamma[c];
      m1 += ga * P[0];
      m2 += ga * P[1];
      if (g.dfilm != nullptr) store_dfilm(g, b, c, P[2], P[3]);
    }
    m1 = wave_sum(m1); m2 = wave_sum(m2);
    if (threadIdx.x == 0) {
      g.Gm[2 * (long long)blockIdx.x] = m1 * g.inv_count;
      g.Gm[2 * (long long)blockIdx.x + 1] = m2 * g.inv_count;
    }
    return;
  }
  const int c = (blockIdx.x - nbg) * 64 + threadIdx.x;
  if (c >= g.C) return;
  float dg = 0.f, db = 0.f;
  for (int b = 0; b < g.B; ++b) {
    const float* P = g.P + ((long long)b * g.C + c) * 4;
    dg += P[1];
    db += P[0];
  }
  g.dgamma[c] += dg;       // the only writer of this entry in this launch
  g.dbeta[c] += db;
}

template <typename T>
__global__ __launch_bounds__(NT) void gn_bwd_dx_kernel(const GnDev g, void* dx_) {
  T* dx = reinterpret_cast<T*>(dx_);
  const long long total = (long long)g.B * g.L * g.C;
  const T* x = reinterpret_cast<const T*>(g.x);
  const T* dy = reinterpret_cast<const T*>(g.dy);
  const T* film = reinterpret_cast<const T*>(g.film);
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
    const int c = (int)(i % g.C);
    const long long row = i / g.C;
    const int b = (int)(row / g.L);
    float mean, rstd;
    gn_moments(g, b, c, mean, rstd);
    const float ga = g.gamma[c];
    const float xh = ((float)x[row * g.ld + c] - mean) * rstd;
    float sc1 = 1.0f, sh = 0.f;
    if (film != nullptr) { sc1 = (float)film[(long long)b * g.film_ld + c] + 1.0f; sh = (float)film[(long long)b * g.film_ld + g.C + c]; }
    float df = (float)dy[row * g.ld + c];
    if (g.flags & 1) df *= silu_grad((xh * ga + g.beta[c]) * sc1 + sh);
    const float dxh = df * sc1 * ga;
    const float* gm = g.Gm + ((long long)b * g.groups + c / g.cpg) * 2;
    float o = rstd * (dxh - gm[0] - xh * gm[1]);
    if (g.dx_add != nullptr) o += (float)reinterpret_cast<const T*>(g.dx_add)[row * g.ld + c];
    if (g.dx_add2 != nullptr) o += (float)reinterpret_cast<const T*>(g.dx_add2)[row * g.ld + c];
    dx[row * g.ld + c] = (T)o;
  }
}

// scratch reset as a kernel node (one launch zeroes up to two buffers)
__global__ __launch_bounds__(NT) void zero2_kernel(float* __restrict__ a, int na, float* __restrict__ b, int nb) {
  const int i = blockIdx.x * NT + threadIdx.x;
  if (i < na) a[i] = 0.f;
  if (i < nb) b[i] = 0.f;
}
int zero2(float* a, int na, float* b, int nb, hipStream_t s) {
  const int n = na > nb ? na : nb;
  hipLaunchKernelGGL(zero2_kernel, dim3((n + NT - 1) / NT), dim3(NT), 0, s, a, na, b, nb);
  JEN1_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- 8-wide variants (rows without padding, groups of a
// multiple of 8 channels): every thread moves 8 consecutive channels per access (16 bytes in bf16, 2 x 16 in float32)
// instead of one element -- these kernels are pure streaming, so bytes per instruction is what sets their speed.
template <typename T>
__global__ __launch_bounds__(NT) void gn_sums_vec_kernel(const void* x_, float* __restrict__ sums, int L, int C, int groups, int cpg,
                                                          int CT, int rows_per_block) {
  const T* x = reinterpret_cast<const T*>(x_);
  const int b = blockIdx.z, VPR = C >> 3;
  const int vc = blockIdx.y * CT + threadIdx.x % CT;          // which 8-channel vector of the row
  const int ty = threadIdx.x / CT, RT = NT / CT;
  const int t0 = blockIdx.x * rows_per_block, t1 = min(L, t0 + rows_per_block);
  float s = 0.f, ss = 0.f;
  if (vc < VPR) {
    const T* xp = x + (long long)b * L * C + vc * 8;
    for (int t = t0 + ty; t < t1; t += RT) {
      float v[8];
      load8(xp + (long long)t * C, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) { s += v[j]; ss += v[j] * v[j]; }
    }
  }
  __shared__ float acc[64];
  const int g_lo = (blockIdx.y * CT * 8) / cpg;
  for (int i = threadIdx.x; i < 64; i += NT) acc[i] = 0.f;
  __syncthreads();
  if (vc < VPR) {
    const int g = (vc * 8) / cpg, gi = g - g_lo;
    if (gi < 32) { atomicAdd(&acc[2 * gi], s); atomicAdd(&acc[2 * gi + 1], ss); }
    else { atomicAdd(&sums[((long long)b * groups + g) * 2], s); atomicAdd(&sums[((long long)b * groups + g) * 2 + 1], ss); }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 32; i += NT) {
    const int g = g_lo + i;
    if (g < groups && (acc[2 * i] != 0.f || acc[2 * i + 1] != 0.f)) {
      atomicAdd(&sums[((long long)b * groups + g) * 2], acc[2 * i]);
      atomicAdd(&sums[((long long)b * groups + g) * 2 + 1], acc[2 * i + 1]);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void gn_apply_vec_kernel(const GnDev g) {
  const int VPR = g.C >> 3;
  const long long total = (long long)g.B * g.L * VPR;
  const T* x = reinterpret_cast<const T*>(g.x);
  const T* film = reinterpret_cast<const T*>(g.film);
  T* y = reinterpret_cast<T*>(g.y);
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
    const int c0 = (int)(i % VPR) * 8;
    const long long row = i / VPR;
    const int b = (int)(row / g.L);
    float mean, rstd;
    gn_moments(g, b, c0, mean, rstd);
    float v[8], ga[8], be[8];
    load8(x + row * g.C + c0, v);
    load8(g.gamma + c0, ga);
    load8(g.beta + c0, be);
    float sc[8], sh[8];
    if (film != nullptr) {
      load8(film + (long long)b * g.film_ld + c0, sc);
      load8(film + (long long)b * g.film_ld + g.C + c0, sh);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float n = (v[j] - mean) * rstd * ga[j] + be[j];
      if (film != nullptr) n = n * (sc[j] + 1.0f) + sh[j];
      if (g.flags & 1) n = silu_precise(n);
      v[j] = n;
    }
    store8(y + row * g.C + c0, v);
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void gn_bwd_dx_vec_kernel(const GnDev g, void* dx_) {
  T* dx = reinterpret_cast<T*>(dx_);
  const int VPR = g.C >> 3;
  const long long total = (long long)g.B * g.L * VPR;
  const T* x = reinterpret_cast<const T*>(g.x);
  const T* dy = reinterpret_cast<const T*>(g.dy);
  const T* film = reinterpret_cast<const T*>(g.film);
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
    const int c0 = (int)(i % VPR) * 8;
    const long long row = i / VPR;
    const int b = (int)(row / g.L);
    float mean, rstd;
    gn_moments(g, b, c0, mean, rstd);
    const float* gm = g.Gm + ((long long)b * g.groups + c0 / g.cpg) * 2;
    const float m1 = gm[0], m2 = gm[1];
    float v[8], d[8], ga[8], be[8], sc[8], sh[8];
    load8(x + row * g.C + c0, v);
    load8(dy + row * g.C + c0, d);
    load8(g.gamma + c0, ga);
    load8(g.beta + c0, be);
    if (film != nullptr) {
      load8(film + (long long)b * g.film_ld + c0, sc);
      load8(film + (long long)b * g.film_ld + g.C + c0, sh);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = (v[j] - mean) * rstd;
      const float s1 = film != nullptr ? sc[j] + 1.0f : 1.0f, s0 = film != nullptr ? sh[j] : 0.f;
      float df = d[j];
      if (g.flags & 1) df *= silu_grad((xh * ga[j] + be[j]) * s1 + s0);
      v[j] = rstd * (df * s1 * ga[j] - m1 - xh * m2);
    }
    if (g.dx_add != nullptr) {
      float ad[8];
      load8(reinterpret_cast<const T*>(g.dx_add) + row * g.C + c0, ad);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += ad[j];
    }
    if (g.dx_add2 != nullptr) {
      float ad[8];
      load8(reinterpret_cast<const T*>(g.dx_add2) + row * g.C + c0, ad);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += ad[j];
    }
    store8(dx + row * g.C + c0, v);
  }
}

// ---------------------------------------------------------------- one launch per GroupNorm: a workgroup owns one (batch element, group).
// The training pass is a chain of ~2 800 short launches; GroupNorm was 3 of them forward (scratch reset, sums, apply) and 4 backward
// (scratch reset, per-channel sums, finish, dx) 125 times per pass.  A group is L x cpg elements (at most 48 000 on the bench shape):
// one workgroup reads it twice (the second read is an L2 hit) and needs no scratch, no atomics on the statistics and no second launch.
// Thread layout: vector column v = tid % VPG (8 channels), rows tid / VPG, tid / VPG + NT / VPG, ...  (VPG = cpg / 8, a power of two).
constexpr int GU = 4;        // rows per thread whose loads are in flight together
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {       // all threads of the NW waves get the sum; fixed order
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < NW; w += 4) t += (red[w] + red[w + 1]) + (red[w + 2] + red[w + 3]);
  return t;
}

// (batch element, group) of a workgroup of the one-launch GroupNorm kernels.  A group is cpg channels = 32 .. 64 bytes of every row; the
// groups of one batch element share the rows' cache lines.  Workgroups go to the 8 XCDs round-robin by blockIdx, so with the plain
// order b = bid / groups the 8 groups that share a line sat on 8 different XCDs and every L2 fetched every line for itself (the long
// levels' 6 MB tensors cost ~120 MB of traffic, 30 us).  With B % 8 == 0 the groups of a batch element go to ONE XCD.
__device__ __forceinline__ void gn_block(const GnDev& g, int& b, int& grp) {
  const int bid = blockIdx.x;
  if ((g.B & 7) == 0) {
    const int slot = bid >> 3;
    grp = slot % g.groups;
    b = (slot / g.groups) * 8 + (bid & 7);
  } else {
    b = bid / g.groups;
    grp = bid - b * g.groups;
  }
}

// NTB threads per workgroup: 256, or 1024 for the long groups (the 1500- and 375-position levels: 12 000 .. 48 000 elements).  There
// are only B x groups = 128 workgroups; with 4 waves each half of the chip's SIMDs had no wave at all and the others one, and the
// per-element arithmetic (SiLU and its derivative: exp, reciprocal) of 94 .. 188 elements per thread ran at one wave's issue rate:
// 30 / 58 us forward / backward for 6 - 12 MB.
template <typename T, int NTB>
__global__ __launch_bounds__(NTB) void gn_fwd_fused_kernel(const GnDev g, float* __restrict__ sums_out, int lvpg) {
  constexpr int NT = NTB;                    // (shadows the file's 256)
  __shared__ float red[NTB / 64];
  int b, grp;
  gn_block(g, b, grp);
  const long long bg = (long long)b * g.groups + grp;
  const int VPG = 1 << lvpg;
  const int v = threadIdx.x & (VPG - 1), r0 = threadIdx.x >> lvpg, RT = NT >> lvpg;
  const int c0 = grp * g.cpg + v * 8;
  const T* x = reinterpret_cast<const T*>(g.x) + (long long)b * g.L * g.C + c0;
  T* y = reinterpret_cast<T*>(g.y) + (long long)b * g.L * g.C + c0;
  // (rows in batches of GU: all of a batch's loads are issued before the first is used -- a long group is 12 .. 24 dependent memory
  // round trips otherwise, 32 / 63 us forward / backward for the 1500-position levels)
  float s = 0.f, ss = 0.f;
  for (int t = r0; t < g.L; t += GU * RT) {
    float w[GU][8];
#pragma unroll
    for (int u = 0; u < GU; ++u)
      if (t + u * RT < g.L) load8(x + (long long)(t + u * RT) * g.C, w[u]);
#pragma unroll
    for (int u = 0; u < GU; ++u)
      if (t + u * RT < g.L) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { s += w[u][j]; ss += w[u][j] * w[u][j]; }
      }
  }
  s = block_sum<NTB / 64>(s, red);
  ss = block_sum<NTB / 64>(ss, red);
  if (threadIdx.x == 0) { sums_out[2 * bg] = s; sums_out[2 * bg + 1] = ss; }
  const float mean = s * g.inv_count;
  const float rstd = 1.0f / sqrtf(fmaxf(ss * g.inv_count - mean * mean, 0.f) + g.eps);
  float ga[8], be[8], sc[8], sh[8];
  load8(g.gamma + c0, ga);
  load8(g.beta + c0, be);
  const T* film = reinterpret_cast<const T*>(g.film);
  if (film != nullptr) {
    load8(film + (long long)b * g.film_ld + c0, sc);
    load8(film + (long long)b * g.film_ld + g.C + c0, sh);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {                       // y = x A + S
    float A = rstd * ga[j], S = be[j] - mean * A;
    if (film != nullptr) { A *= sc[j] + 1.0f; S = S * (sc[j] + 1.0f) + sh[j]; }
    ga[j] = A; be[j] = S;
  }
  for (int t = r0; t < g.L; t += GU * RT) {
    float w[GU][8];
#pragma unroll
    for (int u = 0; u < GU; ++u)
      if (t + u * RT < g.L) load8(x + (long long)(t + u * RT) * g.C, w[u]);
#pragma unroll
    for (int u = 0; u < GU; ++u)
      if (t + u * RT < g.L) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float n = w[u][j] * ga[j] + be[j];
          w[u][j] = (g.flags & 1) ? silu_t<T>(n) : n;
        }
        store8(y + (long long)(t + u * RT) * g.C, w[u]);
      }
  }
}

template <typename T, int NTB>
__global__ __launch_bounds__(NTB) void gn_bwd_fused_kernel(const GnDev g, void* dx_, int lvpg) {
  constexpr int NT = NTB;                    // (shadows the file's 256)
  constexpr int GU = NTB > 256 ? 1 : 4;      // (1024 threads leave 128 registers per lane; their occupancy hides the round trips)
  extern __shared__ float lds[];                       // [NT][32] per-thread partials | [cpg][4] | [NT / 64] | [NT] chunk sums
  int b, grp;
  gn_block(g, b, grp);
  const long long bg = (long long)b * g.groups + grp;
  const int VPG = 1 << lvpg;
  const int v = threadIdx.x & (VPG - 1), r0 = threadIdx.x >> lvpg, RT = NT >> lvpg;
  const int c0 = grp * g.cpg + v * 8;
  const long long base = (long long)b * g.L * g.C + c0;
  const T* x = reinterpret_cast<const T*>(g.x) + base;
  const T* dy = reinterpret_cast<const T*>(g.dy) + base;
  T* dx = reinterpret_cast<T*>(dx_) + base;
  const float* sm = g.sums + 2 * bg;
  const float mean = sm[0] * g.inv_count;
  const float rstd = 1.0f / sqrtf(fmaxf(sm[1] * g.inv_count - mean * mean, 0.f) + g.eps);
  float ga[8], be[8], s1[8], s0[8];
  load8(g.gamma + c0, ga);
  load8(g.beta + c0, be);
  const T* film = reinterpret_cast<const T*>(g.film);
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 1.0f; s0[j] = 0.f; }
  if (film != nullptr) {
    load8(film + (long long)b * g.film_ld + c0, s1);
    load8(film + (long long)b * g.film_ld + g.C + c0, s0);
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] += 1.0f;
  }
  // P[c] = (sum dn, sum dn xhat, sum df n, sum df) over the rows
  float p[8][4];
#pragma unroll
  for (int j = 0; j < 8; ++j) { p[j][0] = p[j][1] = p[j][2] = p[j][3] = 0.f; }
  for (int t = r0; t < g.L; t += GU * RT) {
    float w[GU][8], d[GU][8];
#pragma unroll
    for (int u = 0; u < GU; ++u)
      if (t + u * RT < g.L) {
        load8(x + (long long)(t + u * RT) * g.C, w[u]);
        load8(dy + (long long)(t + u * RT) * g.C, d[u]);
      }
#pragma unroll
    for (int u = 0; u < GU; ++u)
      if (t + u * RT < g.L) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = (w[u][j] - mean) * rstd;
          const float n = xh * ga[j] + be[j];
          float df = d[u][j];
          if (g.flags & 1) df *= silu_grad_t<T>(n * s1[j] + s0[j]);
          const float dn = df * s1[j];
          p[j][0] += dn; p[j][1] += dn * xh; p[j][2] += df * n; p[j][3] += df;
        }
      }
  }
  float* part = lds + (size_t)threadIdx.x * 32;
#pragma unroll
  for (int j = 0; j < 8; ++j) *reinterpret_cast<float4*>(part + 4 * j) = make_float4(p[j][0], p[j][1], p[j][2], p[j][3]);
  __syncthreads();
  float* Pl = lds + (size_t)NT * 32;                   // [cpg][4]
  float* red = Pl + (size_t)g.cpg * 4;
  float* scr = red + NT / 64;                          // [NT / O][O] partial sums of row chunks
  const int rows_active = g.L < RT ? g.L : RT;
  const int O = g.cpg * 4;
  if (2 * O <= NT) {
    // every thread sums a chunk of the rows for one output, then O threads add the chunks: with one thread per output walking all
    // rows (128 .. 512 dependent LDS reads) this reduction WAS the kernel for the long groups (fixed order either way)
    const int chunks = NT / O, per = (rows_active + chunks - 1) / chunks;
    const int o = threadIdx.x % O, ch = threadIdx.x / O;
    if (ch < chunks) {
      const int c = o >> 2, k = o & 3, vv = c >> 3, j = c & 7;
      const int r1 = min(rows_active, (ch + 1) * per);
      float a = 0.f;
      for (int r = ch * per; r < r1; ++r) a += lds[(size_t)((r << lvpg) + vv) * 32 + 4 * j + k];
      scr[ch * O + o] = a;
    }
    __syncthreads();
    if ((int)threadIdx.x < O) {
      float a = 0.f;
      for (int c2 = 0; c2 < chunks; ++c2) a += scr[c2 * O + threadIdx.x];
      Pl[threadIdx.x] = a;
    }
  } else {
    for (int o = threadIdx.x; o < O; o += NT) {
      const int c = o >> 2, k = o & 3, vv = c >> 3, j = c & 7;
      float a = 0.f;
      for (int r = 0; r < rows_active; ++r) a += lds[(size_t)((r << lvpg) + vv) * 32 + 4 * j + k];      // fixed order
      Pl[o] = a;
    }
  }
  __syncthreads();
  float m1 = 0.f, m2 = 0.f;
  if ((int)threadIdx.x < g.cpg) {
    const int c = grp * g.cpg + threadIdx.x;
    const float4 P = *reinterpret_cast<const float4*>(Pl + 4 * threadIdx.x);
    const float gam = g.gamma[c];
    m1 = gam * P.x;
    m2 = gam * P.y;
    if (g.dfilm != nullptr) store_dfilm(g, b, c, P.z, P.w);
    atomicAdd(g.dgamma + c, P.y);                      // (the other batch elements add to the same entry)
    atomicAdd(g.dbeta + c, P.x);
  }
  m1 = block_sum<NTB / 64>(m1, red) * g.inv_count;
  m2 = block_sum<NTB / 64>(m2, red) * g.inv_count;
  for (int t = r0; t < g.L; t += GU * RT) {
    float w[GU][8], d[GU][8], ad[GU][8];
#pragma unroll
    for (int u = 0; u < GU; ++u)
      if (t + u * RT < g.L) {
        load8(x + (long long)(t + u * RT) * g.C, w[u]);
        load8(dy + (long long)(t + u * RT) * g.C, d[u]);
        if (g.dx_add != nullptr) load8(reinterpret_cast<const T*>(g.dx_add) + base + (long long)(t + u * RT) * g.C, ad[u]);
        if (g.dx_add2 != nullptr) {
          float a2[8];
          load8(reinterpret_cast<const T*>(g.dx_add2) + base + (long long)(t + u * RT) * g.C, a2);
#pragma unroll
          for (int j = 0; j < 8; ++j) ad[u][j] = (g.dx_add != nullptr ? ad[u][j] : 0.f) + a2[j];
        }
      }
#pragma unroll
    for (int u = 0; u < GU; ++u)
      if (t + u * RT < g.L) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = (w[u][j] - mean) * rstd;
          float df = d[u][j];
          if (g.flags & 1) df *= silu_grad_t<T>((xh * ga[j] + be[j]) * s1[j] + s0[j]);
          w[u][j] = rstd * (df * s1[j] * ga[j] - m1 - xh * m2);
          if (g.dx_add != nullptr || g.dx_add2 != nullptr) w[u][j] += ad[u][j];
        }
        store8(dx + (long long)(t + u * RT) * g.C, w[u]);
      }
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void act_fwd_vec_kernel(const void* x_, void* y_, long long n8, int mode) {
  const T* x = reinterpret_cast<const T*>(x_);
  T* y = reinterpret_cast<T*>(y_);
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n8; i += (long long)gridDim.x * NT) {
    float v[8];
    load8(x + 8 * i, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = mode == 0 ? gelu_erf(v[j]) : mode == 1 ? silu_precise(v[j]) : (v[j] > 0.f ? v[j] : expm1f(v[j]));
    store8(y + 8 * i, v);
  }
}
template <typename T>
__global__ __launch_bounds__(NT) void act_bwd_vec_kernel(const void* dy_, const void* x_, void* dx_, long long n8, int mode) {
  const T* dy = reinterpret_cast<const T*>(dy_);
  const T* x = reinterpret_cast<const T*>(x_);
  T* dx = reinterpret_cast<T*>(dx_);
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n8; i += (long long)gridDim.x * NT) {
    float v[8], d[8];
    load8(x + 8 * i, v);
    load8(dy + 8 * i, d);
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] *= mode == 0 ? gelu_grad(v[j]) : mode == 1 ? silu_grad(v[j]) : (v[j] > 0.f ? 1.f : expf(v[j]));
    store8(dx + 8 * i, d);
  }
}

__host__ bool vec_ok(const void* p0, const void* p1, const void* p2, int C, int ld, int cpg) {
  return ld == C && (C & 7) == 0 && (cpg & 7) == 0 && (((uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)p2) & 15) == 0;
}

void red_geom(int C, int L, int& CT, int& rows_per_block, int& gx, int& gy) {
  CT = 256;
  while (CT > 1 && CT / 2 >= C) CT /= 2;      // smallest power of two >= C, capped at 256
  const int RT = NT / CT;
  rows_per_block = RT * 16;                    // each thread walks <= 16 rows
  if (rows_per_block > L) rows_per_block = (L + RT - 1) / RT * RT;
  gx = (L + rows_per_block - 1) / rows_per_block;
  gy = (C + CT - 1) / CT;
}

int ew_grid(long long total) {
  long long g = (total + NT - 1) / NT;
  if (g > 256 * 32) g = 256 * 32;
  return (int)(g < 1 ? 1 : g);
}

// ---------------------------------------------------------------- LayerNorm (one wave per row, C <= 64 * 32)
constexpr int LN_MAXPL = 32;

template <typename T>
__global__ __launch_bounds__(NT) void ln_fwd_kernel(const void* x_, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     void* y_, float* __restrict__ stats, int rows, int C, int ld, float eps) {
  const T* x = reinterpret_cast<const T*>(x_);
  T* y = reinterpret_cast<T*>(y_);
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xr = x + (long long)row * ld;
  float v[LN_MAXPL];
  float s = 0.f;
  int n = 0;
  for (int c = lane; c < C; c += 64, ++n) { v[n] = (float)xr[c]; s += v[n]; }
  s = wave_sum(s);
  const float mean = s / C;
  float q = 0.f;
  for (int i = 0; i < n; ++i) { const float d = v[i] - mean; q += d * d; }
  q = wave_sum(q);
  const float rstd = 1.0f / sqrtf(q / C + eps);
  T* yr = y + (long long)row * ld;
  n = 0;
  for (int c = lane; c < C; c += 64, ++n) yr[c] = (T)((v[n] - mean) * rstd * gamma[c] + beta[c]);
  if (lane == 0) { stats[2 * (long long)row] = mean; stats[2 * (long long)row + 1] = rstd; }
}

// the same with 8 channels (16 bytes in bf16) per lane and access: rows of a multiple of 8 channels, C <= 2048
template <typename T>
__global__ __launch_bounds__(NT) void ln_fwd_vec_kernel(const void* x_, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         void* y_, float* __restrict__ stats, int rows, int C, int ld, float eps) {
  const T* x = reinterpret_cast<const T*>(x_);
  T* y = reinterpret_cast<T*>(y_);
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nvec = C >> 3;
  const T* xr = x + (long long)row * ld;
  float v[4][8];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int vi = lane + 64 * j;
    if (vi < nvec) {
      load8(xr + vi * 8, v[j]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[j][e];
    }
  }
  s = wave_sum(s);
  const float mean = s / C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (lane + 64 * j < nvec) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[j][e] - mean; q += d * d; }
    }
  }
  q = wave_sum(q);
  const float rstd = 1.0f / sqrtf(q / C + eps);
  T* yr = y + (long long)row * ld;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int vi = lane + 64 * j;
    if (vi < nvec) {
      float g8[8], b8[8];
      load8(gamma + vi * 8, g8);
      load8(beta + vi * 8, b8);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[j][e] = (v[j][e] - mean) * rstd * g8[e] + b8[e];
      store8(yr + vi * 8, v[j]);
    }
  }
  if (lane == 0) { stats[2 * (long long)row] = mean; stats[2 * (long long)row + 1] = rstd; }
}

// each wave walks rows row0, row0 + stride, ... and keeps the column sums of its lanes in registers
template <typename T, int NTB>
__global__ __launch_bounds__(NTB) void ln_bwd_kernel(const void* dy_, const void* x_, const float* __restrict__ stats,
                                                      const float* __restrict__ gamma, void* dx_, const void* add_,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int C, int ld) {
  constexpr int NT = NTB;                    // (shadows the file's 256: a block of NTB / 64 waves)
  const T* dy = reinterpret_cast<const T*>(dy_);
  const T* x = reinterpret_cast<const T*>(x_);
  T* dx = reinterpret_cast<T*>(dx_);
  const int lane = threadIdx.x & 63;
  const int wid = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  const int nw = gridDim.x * (NT / 64);
  float ag[LN_MAXPL], ab[LN_MAXPL], gv[LN_MAXPL];
  int npl = 0;
  for (int c = lane; c < C; c += 64, ++npl) { ag[npl] = 0.f; ab[npl] = 0.f; gv[npl] = gamma[c]; }
  for (int row = wid; row < rows; row += nw) {
    const float mean = stats[2 * (long long)row], rstd = stats[2 * (long long)row + 1];
    const T* xr = x + (long long)row * ld;
    const T* dr = dy + (long long)row * ld;
    float xh[LN_MAXPL], dh[LN_MAXPL];
    float s1 = 0.f, s2 = 0.f;
    int n = 0;
    for (int c = lane; c < C; c += 64, ++n) {
      const float d = (float)dr[c];
      xh[n] = ((float)xr[c] - mean) * rstd;
      dh[n] = d * gv[n];
      s1 += dh[n];
      s2 += dh[n] * xh[n];
      ag[n] += d * xh[n];
      ab[n] += d;
    }
    if (dx_ == nullptr) continue;
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    s1 /= C; s2 /= C;
    T* xo = dx + (long long)row * ld;
    n = 0;
    for (int c = lane; c < C; c += 64, ++n) {
      float o = rstd * (dh[n] - s1 - xh[n] * s2);
      if (add_ != nullptr) o += (float)reinterpret_cast<const T*>(add_)[(long long)row * ld + c];
      xo[c] = (T)o;
    }
  }
  // the four waves of the block meet in LDS, then ONE atomic per column and block (512 waves adding 2 C columns each was the
  // kernel's time: 26 us at 2 064 rows x 1 024 columns)
  extern __shared__ float ln_col[];            // [NT / 64 - 1][2][C]
  const int w = threadIdx.x >> 6;
  int n = 0;
  if (w > 0) {
    for (int c = lane; c < C; c += 64, ++n) { ln_col[((w - 1) * 2) * C + c] = ag[n]; ln_col[((w - 1) * 2 + 1) * C + c] = ab[n]; }
  }
  __syncthreads();
  if (w == 0) {
    n = 0;
    for (int c = lane; c < C; c += 64, ++n) {
      float a = ag[n], b2 = ab[n];
      for (int k = 0; k < NT / 64 - 1; ++k) { a += ln_col[(k * 2) * C + c]; b2 += ln_col[(k * 2 + 1) * C + c]; }
      atomicAdd(dgamma + c, a);
      atomicAdd(dbeta + c, b2);
    }
  }
}

// 8 channels per lane and access (rows of a multiple of 8 channels): the scalar form above moves two bytes per lane and load -- 43 us for
// the 2 064 x 1 024 rows of the text context, ten times its traffic at HBM speed
constexpr int LN_MAXV = 4;        // 8-channel vectors per lane: C <= 2048
template <typename T, int NTB>
__global__ __launch_bounds__(NTB) void ln_bwd_vec_kernel(const void* dy_, const void* x_, const float* __restrict__ stats,
                                                          const float* __restrict__ gamma, void* dx_, const void* add_,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int C, int ld) {
  extern __shared__ float ln_col[];            // [NTB / 64 - 1][2][C]
  const T* dy = reinterpret_cast<const T*>(dy_);
  const T* x = reinterpret_cast<const T*>(x_);
  T* dx = reinterpret_cast<T*>(dx_);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wid = blockIdx.x * (NTB / 64) + w, nw = gridDim.x * (NTB / 64);
  const int nvec = C >> 3;
  float ag[LN_MAXV][8], ab[LN_MAXV][8], gv[LN_MAXV][8];
#pragma unroll
  for (int j = 0; j < LN_MAXV; ++j) {
    const int v = lane + 64 * j;
#pragma unroll
    for (int e = 0; e < 8; ++e) { ag[j][e] = 0.f; ab[j][e] = 0.f; gv[j][e] = 0.f; }
    if (v < nvec) load8(gamma + v * 8, gv[j]);
  }
  const float inv_C = 1.0f / (float)C;
  for (int row = wid; row < rows; row += nw) {
    const float mean = stats[2 * (long long)row], rstd = stats[2 * (long long)row + 1];
    const T* xr = x + (long long)row * ld;
    const T* dr = dy + (long long)row * ld;
    float xh[LN_MAXV][8], dh[LN_MAXV][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
      const int v = lane + 64 * j;
      if (v < nvec) {
        float d[8];
        load8(xr + v * 8, xh[j]);
        load8(dr + v * 8, d);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xh[j][e] = (xh[j][e] - mean) * rstd;
          dh[j][e] = d[e] * gv[j][e];
          s1 += dh[j][e];
          s2 += dh[j][e] * xh[j][e];
          ag[j][e] += d[e] * xh[j][e];
          ab[j][e] += d[e];
        }
      }
    }
    if (dx_ == nullptr) continue;            // (the input needs no gradient -- the text context: only the column sums are wanted)
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    s1 *= inv_C; s2 *= inv_C;
    T* xo = dx + (long long)row * ld;
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
      const int v = lane + 64 * j;
      if (v < nvec) {
        float o8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] = rstd * (dh[j][e] - s1 - xh[j][e] * s2);
        if (add_ != nullptr) {
          float ad[8];
          load8(reinterpret_cast<const T*>(add_) + (long long)row * ld + v * 8, ad);
#pragma unroll
          for (int e = 0; e < 8; ++e) o8[e] += ad[e];
        }
        store8(xo + v * 8, o8);
      }
    }
  }
  if (w > 0) {
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
      const int v = lane + 64 * j;
      if (v < nvec) {
        store8(ln_col + ((w - 1) * 2) * C + v * 8, ag[j]);
        store8(ln_col + ((w - 1) * 2 + 1) * C + v * 8, ab[j]);
      }
    }
  }
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int j = 0; j < LN_MAXV; ++j) {
      const int v = lane + 64 * j;
      if (v < nvec) {
        for (int k = 0; k < NTB / 64 - 1; ++k) {
          float a8[8], b8[8];
          load8(ln_col + (k * 2) * C + v * 8, a8);
          load8(ln_col + (k * 2 + 1) * C + v * 8, b8);
#pragma unroll
          for (int e = 0; e < 8; ++e) { ag[j][e] += a8[e]; ab[j][e] += b8[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { atomicAdd(dgamma + v * 8 + e, ag[j][e]); atomicAdd(dbeta + v * 8 + e, ab[j][e]); }
      }
    }
  }
}

// ---- two LayerNorms of ONE input in one launch: self-attention normalises x twice, ``norm`` for to_q and ``norm_context`` for to_kv
// (blocks.py:427-429 with context = x): the statistics are shared, only (gamma, beta) differ.  Backward: both branches are linear in
// the standardised row, so dx is ONE LayerNorm backward of the upstream sum g = dy1 gamma1 + dy2 gamma2; four column sums.
constexpr int LN2_MAXV = 2;       // 8-channel vectors per lane: C <= 1024
template <typename T>
__global__ __launch_bounds__(NT) void ln2_fwd_vec_kernel(const void* x_, const float* __restrict__ g1, const float* __restrict__ b1,
                                                          const float* __restrict__ g2, const float* __restrict__ b2, void* y1_, void* y2_,
                                                          float* __restrict__ stats, int rows, int C, int ld, float eps) {
  const T* x = reinterpret_cast<const T*>(x_);
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nvec = C >> 3;
  const T* xr = x + (long long)row * ld;
  float v[LN2_MAXV][8];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < LN2_MAXV; ++j) {
    const int vi = lane + 64 * j;
    if (vi < nvec) {
      load8(xr + vi * 8, v[j]);
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[j][e];
    }
  }
  s = wave_sum(s);
  const float mean = s / C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < LN2_MAXV; ++j) {
    if (lane + 64 * j < nvec) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[j][e] - mean; q += d * d; }
    }
  }
  q = wave_sum(q);
  const float rstd = 1.0f / sqrtf(q / C + eps);
  T* y1 = reinterpret_cast<T*>(y1_) + (long long)row * ld;
  T* y2 = reinterpret_cast<T*>(y2_) + (long long)row * ld;
#pragma unroll
  for (int j = 0; j < LN2_MAXV; ++j) {
    const int vi = lane + 64 * j;
    if (vi < nvec) {
      float ga[8], be[8], o[8];
      load8(g1 + vi * 8, ga);
      load8(b1 + vi * 8, be);
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[j][e] = (v[j][e] - mean) * rstd; o[e] = v[j][e] * ga[e] + be[e]; }
      store8(y1 + vi * 8, o);
      load8(g2 + vi * 8, ga);
      load8(b2 + vi * 8, be);
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = v[j][e] * ga[e] + be[e];
      store8(y2 + vi * 8, o);
    }
  }
  if (lane == 0) { stats[2 * (long long)row] = mean; stats[2 * (long long)row + 1] = rstd; }
}

template <typename T, int NTB>
__global__ __launch_bounds__(NTB) void ln2_bwd_vec_kernel(const void* dy1_, const void* dy2_, const void* x_, const float* __restrict__ stats,
                                                           const float* __restrict__ gamma1, const float* __restrict__ gamma2, void* dx_,
                                                           const void* add_, float* __restrict__ dgamma1, float* __restrict__ dbeta1,
                                                           float* __restrict__ dgamma2, float* __restrict__ dbeta2, int rows, int C, int ld) {
  extern __shared__ float ln_col[];            // [NTB / 64 - 1][4][C]
  const T* dy1 = reinterpret_cast<const T*>(dy1_);
  const T* dy2 = reinterpret_cast<const T*>(dy2_);
  const T* x = reinterpret_cast<const T*>(x_);
  T* dx = reinterpret_cast<T*>(dx_);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wid = blockIdx.x * (NTB / 64) + w, nw = gridDim.x * (NTB / 64);
  const int nvec = C >> 3;
  float ac[4][LN2_MAXV][8], gv1[LN2_MAXV][8], gv2[LN2_MAXV][8];      // column sums: dgamma1, dbeta1, dgamma2, dbeta2
#pragma unroll
  for (int j = 0; j < LN2_MAXV; ++j) {
    const int v = lane + 64 * j;
#pragma unroll
    for (int e = 0; e < 8; ++e) { ac[0][j][e] = ac[1][j][e] = ac[2][j][e] = ac[3][j][e] = 0.f; gv1[j][e] = gv2[j][e] = 0.f; }
    if (v < nvec) { load8(gamma1 + v * 8, gv1[j]); load8(gamma2 + v * 8, gv2[j]); }
  }
  const float inv_C = 1.0f / (float)C;
  for (int row = wid; row < rows; row += nw) {
    const float mean = stats[2 * (long long)row], rstd = stats[2 * (long long)row + 1];
    const long long ro = (long long)row * ld;
    float xh[LN2_MAXV][8], dh[LN2_MAXV][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < LN2_MAXV; ++j) {
      const int v = lane + 64 * j;
      if (v < nvec) {
        float d1[8], d2[8];
        load8(x + ro + v * 8, xh[j]);
        load8(dy1 + ro + v * 8, d1);
        load8(dy2 + ro + v * 8, d2);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xh[j][e] = (xh[j][e] - mean) * rstd;
          dh[j][e] = d1[e] * gv1[j][e] + d2[e] * gv2[j][e];
          s1 += dh[j][e];
          s2 += dh[j][e] * xh[j][e];
          ac[0][j][e] += d1[e] * xh[j][e];
          ac[1][j][e] += d1[e];
          ac[2][j][e] += d2[e] * xh[j][e];
          ac[3][j][e] += d2[e];
        }
      }
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    s1 *= inv_C; s2 *= inv_C;
#pragma unroll
    for (int j = 0; j < LN2_MAXV; ++j) {
      const int v = lane + 64 * j;
      if (v < nvec) {
        float o8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o8[e] = rstd * (dh[j][e] - s1 - xh[j][e] * s2);
        if (add_ != nullptr) {
          float ad[8];
          load8(reinterpret_cast<const T*>(add_) + ro + v * 8, ad);
#pragma unroll
          for (int e = 0; e < 8; ++e) o8[e] += ad[e];
        }
        store8(dx + ro + v * 8, o8);
      }
    }
  }
  if (w > 0) {
#pragma unroll
    for (int j = 0; j < LN2_MAXV; ++j) {
      const int v = lane + 64 * j;
      if (v < nvec) {
#pragma unroll
        for (int k = 0; k < 4; ++k) store8(ln_col + ((w - 1) * 4 + k) * C + v * 8, ac[k][j]);
      }
    }
  }
  __syncthreads();
  if (w == 0) {
    float* outs[4] = {dgamma1, dbeta1, dgamma2, dbeta2};
#pragma unroll
    for (int j = 0; j < LN2_MAXV; ++j) {
      const int v = lane + 64 * j;
      if (v < nvec) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          for (int ww = 0; ww < NTB / 64 - 1; ++ww) {
            float a8[8];
            load8(ln_col + (ww * 4 + k) * C + v * 8, a8);
#pragma unroll
            for (int e = 0; e < 8; ++e) ac[k][j][e] += a8[e];
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) atomicAdd(outs[k] + v * 8 + e, ac[k][j][e]);
        }
      }
    }
  }
}

// ---------------------------------------------------------------- pointwise
template <typename T>
__global__ __launch_bounds__(NT) void act_fwd_kernel(const void* x_, void* y_, long long n, int mode) {
  const T* x = reinterpret_cast<const T*>(x_);
  T* y = reinterpret_cast<T*>(y_);
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
    const float v = (float)x[i];
    y[i] = (T)(mode == 0 ? gelu_erf(v) : mode == 1 ? silu_precise(v) : (v > 0.f ? v : expm1f(v)));
  }
}
template <typename T>
__global__ __launch_bounds__(NT) void act_bwd_kernel(const void* dy_, const void* x_, void* dx_, long long n, int mode) {
  const T* dy = reinterpret_cast<const T*>(dy_);
  const T* x = reinterpret_cast<const T*>(x_);
  T* dx = reinterpret_cast<T*>(dx_);
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n; i += (long long)gridDim.x * NT) {
    const float v = (float)x[i];
    dx[i] = (T)((float)dy[i] * (mode == 0 ? gelu_grad(v) : mode == 1 ? silu_grad(v) : (v > 0.f ? 1.f : expf(v))));
  }
}

// ---------------------------------------------------------------- softmax (one wave per row)
template <typename T>
__global__ __launch_bounds__(NT) void softmax_fwd_kernel(const float* __restrict__ s, void* p_, int rows, int Nq, int Nk,
                                                          int ld_s, int ld_p, int causal) {
  T* p = reinterpret_cast<T*>(p_);
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int i = row % Nq;
  const int lim = causal ? min(Nk, i + (Nk - Nq) + 1) : Nk;     // keys j < lim are kept (blocks.py:315-319)
  const float* sr = s + (long long)row * ld_s;
  float m = -3.0e38f;
  for (int j = lane; j < lim; j += 64) m = fmaxf(m, sr[j]);
  m = wave_max(m);
  float z = 0.f;
  for (int j = lane; j < lim; j += 64) z += expf(sr[j] - m);
  z = wave_sum(z);
  const float inv = 1.0f / z;
  T* pr = p + (long long)row * ld_p;
  for (int j = lane; j < ld_p; j += 64) pr[j] = (T)(j < lim ? expf(sr[j] - m) * inv : 0.f);
}
template <typename T>
__global__ __launch_bounds__(NT) void softmax_bwd_kernel(const void* p_, const float* __restrict__ dp, void* ds_, int rows,
                                                          int Nk, int ld_s, int ld_p) {
  const T* p = reinterpret_cast<const T*>(p_);
  T* ds = reinterpret_cast<T*>(ds_);
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* pr = p + (long long)row * ld_p;
  const float* dr = dp + (long long)row * ld_s;
  float d = 0.f;
  for (int j = lane; j < Nk; j += 64) d += (float)pr[j] * dr[j];
  d = wave_sum(d);
  T* so = ds + (long long)row * ld_p;
  for (int j = lane; j < ld_p; j += 64) so[j] = (T)(j < Nk ? (float)pr[j] * (dr[j] - d) : 0.f);
}

// ---------------------------------------------------------------- column sums
template <typename T>
__global__ __launch_bounds__(NT) void colsum_kernel(const void* x_, float* __restrict__ out, int rows, int C, int ld, int CT,
                                                     int rows_per_block) {
  const T* x = reinterpret_cast<const T*>(x_);
  const int c = blockIdx.y * CT + threadIdx.x % CT;
  const int ty = threadIdx.x / CT, RT = NT / CT;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  if (c >= C) return;
  float s = 0.f;
  for (int r = r0 + ty; r < r1; r += RT) s += (float)x[(long long)r * ld + c];
  atomicAdd(out + c, s);
}

// the same for many rows of 8-channel vectors (the bias gradient of a long level: 24 000 rows x 128): a block walks a long slice of the
// rows with eight 16-byte loads in flight per thread and adds its sums ONCE -- colsum_kernel's 750 blocks of 32 rows put 96 000
// float atomics on four lines (52 us)
template <typename T>
__global__ __launch_bounds__(256) void colsum_rows_kernel(const void* x_, float* __restrict__ out, int rows, int C, int ld, int rows_per_block) {
  __shared__ float part[16][128];
  const T* x = reinterpret_cast<const T*>(x_);
  const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  for (int cb = 0; cb < C; cb += 128) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int c = cb + cg * 8;
    if (c < C) {
      for (int r = r0 + rl; r < r1; r += 16 * 8) {
        float v[8][8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int ru = r + 16 * u;
          load8(x + (long long)(ru < r1 ? ru : r0) * ld + c, v[u]);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (r + 16 * u < r1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += v[u][e];
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) part[rl][cg * 8 + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 128 && cb + (int)threadIdx.x < C) {
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < 16; ++q) t += part[q][threadIdx.x];
      atomicAdd(out + cb + threadIdx.x, t);
    }
  }
}

// dst = (T) src, src = 0: hands a float32 split-K accumulator over in the compute dtype and leaves it zero for the
// next GEMM of the stream (the accumulator is one persistent scratch buffer, so no fill launch is ever needed)
template <typename T>
__global__ __launch_bounds__(NT) void convert_clear_kernel(float* __restrict__ src, void* dst_, long long n4) {
  T* dst = reinterpret_cast<T*>(dst_);
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long long)gridDim.x * NT) {
    const float4 v = reinterpret_cast<const float4*>(src)[i];
    reinterpret_cast<float4*>(src)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float o[4] = {v.x, v.y, v.z, v.w};
    store4(dst + 4 * i, o);
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void convert_clear_add_kernel(float* __restrict__ src, void* dst_, const void* res_, long long n4) {
  T* dst = reinterpret_cast<T*>(dst_);
  const T* res = reinterpret_cast<const T*>(res_);
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long long)gridDim.x * NT) {
    const float4 v = reinterpret_cast<const float4*>(src)[i];
    reinterpret_cast<float4*>(src)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    float r[4];
    load4(res + 4 * i, r);
    const float o[4] = {v.x + r[0], v.y + r[1], v.z + r[2], v.w + r[3]};
    store4(dst + 4 * i, o);
  }
}

#define DISPATCH(dtype, KERNEL, grid, ...)                                                        \
  do {                                                                                            \
    if ((dtype) == JEN1_F32) hipLaunchKernelGGL(KERNEL<float>, grid, dim3(NT), 0, s, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL<bf16_t>, grid, dim3(NT), 0, s, __VA_ARGS__);                    \
    JEN1_HIP(hipGetLastError());                                                                  \
  } while (0)

int check_dtype(int dtype, const char* who) {
  JEN1_CHECK(dtype == JEN1_F32 || dtype == JEN1_BF16, "%s: dtype must be JEN1_F32 or JEN1_BF16", who);
  return 0;
}

int gn_fill(GnDev& g, const char* who, const void* x, const float* sums, const float* gamma, const float* beta, const void* film,
            int film_ld, int B, int L, int C, int ld, int groups, float eps, int flags) {
  JEN1_CHECK(x && sums && gamma && beta, "%s: NULL argument", who);
  JEN1_CHECK(B >= 1 && L >= 1 && C >= 1 && ld >= C, "%s: bad shape B=%d L=%d C=%d ld=%d", who, B, L, C, ld);
  JEN1_CHECK(groups >= 1 && C % groups == 0, "%s: num_channels must be divisible by num_groups", who);   // torch GroupNorm's check
  JEN1_CHECK(film == nullptr || film_ld >= 2 * C, "%s: film_ld must be >= 2 C", who);
  memset(&g, 0, sizeof(g));
  g.x = x; g.sums = sums; g.gamma = gamma; g.beta = beta; g.film = film; g.film_ld = film_ld;
  g.B = B; g.L = L; g.C = C; g.ld = ld; g.groups = groups; g.cpg = C / groups; g.eps = eps; g.flags = flags;
  g.inv_count = 1.0f / ((float)(C / groups) * (float)L);
  return 0;
}

}  // namespace

extern "C" int jen1_gn_sums(const void* x, float* sums, int B, int L, int C, int ld, int groups, int dtype, void* stream) {
  if (check_dtype(dtype, "jen1_gn_sums")) return 1;
  JEN1_CHECK(x && sums, "jen1_gn_sums: NULL argument");
  JEN1_CHECK(B >= 1 && L >= 1 && C >= 1 && ld >= C, "jen1_gn_sums: bad shape B=%d L=%d C=%d ld=%d", B, L, C, ld);
  JEN1_CHECK(groups >= 1 && C % groups == 0, "jen1_gn_sums: num_channels must be divisible by num_groups");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (zero2(sums, 2 * B * groups, nullptr, 0, s)) return 1;
  int CT, rpb, gx, gy;
  if (vec_ok(x, nullptr, nullptr, C, ld, C / groups)) {
    red_geom(C / 8, L, CT, rpb, gx, gy);
    DISPATCH(dtype, gn_sums_vec_kernel, dim3(gx, gy, B), x, sums, L, C, groups, C / groups, CT, rpb);
    return 0;
  }
  red_geom(C, L, CT, rpb, gx, gy);
  DISPATCH(dtype, gn_sums_kernel, dim3(gx, gy, B), x, sums, L, C, ld, groups, C / groups, CT, rpb);
  return 0;
}

namespace {
// the one-launch form: vector-aligned rows, a power-of-two number of 8-channel vectors per group, enough (batch element, group) pairs
// to fill the chip reasonably and a group small enough for one workgroup
constexpr long long GN_LONG_GROUP = 12000;       // elements per (batch element, group) from which the 1024-thread form runs
constexpr size_t lds_cap = 160 * 1024;

bool gn_fused_ok(const void* x, const void* y, const void* dy, const float* gamma, const float* beta, const void* film, int film_ld, int B, int L,
                 int C, int ld, int groups, int& lvpg) {
  const int cpg = C / groups;
  if (!vec_ok(x, y, dy, C, ld, cpg) || ((((uintptr_t)gamma | (uintptr_t)beta) & 15) != 0)) return false;
  if (film != nullptr && ((film_ld & 7) != 0 || ((uintptr_t)film & 15) != 0)) return false;
  const int vpg = cpg / 8;
  if ((vpg & (vpg - 1)) != 0 || vpg > NT) return false;
  lvpg = 0;
  while ((1 << lvpg) < vpg) ++lvpg;
  if (getenv("JEN1_GN_FUSED") && atoi(getenv("JEN1_GN_FUSED")) == 0) return false;
  return B * groups >= 32 && (long long)L * cpg <= (1 << 17);
}
}  // namespace

extern "C" int jen1_gn_forward(const void* x, float* sums, const float* gamma, const float* beta, const void* film, int film_ld, void* y, int B,
                               int L, int C, int ld, int groups, float eps, int flags, int dtype, void* stream) {
  if (check_dtype(dtype, "jen1_gn_forward")) return 1;
  GnDev g;
  if (gn_fill(g, "jen1_gn_forward", x, sums, gamma, beta, film, film_ld, B, L, C, ld, groups, eps, flags)) return 1;
  JEN1_CHECK(y != nullptr, "jen1_gn_forward: y is NULL");
  int lvpg = 0;
  if (!gn_fused_ok(x, y, nullptr, gamma, beta, film, film_ld, B, L, C, ld, groups, lvpg)) {
    if (jen1_gn_sums(x, sums, B, L, C, ld, groups, dtype, stream)) return 1;
    return jen1_gn_apply(x, sums, gamma, beta, film, film_ld, y, B, L, C, ld, groups, eps, flags, dtype, stream);
  }
  g.y = y;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if ((long long)L * (C / groups) >= GN_LONG_GROUP) {
    if (dtype == JEN1_F32) hipLaunchKernelGGL((gn_fwd_fused_kernel<float, 1024>), dim3(B * groups), dim3(1024), 0, s, g, sums, lvpg);
    else hipLaunchKernelGGL((gn_fwd_fused_kernel<bf16_t, 1024>), dim3(B * groups), dim3(1024), 0, s, g, sums, lvpg);
  } else {
    if (dtype == JEN1_F32) hipLaunchKernelGGL((gn_fwd_fused_kernel<float, 256>), dim3(B * groups), dim3(256), 0, s, g, sums, lvpg);
    else hipLaunchKernelGGL((gn_fwd_fused_kernel<bf16_t, 256>), dim3(B * groups), dim3(256), 0, s, g, sums, lvpg);
  }
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_gn_apply(const void* x, const float* sums, const float* gamma, const float* beta, const void* film, int film_ld,
                             void* y, int B, int L, int C, int ld, int groups, float eps, int flags, int dtype, void* stream) {
  if (check_dtype(dtype, "jen1_gn_apply")) return 1;
  GnDev g;
  if (gn_fill(g, "jen1_gn_apply", x, sums, gamma, beta, film, film_ld, B, L, C, ld, groups, eps, flags)) return 1;
  JEN1_CHECK(y != nullptr, "jen1_gn_apply: y is NULL");
  g.y = y;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool film_ok = film == nullptr || ((film_ld & 7) == 0 && ((uintptr_t)film & 15) == 0);
  if (film_ok && vec_ok(x, y, nullptr, C, ld, C / groups) && (((uintptr_t)gamma | (uintptr_t)beta) & 15) == 0) {
    DISPATCH(dtype, gn_apply_vec_kernel, dim3(ew_grid((long long)B * L * C / 8)), g);
    return 0;
  }
  DISPATCH(dtype, gn_apply_kernel, dim3(ew_grid((long long)B * L * C)), g);
  return 0;
}

extern "C" int jen1_gn_backward(const void* dy, const void* x, const float* sums, const float* gamma, const float* beta,
                                const void* film, int film_ld, void* dx, float* dgamma, float* dbeta, void* dfilm, float* P,
                                float* Gm, int B, int L, int C, int ld, int groups, float eps, int flags, int dtype, void* stream) {
  return jen1_gn_backward_add(dy, x, sums, gamma, beta, film, film_ld, dx, nullptr, dgamma, dbeta, dfilm, P, Gm, B, L, C, ld, groups, eps,
                              flags, dtype, stream);
}

extern "C" int jen1_gn_backward_add(const void* dy, const void* x, const float* sums, const float* gamma, const float* beta,
                                    const void* film, int film_ld, void* dx, const void* dx_add, float* dgamma, float* dbeta,
                                    void* dfilm, float* P, float* Gm, int B, int L, int C, int ld, int groups, float eps, int flags,
                                    int dtype, void* stream) {
  return jen1_gn_backward_add2(dy, x, sums, gamma, beta, film, film_ld, dx, dx_add, nullptr, dgamma, dbeta, dfilm, P, Gm, B, L, C, ld, groups, eps,
                               flags, dtype, stream);
}

extern "C" int jen1_gn_backward_add2(const void* dy, const void* x, const float* sums, const float* gamma, const float* beta,
                                     const void* film, int film_ld, void* dx, const void* dx_add, const void* dx_add2, float* dgamma,
                                     float* dbeta, void* dfilm, float* P, float* Gm, int B, int L, int C, int ld, int groups, float eps,
                                     int flags, int dtype, void* stream) {
  if (check_dtype(dtype, "jen1_gn_backward")) return 1;
  JEN1_CHECK((((uintptr_t)dx_add | (uintptr_t)dx_add2) & 15) == 0, "jen1_gn_backward_add: dx_add must start on a 16-byte boundary");
  GnDev g;
  if (gn_fill(g, "jen1_gn_backward", x, sums, gamma, beta, film, film_ld, B, L, C, ld, groups, eps, flags)) return 1;
  JEN1_CHECK(dy && dx && dgamma && dbeta && P && Gm, "jen1_gn_backward: NULL argument");
  JEN1_CHECK((film == nullptr) == (dfilm == nullptr), "jen1_gn_backward: dfilm must be given exactly when film is");
  g.dy = dy; g.P = P; g.Gm = Gm; g.dgamma = dgamma; g.dbeta = dbeta; g.dfilm = reinterpret_cast<float*>(dfilm);
  g.dx_add = dx_add;
  g.dx_add2 = dx_add2;
  g.film_bf16 = dtype == JEN1_BF16 ? 1 : 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int lvpg = 0;
  if (gn_fused_ok(x, dx, dy, gamma, beta, film, film_ld, B, L, C, ld, groups, lvpg)) {
    if ((long long)L * (C / groups) >= GN_LONG_GROUP) {
      const size_t lds = ((size_t)1024 * 32 + (size_t)(C / groups) * 4 + 16 + 1024) * sizeof(float);
      if (dtype == JEN1_F32) {
        JEN1_MAX_LDS_ONCE((gn_bwd_fused_kernel<float, 1024>), (int)lds_cap);
        hipLaunchKernelGGL((gn_bwd_fused_kernel<float, 1024>), dim3(B * groups), dim3(1024), lds, s, g, dx, lvpg);
      } else {
        JEN1_MAX_LDS_ONCE((gn_bwd_fused_kernel<bf16_t, 1024>), (int)lds_cap);
        hipLaunchKernelGGL((gn_bwd_fused_kernel<bf16_t, 1024>), dim3(B * groups), dim3(1024), lds, s, g, dx, lvpg);
      }
    } else {
      const size_t lds = ((size_t)256 * 32 + (size_t)(C / groups) * 4 + 4 + 256) * sizeof(float);
      if (dtype == JEN1_F32) hipLaunchKernelGGL((gn_bwd_fused_kernel<float, 256>), dim3(B * groups), dim3(256), lds, s, g, dx, lvpg);
      else hipLaunchKernelGGL((gn_bwd_fused_kernel<bf16_t, 256>), dim3(B * groups), dim3(256), lds, s, g, dx, lvpg);
    }
    JEN1_HIP(hipGetLastError());
    return 0;
  }
  if (zero2(P, 4 * B * C, nullptr, 0, s)) return 1;
  int CT, rpb, gx, gy;
  red_geom(C, L, CT, rpb, gx, gy);
  DISPATCH(dtype, gn_bwd_sums_kernel, dim3(gx, gy, B), g, CT, rpb);
  hipLaunchKernelGGL(gn_bwd_finish_kernel, dim3(B * groups + (C + 63) / 64), dim3(64), 0, s, g);
  JEN1_HIP(hipGetLastError());
  const bool film_ok = film == nullptr || ((film_ld & 7) == 0 && ((uintptr_t)film & 15) == 0);
  if (film_ok && vec_ok(x, dy, dx, C, ld, C / groups) && (((uintptr_t)gamma | (uintptr_t)beta) & 15) == 0) {
    DISPATCH(dtype, gn_bwd_dx_vec_kernel, dim3(ew_grid((long long)B * L * C / 8)), g, dx);
    return 0;
  }
  DISPATCH(dtype, gn_bwd_dx_kernel, dim3(ew_grid((long long)B * L * C)), g, dx);
  return 0;
}

extern "C" int jen1_ln_forward(const void* x, const float* gamma, const float* beta, void* y, float* stats, int rows, int C, int ld,
                               float eps, int dtype, void* stream) {
  if (check_dtype(dtype, "jen1_ln_forward")) return 1;
  JEN1_CHECK(x && gamma && beta && y && stats, "jen1_ln_forward: NULL argument");
  JEN1_CHECK(rows >= 1 && C >= 1 && ld >= C && C <= 64 * LN_MAXPL, "jen1_ln_forward: bad shape rows=%d C=%d ld=%d (C <= %d)", rows, C, ld, 64 * LN_MAXPL);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if ((C & 7) == 0 && (ld & 7) == 0 && C <= 2048 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0) {
    DISPATCH(dtype, ln_fwd_vec_kernel, dim3((rows + 3) / 4), x, gamma, beta, y, stats, rows, C, ld, eps);
    return 0;
  }
  DISPATCH(dtype, ln_fwd_kernel, dim3((rows + 3) / 4), x, gamma, beta, y, stats, rows, C, ld, eps);
  return 0;
}

extern "C" int jen1_ln_backward(const void* dy, const void* x, const float* stats, const float* gamma, void* dx, float* dgamma,
                                float* dbeta, int rows, int C, int ld, int dtype, void* stream) {
  return jen1_ln_backward_add(dy, x, stats, gamma, dx, nullptr, dgamma, dbeta, rows, C, ld, dtype, stream);
}

extern "C" int jen1_ln_backward_add(const void* dy, const void* x, const float* stats, const float* gamma, void* dx, const void* dx_add,
                                    float* dgamma, float* dbeta, int rows, int C, int ld, int dtype, void* stream) {
  if (check_dtype(dtype, "jen1_ln_backward")) return 1;
  JEN1_CHECK(dy && x && stats && gamma && dgamma && dbeta, "jen1_ln_backward: NULL argument");
  JEN1_CHECK(dx != nullptr || dx_add == nullptr, "jen1_ln_backward_add: dx_add without dx");
  JEN1_CHECK(rows >= 1 && C >= 1 && ld >= C && C <= 64 * LN_MAXPL, "jen1_ln_backward: bad shape rows=%d C=%d ld=%d", rows, C, ld);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // many rows (the text context: 2B x 129): 8 waves per block meet in LDS, so 32 blocks x 2 C atomics finish the column sums
  // (the atomics of hundreds of waves on 2 C addresses were the kernel's time: 26 us at 2 064 x 1 024); 16 waves per block would
  // leave 128 registers per lane and push the per-lane column sums into scratch
  if ((C & 7) == 0 && (ld & 7) == 0 && C <= 512 * LN_MAXV && (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)dx_add | (uintptr_t)gamma) & 15) == 0 &&
      (size_t)7 * 2 * C * sizeof(float) <= 64 * 1024) {
    // 8 waves per block when there are rows for them, else 4; at most 32 blocks: each wave keeps its column sums over many rows
    const bool big8 = rows >= 256;
    const int wpb = big8 ? 8 : 4;
    int blocks = (rows + wpb - 1) / wpb;
    if (blocks > 32) blocks = 32;
    const size_t lds = (size_t)(wpb - 1) * 2 * C * sizeof(float);
    if (big8) {
      if (dtype == JEN1_F32) hipLaunchKernelGGL((ln_bwd_vec_kernel<float, 512>), dim3(blocks), dim3(512), lds, s, dy, x, stats, gamma, dx, dx_add, dgamma, dbeta, rows, C, ld);
      else hipLaunchKernelGGL((ln_bwd_vec_kernel<bf16_t, 512>), dim3(blocks), dim3(512), lds, s, dy, x, stats, gamma, dx, dx_add, dgamma, dbeta, rows, C, ld);
    } else {
      if (dtype == JEN1_F32) hipLaunchKernelGGL((ln_bwd_vec_kernel<float, 256>), dim3(blocks), dim3(256), lds, s, dy, x, stats, gamma, dx, dx_add, dgamma, dbeta, rows, C, ld);
      else hipLaunchKernelGGL((ln_bwd_vec_kernel<bf16_t, 256>), dim3(blocks), dim3(256), lds, s, dy, x, stats, gamma, dx, dx_add, dgamma, dbeta, rows, C, ld);
    }
    JEN1_HIP(hipGetLastError());
    return 0;
  }
  const bool big = rows >= 512 && (size_t)7 * 2 * C * sizeof(float) <= 64 * 1024;
  if (big) {
    int blocks = (rows + 31) / 32;
    if (blocks > 32) blocks = 32;
    const size_t lds = (size_t)7 * 2 * C * sizeof(float);
    if (dtype == JEN1_F32) hipLaunchKernelGGL((ln_bwd_kernel<float, 512>), dim3(blocks), dim3(512), lds, s, dy, x, stats, gamma, dx, dx_add, dgamma, dbeta, rows, C, ld);
    else hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, 512>), dim3(blocks), dim3(512), lds, s, dy, x, stats, gamma, dx, dx_add, dgamma, dbeta, rows, C, ld);
    JEN1_HIP(hipGetLastError());
    return 0;
  }
  int blocks = (rows + 3) / 4;
  if (blocks > 64) blocks = 64;             // each wave keeps column sums over many rows: few atomics
  const size_t lds = (size_t)3 * 2 * C * sizeof(float);
  if (dtype == JEN1_F32) hipLaunchKernelGGL((ln_bwd_kernel<float, 256>), dim3(blocks), dim3(256), lds, s, dy, x, stats, gamma, dx, dx_add, dgamma, dbeta, rows, C, ld);
  else hipLaunchKernelGGL((ln_bwd_kernel<bf16_t, 256>), dim3(blocks), dim3(256), lds, s, dy, x, stats, gamma, dx, dx_add, dgamma, dbeta, rows, C, ld);
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_ln2_forward(const void* x, const float* gamma1, const float* beta1, const float* gamma2, const float* beta2, void* y1,
                                void* y2, float* stats, int rows, int C, int ld, float eps, int dtype, void* stream) {
  if (check_dtype(dtype, "jen1_ln2_forward")) return 1;
  JEN1_CHECK(x && gamma1 && beta1 && gamma2 && beta2 && y1 && y2 && stats, "jen1_ln2_forward: NULL argument");
  JEN1_CHECK(rows >= 1 && C >= 8 && ld >= C && (C & 7) == 0 && (ld & 7) == 0 && C <= 512 * LN2_MAXV,
             "jen1_ln2_forward: rows of a multiple of 8 channels, C <= %d (rows=%d C=%d ld=%d)", 512 * LN2_MAXV, rows, C, ld);
  JEN1_CHECK((((uintptr_t)x | (uintptr_t)y1 | (uintptr_t)y2 | (uintptr_t)gamma1 | (uintptr_t)beta1 | (uintptr_t)gamma2 | (uintptr_t)beta2) & 15) == 0,
             "jen1_ln2_forward: 16-byte aligned tensors");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  DISPATCH(dtype, ln2_fwd_vec_kernel, dim3((rows + 3) / 4), x, gamma1, beta1, gamma2, beta2, y1, y2, stats, rows, C, ld, eps);
  return 0;
}

extern "C" int jen1_ln2_backward_add(const void* dy1, const void* dy2, const void* x, const float* stats, const float* gamma1,
                                     const float* gamma2, void* dx, const void* dx_add, float* dgamma1, float* dbeta1, float* dgamma2,
                                     float* dbeta2, int rows, int C, int ld, int dtype, void* stream) {
  if (check_dtype(dtype, "jen1_ln2_backward_add")) return 1;
  JEN1_CHECK(dy1 && dy2 && x && stats && gamma1 && gamma2 && dx && dgamma1 && dbeta1 && dgamma2 && dbeta2, "jen1_ln2_backward_add: NULL argument");
  JEN1_CHECK(rows >= 1 && C >= 8 && ld >= C && (C & 7) == 0 && (ld & 7) == 0 && C <= 512 * LN2_MAXV,
             "jen1_ln2_backward_add: rows of a multiple of 8 channels, C <= %d (rows=%d C=%d ld=%d)", 512 * LN2_MAXV, rows, C, ld);
  JEN1_CHECK((((uintptr_t)dy1 | (uintptr_t)dy2 | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)dx_add | (uintptr_t)gamma1 | (uintptr_t)gamma2) & 15) == 0,
             "jen1_ln2_backward_add: 16-byte aligned tensors");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int blocks = (rows + 3) / 4;
  if (blocks > 32) blocks = 32;
  const size_t lds = (size_t)3 * 4 * C * sizeof(float);
  if (dtype == JEN1_F32) hipLaunchKernelGGL((ln2_bwd_vec_kernel<float, 256>), dim3(blocks), dim3(256), lds, s, dy1, dy2, x, stats, gamma1, gamma2, dx, dx_add, dgamma1, dbeta1, dgamma2, dbeta2, rows, C, ld);
  else hipLaunchKernelGGL((ln2_bwd_vec_kernel<bf16_t, 256>), dim3(blocks), dim3(256), lds, s, dy1, dy2, x, stats, gamma1, gamma2, dx, dx_add, dgamma1, dbeta1, dgamma2, dbeta2, rows, C, ld);
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_act_forward(const void* x, void* y, int64_t n, int mode, int dtype, void* stream) {
  if (check_dtype(dtype, "jen1_act_forward")) return 1;
  JEN1_CHECK(x && y && n >= 1 && mode >= 0 && mode <= 2, "jen1_act_forward: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if ((n & 7) == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {
    DISPATCH(dtype, act_fwd_vec_kernel, dim3(ew_grid(n / 8)), x, y, (long long)(n / 8), mode);
    return 0;
  }
  DISPATCH(dtype, act_fwd_kernel, dim3(ew_grid(n)), x, y, (long long)n, mode);
  return 0;
}

extern "C" int jen1_act_backward(const void* dy, const void* x, void* dx, int64_t n, int mode, int dtype, void* stream) {
  if (check_dtype(dtype, "jen1_act_backward")) return 1;
  JEN1_CHECK(dy && x && dx && n >= 1 && mode >= 0 && mode <= 2, "jen1_act_backward: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if ((n & 7) == 0 && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0) {
    DISPATCH(dtype, act_bwd_vec_kernel, dim3(ew_grid(n / 8)), dy, x, dx, (long long)(n / 8), mode);
    return 0;
  }
  DISPATCH(dtype, act_bwd_kernel, dim3(ew_grid(n)), dy, x, dx, (long long)n, mode);
  return 0;
}

extern "C" int jen1_softmax_forward(const float* sc, void* p, int rows, int Nq, int Nk, int ld_s, int ld_p, int causal, int dtype,
                                    void* stream) {
  if (check_dtype(dtype, "jen1_softmax_forward")) return 1;
  JEN1_CHECK(sc && p && rows >= 1 && Nq >= 1 && Nk >= 1 && ld_s >= Nk && ld_p >= Nk && rows % Nq == 0, "jen1_softmax_forward: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  DISPATCH(dtype, softmax_fwd_kernel, dim3((rows + 3) / 4), sc, p, rows, Nq, Nk, ld_s, ld_p, causal);
  return 0;
}

extern "C" int jen1_softmax_backward(const void* p, const float* dp, void* ds, int rows, int Nk, int ld_s, int ld_p, int dtype,
                                     void* stream) {
  if (check_dtype(dtype, "jen1_softmax_backward")) return 1;
  JEN1_CHECK(p && dp && ds && rows >= 1 && Nk >= 1 && ld_s >= Nk && ld_p >= Nk, "jen1_softmax_backward: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  DISPATCH(dtype, softmax_bwd_kernel, dim3((rows + 3) / 4), p, dp, ds, rows, Nk, ld_s, ld_p);
  return 0;
}

extern "C" int jen1_colsum(const void* x, float* out, int rows, int C, int ld, int dtype, void* stream) {
  if (check_dtype(dtype, "jen1_colsum")) return 1;
  JEN1_CHECK(x && out && rows >= 1 && C >= 1 && ld >= C, "jen1_colsum: bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int es = dtype == JEN1_F32 ? 4 : 2;
  if (rows >= 2048 && C % 8 == 0 && ld % 8 == 0 && ((uintptr_t)x & 15) == 0 && ((long long)ld * es) % 16 == 0) {
    int blocks = rows / 1024;
    blocks = blocks < 1 ? 1 : (blocks > 32 ? 32 : blocks);
    const int rpb = (rows + blocks - 1) / blocks;
    DISPATCH(dtype, colsum_rows_kernel, dim3((rows + rpb - 1) / rpb), x, out, rows, C, ld, rpb);
    return 0;
  }
  int CT, rpb, gx, gy;
  red_geom(C, rows, CT, rpb, gx, gy);
  DISPATCH(dtype, colsum_kernel, dim3(gx, gy), x, out, rows, C, ld, CT, rpb);
  return 0;
}

namespace {
// every compute copy of the parameters in ONE launch (they are refreshed after each optimiser step: [k][C_out][C_in] and its
// transposes from the reference layouts -- 387 strided torch copies of 10 us each, 4 ms per step, before this).  A block moves one
// 32 x 32 tile of one (entry, i0) slice; lanes run along whichever of the two inner axes is closer to contiguous in the source and
// the tile turns in LDS when that is not the destination's inner axis.
template <typename T>
__global__ __launch_bounds__(256) void repack_kernel(const jen1_repack_entry* __restrict__ ent, int n, int total_tiles) {
  __shared__ float tile[32][33];
  const int tileno = blockIdx.x;
  if (tileno >= total_tiles) return;
  int lo = 0, hi = n - 1;                              // last entry with tile0 <= tileno
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (ent[mid].tile0 <= tileno) lo = mid; else hi = mid - 1;
  }
  const jen1_repack_entry e = ent[lo];
  const int t1n = (e.d1 + 31) >> 5, t2n = (e.d2 + 31) >> 5;
  int r = tileno - e.tile0;
  const int i0 = r / (t1n * t2n);
  r -= i0 * t1n * t2n;
  const int r1 = (r / t2n) * 32, c2 = (r - (r / t2n) * t2n) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  T* dst = reinterpret_cast<T*>(e.dst);
  const float* src = e.src + (long long)i0 * e.s0;
  if (e.dst2 != nullptr) {
    // both compute copies of a weight from ONE read of the parameter: dst [i0][i1][i2] and dst2 [i0][i2][i1] (the data-gradient
    // transpose).  The tile is read along whichever inner axis is closer to contiguous in the source and written once each way.
    T* dst2 = reinterpret_cast<T*>(e.dst2);
    const bool along2 = e.s2 <= e.s1;
    for (int k = ty; k < 32; k += 8) {
      const int i1 = along2 ? r1 + k : r1 + tx, i2 = along2 ? c2 + tx : c2 + k;
      const float v = (i1 < e.d1 && i2 < e.d2) ? src[(long long)i1 * e.s1 + (long long)i2 * e.s2] : 0.f;
      if (along2) tile[k][tx] = v; else tile[tx][k] = v;           // tile[i1 - r1][i2 - c2]
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
      int i1 = r1 + k, i2 = c2 + tx;                                // dst: lanes along i2
      if (i1 < e.d1 && i2 < e.d2) dst[((long long)i0 * e.d1 + i1) * e.ld + i2] = (T)tile[k][tx];
      i1 = r1 + tx; i2 = c2 + k;                                    // dst2: lanes along i1
      if (i1 < e.d1 && i2 < e.d2) dst2[((long long)i0 * e.d2 + i2) * e.ld2 + i1] = (T)tile[tx][k];
    }
    return;
  }
  if (e.s2 <= e.s1) {
    for (int k = ty; k < 32; k += 8) {
      const int i1 = r1 + k, i2 = c2 + tx;
      if (i1 < e.d1 && i2 < e.d2) dst[((long long)i0 * e.d1 + i1) * e.ld + i2] = (T)src[(long long)i1 * e.s1 + (long long)i2 * e.s2];
    }
    return;
  }
  for (int k = ty; k < 32; k += 8) {                   // lanes along i1 (the source's near-contiguous axis)
    const int i1 = r1 + tx, i2 = c2 + k;
    tile[k][tx] = (i1 < e.d1 && i2 < e.d2) ? src[(long long)i1 * e.s1 + (long long)i2 * e.s2] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int i1 = r1 + k, i2 = c2 + tx;
    if (i1 < e.d1 && i2 < e.d2) dst[((long long)i0 * e.d1 + i1) * e.ld + i2] = (T)tile[tx][k];
  }
}
}  // namespace

extern "C" int jen1_repack(const jen1_repack_entry* entries_dev, int n, int total_tiles, int dtype, void* stream) {
  if (check_dtype(dtype, "jen1_repack")) return 1;
  JEN1_CHECK(entries_dev && n >= 1 && total_tiles >= 1, "jen1_repack: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  DISPATCH(dtype, repack_kernel, dim3(total_tiles), entries_dev, n, total_tiles);
  return 0;
}

extern "C" int jen1_convert_clear_add(float* src, void* dst, const void* res, int64_t n, int dtype, void* stream) {
  if (check_dtype(dtype, "jen1_convert_clear_add")) return 1;
  JEN1_CHECK(src && dst && res && n >= 4 && (n & 3) == 0, "jen1_convert_clear_add: n must be a positive multiple of 4");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  DISPATCH(dtype, convert_clear_add_kernel, dim3(ew_grid(n / 4)), src, dst, res, (long long)(n / 4));
  return 0;
}

namespace {
// out[row] = [a[row] | scale * b[row]]  (the skip concat of the up path, blocks.py:732-734) and its transpose: one launch each
// instead of a scale launch + a concat launch forward and two strided copies + a scale launch backward
template <typename T>
__global__ __launch_bounds__(NT) void concat2_kernel(const void* a_, const void* b_, void* out_, long long rows, int Ca, int Cb, float scale) {
  const T* a = reinterpret_cast<const T*>(a_);
  const T* b = reinterpret_cast<const T*>(b_);
  T* out = reinterpret_cast<T*>(out_);
  const int va = Ca >> 3, vb = Cb >> 3, vt = va + vb;
  const long long total = rows * vt;
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
    const long long row = i / vt;
    const int v = (int)(i - row * vt);
    float w[8];
    if (v < va) {
      load8(a + row * Ca + v * 8, w);
    } else {
      load8(b + row * Cb + (v - va) * 8, w);
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] *= scale;
    }
    store8(out + row * (Ca + Cb) + v * 8, w);
  }
}
template <typename T>
__global__ __launch_bounds__(NT) void split2_kernel(const void* d_, void* da_, void* db_, long long rows, int Ca, int Cb, float scale) {
  const T* d = reinterpret_cast<const T*>(d_);
  T* da = reinterpret_cast<T*>(da_);
  T* db = reinterpret_cast<T*>(db_);
  const int va = Ca >> 3, vb = Cb >> 3, vt = va + vb;
  const long long total = rows * vt;
  for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < total; i += (long long)gridDim.x * NT) {
    const long long row = i / vt;
    const int v = (int)(i - row * vt);
    float w[8];
    load8(d + row * (Ca + Cb) + v * 8, w);
    if (v < va) {
      store8(da + row * Ca + v * 8, w);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) w[j] *= scale;
      store8(db + row * Cb + (v - va) * 8, w);
    }
  }
}
}  // namespace

extern "C" int jen1_concat2(const void* a, const void* b, void* out, int64_t rows, int Ca, int Cb, float scale_b, int dtype, void* stream) {
  if (check_dtype(dtype, "jen1_concat2")) return 1;
  JEN1_CHECK(a && b && out && rows >= 1 && Ca >= 8 && Cb >= 8 && (Ca & 7) == 0 && (Cb & 7) == 0, "jen1_concat2: channel counts must be positive multiples of 8");
  JEN1_CHECK((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0, "jen1_concat2: pointers must be 16-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  DISPATCH(dtype, concat2_kernel, dim3(ew_grid(rows * ((Ca + Cb) / 8))), a, b, out, (long long)rows, Ca, Cb, scale_b);
  return 0;
}

extern "C" int jen1_split2(const void* d, void* da, void* db, int64_t rows, int Ca, int Cb, float scale_b, int dtype, void* stream) {
  if (check_dtype(dtype, "jen1_split2")) return 1;
  JEN1_CHECK(d && da && db && rows >= 1 && Ca >= 8 && Cb >= 8 && (Ca & 7) == 0 && (Cb & 7) == 0, "jen1_split2: channel counts must be positive multiples of 8");
  JEN1_CHECK((((uintptr_t)d | (uintptr_t)da | (uintptr_t)db) & 15) == 0, "jen1_split2: pointers must be 16-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  DISPATCH(dtype, split2_kernel, dim3(ew_grid(rows * ((Ca + Cb) / 8))), d, da, db, (long long)rows, Ca, Cb, scale_b);
  return 0;
}

extern "C" int jen1_convert_clear(float* src, void* dst, int64_t n, int dtype, void* stream) {
  if (check_dtype(dtype, "jen1_convert_clear")) return 1;
  JEN1_CHECK(src && dst && n >= 4 && (n & 3) == 0, "jen1_convert_clear: n must be a positive multiple of 4");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  DISPATCH(dtype, convert_clear_kernel, dim3(ew_grid(n / 4)), src, dst, (long long)(n / 4));
  return 0;
}
