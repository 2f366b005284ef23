// The ends of a training pass that the reference writes as ATen elementwise / cat / reduce ops, as a handful of launches
// (C ABI: include/jen1_train.h).  Between them sits the network itself (train_gemm / train_ops / train_attn / big_gemm).
//   jen1_train_pack_input   q_sample (gdm.py:232-243) + torch.cat([x_t, input_concat_cond]) (model.py:240) + the CFG pair's
//                           torch.cat([x, x]) (model.py:332) + the layout change to channel-last rows, in one pass
//   jen1_train_context      cat([embedding, time token]) (model.py:315-316), CFG dropout rows swapped to the fixed embedding
//                           (model.py:323-328), the unconditional half of the pair (model.py:333), cast to the compute dtype;
//                           its backward: gradients of the time token and of the fixed embedding
//   jen1_time_features      [t, sin(2 pi t w), cos(2 pi t w)] (utils/module.py:58-72) and its backward (gradient of w)
//   jen1_cfg_loss           out_masked + (out - out_masked) s, the unbiased-std rescale (model.py:362-369), the target of the
//                           objective (gdm.py:260-266), l2 / l1 and the mean over (C, T) (gdm.py:268-272) -> per-sample losses;
//                           its backward writes the gradient of the network's output rows (both halves of the pair) directly
#include "common.h"
#include "jen1_train.h"

namespace {

// ---- pack: [B][C][T] float32 sources -> rows [nrep * B][T][ld] -------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pack_input_train_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                                               const float* __restrict__ ca, const float* __restrict__ cb,
                                                               const float* __restrict__ ctx, T* __restrict__ y, int B, int C, int Cc, int Tn,
                                                               int ld, int nrep, const float* __restrict__ ta, const float* __restrict__ tb,
                                                               float* __restrict__ tgt) {
  __shared__ float tile[32][33];
  __shared__ float tile2[32][33];        // the loss target ta[b] noise + tb[b] x0 (gdm.py:260-266), written as rows [B][T][C] too
  const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32, b = blockIdx.z;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float a_ = ca ? ca[b] : 1.0f, b_ = cb ? cb[b] : 0.0f;
  const float ta_ = tgt ? ta[b] : 0.f, tb_ = tgt ? tb[b] : 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, t = t0 + tx;
    float v = 0.f, v2 = 0.f;
    if (t < Tn) {
      if (c < C) {
        const size_t i = ((size_t)b * C + c) * Tn + t;
        const float xs = x0[i], nz = noise ? noise[i] : 0.f;
        v = a_ * xs + b_ * nz;
        v2 = ta_ * nz + tb_ * xs;
      } else if (c < C + Cc) {
        v = ctx[((size_t)b * Cc + (c - C)) * Tn + t];
      }
    }
    tile[ty + 8 * k][tx] = v;
    tile2[ty + 8 * k][tx] = v2;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int t = t0 + ty + 8 * k, c = c0 + tx;
    if (t < Tn && c < ld) {
      const T v = (T)tile[tx][ty + 8 * k];
      for (int r = 0; r < nrep; ++r) y[((size_t)(r * B + b) * Tn + t) * ld + c] = v;
    }
    if (tgt && t < Tn && c < C) tgt[((size_t)b * Tn + t) * C + c] = tile2[tx][ty + 8 * k];
  }
}

// ---- context rows -----------------------------------------------------------------------------------------------------------
// out[r][n][:] for r < B: drop[r] ? fixed[n] : (n < NL ? emb[r][n] : tok[r]);  r >= B (the pair's second half): fixed[n]
template <typename T>
__global__ __launch_bounds__(256) void context_fwd_kernel(const float* __restrict__ emb, const float* __restrict__ tok, const float* __restrict__ fixed,
                                                          const unsigned char* __restrict__ drop, T* __restrict__ out, int B, int NL, int N, int F,
                                                          int rows_total) {
  const int vpr = F >> 2;
  const long long total = (long long)rows_total * N * vpr;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % vpr) * 4;
    const long long rn = i / vpr;
    const int n = (int)(rn % N), r = (int)(rn / N);
    const bool use_fixed = r >= B || (drop && drop[r]);
    const float* src = use_fixed ? fixed + (size_t)n * F : (n < NL ? emb + ((size_t)r * NL + n) * F : tok + (size_t)r * F);
    float v[4];
    load4(src + c, v);
    store4(out + ((size_t)r * N + n) * F + c, v);
  }
}
// d_fixed[n][c] += sum over rows that used the fixed embedding;  d_tok[r][c] = d[r][N - 1][c] for the kept rows (else 0)
template <typename T>
__global__ __launch_bounds__(256) void context_bwd_kernel(const T* __restrict__ d, const unsigned char* __restrict__ drop, float* __restrict__ d_fixed,
                                                          float* __restrict__ d_tok, int B, int NL, int N, int F, int rows_total) {
  const int vpr = F >> 2;
  const int total = N * vpr;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int c = (i % vpr) * 4, n = i / vpr;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < rows_total; ++r) {
      if (r >= B || (drop && drop[r])) {
        float v[4];
        load4(d + ((size_t)r * N + n) * F + c, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += v[j];
      }
    }
    float o[4];
    load4(d_fixed + (size_t)n * F + c, o);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] += acc[j];
    store4(d_fixed + (size_t)n * F + c, o);
    if (d_tok && n == N - 1 && N > NL) {
      for (int r = 0; r < B; ++r) {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (!(drop && drop[r])) load4(d + ((size_t)r * N + n) * F + c, v);
        store4(d_tok + (size_t)r * F + c, v);
      }
    }
  }
}

// ---- time features ----------------------------------------------------------------------------------------------------------
// f[b] = [t, sin(((t w) 2) pi), cos(...)] padded with zeros to ld; phases evaluated left to right in float32 like the reference
template <typename TT>
__global__ __launch_bounds__(256) void time_feat_fwd_kernel(const TT* __restrict__ t, const float* __restrict__ w, float* __restrict__ f, int B, int half, int ld) {
  const int b = blockIdx.x;
  const float tv = (float)t[b];
  for (int i = threadIdx.x; i < ld; i += 256) {
    float v = 0.f;
    if (i == 0) v = tv;
    else if (i <= 2 * half) {
      const int k = (i - 1) % half;
      const float ph = __fmul_rn(__fmul_rn(__fmul_rn(tv, w[k]), 2.0f), 3.14159274101257324f);
      v = (i <= half) ? sinf(ph) : cosf(ph);
    }
    f[(size_t)b * ld + i] = v;
  }
}
// dw[k] += sum_b (df[b][1 + k] cos(ph) - df[b][1 + half + k] sin(ph)) * t_b * 2 pi
template <typename TT>
__global__ __launch_bounds__(256) void time_feat_bwd_kernel(const TT* __restrict__ t, const float* __restrict__ w, const float* __restrict__ df,
                                                            float* __restrict__ dw, int B, int half, int ld) {
  for (int k = blockIdx.x * 256 + threadIdx.x; k < half; k += gridDim.x * 256) {
    float acc = 0.f;
    for (int b = 0; b < B; ++b) {
      const float tv = (float)t[b];
      const float ph = __fmul_rn(__fmul_rn(__fmul_rn(tv, w[k]), 2.0f), 3.14159274101257324f);
      acc += (df[(size_t)b * ld + 1 + k] * cosf(ph) - df[(size_t)b * ld + 1 + half + k] * sinf(ph)) * tv * 2.0f * 3.14159274101257324f;
    }
    dw[k] += acc;
  }
}

// ---- CFG combine + rescale + loss -------------------------------------------------------------------------------------------
// one wave per (b, t) row, lane owns channels lane, lane + 64, ... (C <= 256)
constexpr int CPL = 4;
struct RowOut {
  float oc[CPL], ou[CPL], g[CPL], y[CPL];
  float mc, mg, sc, sg, k;
};
template <typename T>
__device__ __forceinline__ void cfg_row(const T* __restrict__ pc, const T* __restrict__ pu, int lane, int C, int nrep, float s, int scale_cfg, float phi, RowOut& r) {
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int c = lane + 64 * j;
    r.oc[j] = c < C ? (float)pc[c] : 0.f;
    r.ou[j] = (nrep == 2 && c < C) ? (float)pu[c] : 0.f;
    r.g[j] = nrep == 2 ? r.ou[j] + (r.oc[j] - r.ou[j]) * s : r.oc[j];
  }
  r.k = 1.0f;
  r.mc = r.mg = 0.f;
  r.sc = r.sg = 1.f;
  if (nrep == 2 && scale_cfg) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) { a += r.oc[j]; b += r.g[j]; }
    a = wave_sum(a); b = wave_sum(b);
    r.mc = a / (float)C;
    r.mg = b / (float)C;
    float va = 0.f, vb = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const bool in = lane + 64 * j < C;
      const float dc = r.oc[j] - r.mc, dg = r.g[j] - r.mg;
      va += in ? dc * dc : 0.f;
      vb += in ? dg * dg : 0.f;
    }
    va = wave_sum(va); vb = wave_sum(vb);
    r.sc = sqrtf(va / (float)(C - 1));           // unbiased (torch.std)
    r.sg = sqrtf(vb / (float)(C - 1));
    r.k = phi * (r.sc / r.sg) + (1.0f - phi);
  }
#pragma unroll
  for (int j = 0; j < CPL; ++j) r.y[j] = nrep == 2 && scale_cfg ? phi * (r.g[j] * (r.sc / r.sg)) + (1.0f - phi) * r.g[j] : r.g[j];
}

// loss_ps[b] += sum over this block's rows of elem(y - target) / (C T);  target = ta[b] noise + tb[b] x0
template <typename T>
__global__ __launch_bounds__(256) void cfg_loss_fwd_kernel(const T* __restrict__ net, const float* __restrict__ tgt, float* __restrict__ loss_ps,
                                                           int B, int C, int Tn, int ld, int nrep, float s, int scale_cfg, float phi, int l1) {
  __shared__ float red[4];
  const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int t = blockIdx.x * 4 + wave; t < Tn; t += gridDim.x * 4) {
    RowOut r;
    cfg_row(net + ((size_t)b * Tn + t) * ld, net + ((size_t)(B + b) * Tn + t) * ld, lane, C, nrep, s, scale_cfg, phi, r);
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int c = lane + 64 * j;
      if (c < C) {
        const float d = r.y[j] - tgt[((size_t)b * Tn + t) * C + c];
        acc += l1 ? fabsf(d) : d * d;
      }
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss_ps + b, ((red[0] + red[1]) + (red[2] + red[3])) / ((float)C * (float)Tn));
}

template <typename T>
__global__ __launch_bounds__(256) void cfg_loss_bwd_kernel(const T* __restrict__ net, const float* __restrict__ tgt, const float* __restrict__ gps,
                                                           T* __restrict__ dnet, int B, int C, int Tn, int ld, int nrep, float s, int scale_cfg,
                                                           float phi, int l1) {
  const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float gscale = gps[b] / ((float)C * (float)Tn);
  for (int t = blockIdx.x * 4 + wave; t < Tn; t += gridDim.x * 4) {
    RowOut r;
    cfg_row(net + ((size_t)b * Tn + t) * ld, net + ((size_t)(B + b) * Tn + t) * ld, lane, C, nrep, s, scale_cfg, phi, r);
    float dy[CPL], dg[CPL], doc[CPL];
    float dk = 0.f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int c = lane + 64 * j;
      dy[j] = 0.f;
      if (c < C) {
        const float d = r.y[j] - tgt[((size_t)b * Tn + t) * C + c];
        dy[j] = gscale * (l1 ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 2.0f * d);
      }
      dg[j] = dy[j] * r.k;
      doc[j] = 0.f;
      dk += dy[j] * r.g[j];
    }
    if (nrep == 2 && scale_cfg) {
      dk = wave_sum(dk);
      const float dr = phi * dk;
      const float dsc = dr / r.sg, dsg = -dr * r.sc / (r.sg * r.sg);
      const float fc = dsc / ((float)(C - 1) * r.sc), fg = dsg / ((float)(C - 1) * r.sg);
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        doc[j] = fc * (r.oc[j] - r.mc);
        dg[j] += fg * (r.g[j] - r.mg);
      }
    }
    T* dc = dnet + ((size_t)b * Tn + t) * ld;
    T* du = dnet + ((size_t)(B + b) * Tn + t) * ld;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const int c = lane + 64 * j;
      if (c < C) {
        if (nrep == 2) {
          dc[c] = (T)(doc[j] + s * dg[j]);
          du[c] = (T)((1.0f - s) * dg[j]);
        } else {
          dc[c] = (T)dg[j];
        }
      } else if (c < ld) {
        dc[c] = (T)0.f;
        if (nrep == 2) du[c] = (T)0.f;
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void sum_rows_kernel(T* __restrict__ p, int rows, long long n) {
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += (long long)gridDim.x * 256 * 8) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < rows; ++r) {
      float v[8];
      load8(p + (long long)r * n + i, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += v[j];
    }
    store8(p + i, acc);
  }
}

__global__ void zero_f32_kernel(float* p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

}  // namespace

extern "C" int jen1_train_pack_input(const float* x0, const float* noise, const float* ca, const float* cb, const float* ctx, void* y, int B, int C,
                                     int Cc, int T, int ld, int nrep, const float* ta, const float* tb, float* tgt, int dtype, void* stream) {
  JEN1_CHECK(x0 && y && (Cc == 0 || ctx) && B >= 1 && C >= 1 && T >= 1 && ld >= C + Cc && ld % 8 == 0 && nrep >= 1 && nrep <= 2 &&
             (!tgt || (ta && tb)), "train_pack_input: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((T + 31) / 32, (ld + 31) / 32, B);
  if (dtype == JEN1_F32) hipLaunchKernelGGL(pack_input_train_kernel<float>, grid, dim3(256), 0, s, x0, noise, ca, cb, ctx, (float*)y, B, C, Cc, T, ld, nrep, ta, tb, tgt);
  else if (dtype == JEN1_BF16) hipLaunchKernelGGL(pack_input_train_kernel<bf16_t>, grid, dim3(256), 0, s, x0, noise, ca, cb, ctx, (bf16_t*)y, B, C, Cc, T, ld, nrep, ta, tb, tgt);
  else return jen1_set_error("train_pack_input: bad dtype");
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_train_context(const float* emb, const float* tok, const float* fixed, const uint8_t* drop, void* out, int B, int NL, int N, int F,
                                  int nrep, int dtype, void* stream) {
  JEN1_CHECK(emb && fixed && out && B >= 1 && NL >= 1 && (N == NL || (N == NL + 1 && tok)) && F % 4 == 0 && nrep >= 0 && nrep <= 2,
             "train_context: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int rows = nrep == 0 ? B + 1 : nrep * B;        // nrep = 0: ONE shared set of unconditional rows behind the B conditional ones
  const long long total = (long long)rows * N * (F / 4);
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  if (dtype == JEN1_F32) hipLaunchKernelGGL(context_fwd_kernel<float>, dim3(blocks), dim3(256), 0, s, emb, tok, fixed, drop, (float*)out, B, NL, N, F, rows);
  else if (dtype == JEN1_BF16) hipLaunchKernelGGL(context_fwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, emb, tok, fixed, drop, (bf16_t*)out, B, NL, N, F, rows);
  else return jen1_set_error("train_context: bad dtype");
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_train_context_backward(const void* d, const uint8_t* drop, float* d_fixed, float* d_tok, int B, int NL, int N, int F, int nrep,
                                           int dtype, void* stream) {
  JEN1_CHECK(d && d_fixed && B >= 1 && (N == NL || N == NL + 1) && F % 4 == 0 && nrep >= 0 && nrep <= 2, "train_context_backward: bad arguments");
  const int rows_total = nrep == 0 ? B + 1 : nrep * B;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int total = N * (F / 4);
  const int blocks = (total + 255) / 256;
  if (dtype == JEN1_F32) hipLaunchKernelGGL(context_bwd_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)d, drop, d_fixed, d_tok, B, NL, N, F, rows_total);
  else if (dtype == JEN1_BF16) hipLaunchKernelGGL(context_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)d, drop, d_fixed, d_tok, B, NL, N, F, rows_total);
  else return jen1_set_error("train_context_backward: bad dtype");
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_time_features_fwd(const void* t, int t_is_float, const float* w, float* f, int B, int half, int ld, void* stream) {
  JEN1_CHECK(t && w && f && B >= 1 && half >= 1 && ld >= 2 * half + 1, "time_features_fwd: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (t_is_float) hipLaunchKernelGGL(time_feat_fwd_kernel<float>, dim3(B), dim3(256), 0, s, (const float*)t, w, f, B, half, ld);
  else hipLaunchKernelGGL(time_feat_fwd_kernel<int64_t>, dim3(B), dim3(256), 0, s, (const int64_t*)t, w, f, B, half, ld);
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_time_features_bwd(const void* t, int t_is_float, const float* w, const float* df, float* dw, int B, int half, int ld, void* stream) {
  JEN1_CHECK(t && w && df && dw && B >= 1 && half >= 1 && ld >= 2 * half + 1, "time_features_bwd: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (t_is_float) hipLaunchKernelGGL(time_feat_bwd_kernel<float>, dim3((half + 255) / 256), dim3(256), 0, s, (const float*)t, w, df, dw, B, half, ld);
  else hipLaunchKernelGGL(time_feat_bwd_kernel<int64_t>, dim3((half + 255) / 256), dim3(256), 0, s, (const int64_t*)t, w, df, dw, B, half, ld);
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_cfg_loss_forward(const void* net, const float* tgt, float* loss_ps, int B, int C, int T, int ld, int nrep, float embedding_scale,
                                     int scale_cfg, float scale_phi, int l1, int dtype, void* stream) {
  JEN1_CHECK(net && tgt && loss_ps && B >= 1 && C >= 2 && C <= 64 * CPL && ld >= C && nrep >= 1 && nrep <= 2, "cfg_loss_forward: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(zero_f32_kernel, dim3((B + 255) / 256), dim3(256), 0, s, loss_ps, B);
  const int bx = (T + 3) / 4 > 96 ? 96 : (T + 3) / 4;
  if (dtype == JEN1_F32) hipLaunchKernelGGL(cfg_loss_fwd_kernel<float>, dim3(bx, B), dim3(256), 0, s, (const float*)net, tgt, loss_ps, B, C, T, ld, nrep, embedding_scale, scale_cfg, scale_phi, l1);
  else if (dtype == JEN1_BF16) hipLaunchKernelGGL(cfg_loss_fwd_kernel<bf16_t>, dim3(bx, B), dim3(256), 0, s, (const bf16_t*)net, tgt, loss_ps, B, C, T, ld, nrep, embedding_scale, scale_cfg, scale_phi, l1);
  else return jen1_set_error("cfg_loss_forward: bad dtype");
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_cfg_loss_backward(const void* net, const float* tgt, const float* gps, void* dnet, int B, int C, int T, int ld, int nrep,
                                      float embedding_scale, int scale_cfg, float scale_phi, int l1, int dtype, void* stream) {
  JEN1_CHECK(net && tgt && gps && dnet && B >= 1 && C >= 2 && C <= 64 * CPL && ld >= C && nrep >= 1 && nrep <= 2, "cfg_loss_backward: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int bx = (T + 3) / 4 > 96 ? 96 : (T + 3) / 4;
  if (dtype == JEN1_F32) hipLaunchKernelGGL(cfg_loss_bwd_kernel<float>, dim3(bx, B), dim3(256), 0, s, (const float*)net, tgt, gps, (float*)dnet, B, C, T, ld, nrep, embedding_scale, scale_cfg, scale_phi, l1);
  else if (dtype == JEN1_BF16) hipLaunchKernelGGL(cfg_loss_bwd_kernel<bf16_t>, dim3(bx, B), dim3(256), 0, s, (const bf16_t*)net, tgt, gps, (bf16_t*)dnet, B, C, T, ld, nrep, embedding_scale, scale_cfg, scale_phi, l1);
  else return jen1_set_error("cfg_loss_backward: bad dtype");
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_sum_rows_inplace(void* p, int rows, int64_t n, int dtype, void* stream) {
  JEN1_CHECK(p && rows >= 1 && n >= 8 && n % 8 == 0 && ((uintptr_t)p & 15) == 0, "sum_rows_inplace: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int blocks = (int)((n / 8 + 255) / 256 > 1024 ? 1024 : (n / 8 + 255) / 256);
  if (dtype == JEN1_F32) hipLaunchKernelGGL(sum_rows_kernel<float>, dim3(blocks), dim3(256), 0, s, (float*)p, rows, (long long)n);
  else if (dtype == JEN1_BF16) hipLaunchKernelGGL(sum_rows_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (bf16_t*)p, rows, (long long)n);
  else return jen1_set_error("sum_rows_inplace: bad dtype");
  JEN1_HIP(hipGetLastError());
  return 0;
}
