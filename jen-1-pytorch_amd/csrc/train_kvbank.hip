// The text-context K / V projections of ALL cross-attention layers of a training pass as one product (C ABI: include/jen1_train.h).
//
// Reference: every cross-attention computes  k, v = chunk(to_kv(norm_context(context)))  (jen1/model/blocks.py:400-407, :427-434) on
// the SAME context rows -- 13 LayerNorms over [rows, 1024] and 13 bias-free Linear(1024 -> 2C) per pass, and as many data / weight
// gradient products and LayerNorm backward passes.  LayerNorm = an affine map of the standardised rows, so
//     to_kv_l(norm_context_l(x)) = xhat (W_l diag(gamma_l))^T + W_l beta_l ,      xhat = (x - mean) / sqrt(var + eps)   (shared)
// (the fold the sampling engine uses: engine.Weights "kv2_all").  The pass then runs ONE standardisation, ONE [rows x 1024] x
// [17408 x 1024]^T product on the matrix cores (jen1_big_gemm), ONE weight-gradient product into the folded layout
// (jen1_big_gemm_tn), ONE data-gradient product and ONE standardisation backward; the two kernels here move between the
// parameters and the folded operands:
//   jen1_kv_fold            W_l, gamma_l, beta_l -> Wf = bf16(W diag(gamma)) [Ntot][K], its transpose WfT [K][Ntot] (the data gradient
//                           is K-contiguous too), bias = W beta (float32).  Once per optimiser step, with the other compute copies.
//   jen1_kv_fold_backward   dWf (float32 [Ntot][K]), dbias [Ntot] -> W_l.grad += dWf diag(gamma) + dbias beta^T,
//                           gamma_l.grad += colsum(dWf * W), beta_l.grad += W^T dbias                       (chain rule of the fold)
//   jen1_sum_rows_strided   the CFG pair's unconditional half shares one set of context rows: dK / dV of the sharers are written per
//                           batch element into row blocks of the big gradient matrix and added up in place (rows a pitch apart)
#include "common.h"
#include "jen1_train.h"

namespace {

constexpr int FR = 32;                 // rows (output features o) of a workgroup
constexpr int FT = 256;                // threads: a thread owns 4 consecutive input features of every 1024-wide slab

struct LayerRef {
  const float* w;
  const float* gamma;
  const float* beta;
  float* gw;
  float* ggamma;
  float* gbeta;
  int o_local;                         // row inside the layer's own weight
};

__device__ __forceinline__ LayerRef find_layer(const jen1_kv_layer* __restrict__ tab, int n, int o) {
  int l = 0;
  for (int k = 1; k < n; ++k) l += (o >= tab[k].n0) ? 1 : 0;
  LayerRef r;
  r.w = tab[l].w; r.gamma = tab[l].gamma; r.beta = tab[l].beta;
  r.gw = tab[l].gw; r.ggamma = tab[l].ggamma; r.gbeta = tab[l].gbeta;
  r.o_local = o - tab[l].n0;
  return r;
}

__device__ __forceinline__ unsigned pack2(float a, float b) {
  const bf16_t x = (bf16_t)a, y = (bf16_t)b;
  return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}

// one workgroup: FR consecutive rows o (never straddling a layer: every n0 / N is a multiple of FR) x all K input features
__global__ __launch_bounds__(FT) void kv_fold_kernel(const jen1_kv_layer* __restrict__ tab, int n_layers, int K, bf16_t* __restrict__ wf,
                                                     bf16_t* __restrict__ wft, int ld_wft, float* __restrict__ bias) {
  __shared__ float red[FR][FT / 64];
  const int o0 = blockIdx.x * FR, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const LayerRef L_ = find_layer(tab, n_layers, o0);
  float bsum[FR];
#pragma unroll
  for (int r = 0; r < FR; ++r) bsum[r] = 0.f;
  for (int i0 = tid * 4; i0 < K; i0 += FT * 4) {
    const float4 g4 = *reinterpret_cast<const float4*>(L_.gamma + i0);
    const float4 b4 = *reinterpret_cast<const float4*>(L_.beta + i0);
    unsigned tp[4][FR / 2];                              // the transposed copy: column i0 + j, rows o0 .. o0 + FR - 1 as bf16 pairs
#pragma unroll
    for (int r = 0; r < FR; ++r) {
      const float4 w4 = *reinterpret_cast<const float4*>(L_.w + (size_t)(L_.o_local + r) * K + i0);
      const float f0 = w4.x * g4.x, f1 = w4.y * g4.y, f2 = w4.z * g4.z, f3 = w4.w * g4.w;
      bsum[r] += (w4.x * b4.x + w4.y * b4.y) + (w4.z * b4.z + w4.w * b4.w);
      uint2 pk;
      pk.x = pack2(f0, f1);
      pk.y = pack2(f2, f3);
      *reinterpret_cast<uint2*>(wf + (size_t)(o0 + r) * K + i0) = pk;
      const float f[4] = {f0, f1, f2, f3};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned short h = __builtin_bit_cast(unsigned short, (bf16_t)f[j]);
        if (r & 1) tp[j][r >> 1] |= (unsigned)h << 16;
        else tp[j][r >> 1] = (unsigned)h;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint4* dst = reinterpret_cast<uint4*>(wft + (size_t)(i0 + j) * ld_wft + o0);          // 64 contiguous bytes per column
#pragma unroll
      for (int q = 0; q < FR / 8; ++q) dst[q] = make_uint4(tp[j][4 * q], tp[j][4 * q + 1], tp[j][4 * q + 2], tp[j][4 * q + 3]);
    }
  }
  // bias[o] = sum_i W[o][i] beta[i]: wave sums, then the four waves in a fixed order
#pragma unroll
  for (int r = 0; r < FR; ++r) {
    float v = bsum[r];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) red[r][wv] = v;
  }
  __syncthreads();
  if (tid < FR) bias[o0 + tid] = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
}

__global__ __launch_bounds__(FT) void kv_fold_bwd_kernel(const jen1_kv_layer* __restrict__ tab, int n_layers, int K, const float* __restrict__ dwf,
                                                         const float* __restrict__ dbias) {
  const int o0 = blockIdx.x * FR, tid = threadIdx.x;
  const LayerRef L_ = find_layer(tab, n_layers, o0);
  for (int i0 = tid * 4; i0 < K; i0 += FT * 4) {
    const float4 g4 = *reinterpret_cast<const float4*>(L_.gamma + i0);
    const float4 b4 = *reinterpret_cast<const float4*>(L_.beta + i0);
    float cg[4] = {0.f, 0.f, 0.f, 0.f}, cb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int r = 0; r < FR; ++r) {
      const size_t wo = (size_t)(L_.o_local + r) * K + i0;
      const float4 d4 = *reinterpret_cast<const float4*>(dwf + (size_t)(o0 + r) * K + i0);
      const float4 w4 = *reinterpret_cast<const float4*>(L_.w + wo);
      const float db = dbias[o0 + r];
      float4 gw4 = *reinterpret_cast<const float4*>(L_.gw + wo);
      gw4.x += d4.x * g4.x + db * b4.x;
      gw4.y += d4.y * g4.y + db * b4.y;
      gw4.z += d4.z * g4.z + db * b4.z;
      gw4.w += d4.w * g4.w + db * b4.w;
      *reinterpret_cast<float4*>(L_.gw + wo) = gw4;
      cg[0] += d4.x * w4.x; cg[1] += d4.y * w4.y; cg[2] += d4.z * w4.z; cg[3] += d4.w * w4.w;
      cb[0] += db * w4.x; cb[1] += db * w4.y; cb[2] += db * w4.z; cb[3] += db * w4.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      atomicAdd(L_.ggamma + i0 + j, cg[j]);
      atomicAdd(L_.gbeta + i0 + j, cb[j]);
    }
  }
}

// block 0 += blocks 1 .. nblk - 1; a block = rows_per_block rows of row_elems elements, rows ld apart, blocks rows_per_block * ld apart
template <typename T>
__global__ __launch_bounds__(256) void sum_rows_strided_kernel(T* __restrict__ p, int nblk, int rows_per_block, int row_elems, long long ld) {
  const int vpr = row_elems >> 3;
  const long long nvec = (long long)rows_per_block * vpr;
  const long long bstride = (long long)rows_per_block * ld;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
    const long long row = i / vpr;
    const int c = (int)(i - row * vpr) * 8;
    T* q = p + row * ld + c;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < nblk; ++b) {
      float v[8];
      load8(q + (long long)b * bstride, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += v[j];
    }
    store8(q, acc);
  }
}

}  // namespace

extern "C" int jen1_kv_fold(const jen1_kv_layer* table_dev, int n_layers, int Ntot, int K, void* wf, void* wft, int ld_wft, float* bias, void* stream) {
  JEN1_CHECK(table_dev && wf && wft && bias && n_layers >= 1, "kv_fold: null argument");
  JEN1_CHECK(Ntot % FR == 0 && K % 4 == 0 && ld_wft >= Ntot && ld_wft % 8 == 0, "kv_fold: Ntot must be a multiple of %d, K of 4, ld_wft of 8", FR);
  hipLaunchKernelGGL(kv_fold_kernel, dim3(Ntot / FR), dim3(FT), 0, reinterpret_cast<hipStream_t>(stream), table_dev, n_layers, K,
                     reinterpret_cast<bf16_t*>(wf), reinterpret_cast<bf16_t*>(wft), ld_wft, bias);
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_kv_fold_backward(const jen1_kv_layer* table_dev, int n_layers, int Ntot, int K, const float* dwf, const float* dbias, void* stream) {
  JEN1_CHECK(table_dev && dwf && dbias && n_layers >= 1, "kv_fold_backward: null argument");
  JEN1_CHECK(Ntot % FR == 0 && K % 4 == 0, "kv_fold_backward: Ntot must be a multiple of %d, K of 4", FR);
  hipLaunchKernelGGL(kv_fold_bwd_kernel, dim3(Ntot / FR), dim3(FT), 0, reinterpret_cast<hipStream_t>(stream), table_dev, n_layers, K, dwf, dbias);
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_sum_rows_strided(void* p, int nblk, int rows_per_block, int row_elems, int64_t ld, int dtype, void* stream) {
  JEN1_CHECK(p && nblk >= 1 && rows_per_block >= 1 && row_elems >= 8 && row_elems % 8 == 0 && ld >= row_elems && ld % 8 == 0 && ((uintptr_t)p & 15) == 0,
             "sum_rows_strided: bad arguments");
  if (nblk == 1) return 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long long nvec = (long long)rows_per_block * (row_elems / 8);
  const int blocks = (int)((nvec + 255) / 256 > 1024 ? 1024 : (nvec + 255) / 256);
  if (dtype == JEN1_F32) hipLaunchKernelGGL(sum_rows_strided_kernel<float>, dim3(blocks), dim3(256), 0, s, (float*)p, nblk, rows_per_block, row_elems, (long long)ld);
  else if (dtype == JEN1_BF16) hipLaunchKernelGGL(sum_rows_strided_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (bf16_t*)p, nblk, rows_per_block, row_elems, (long long)ld);
  else return jen1_set_error("sum_rows_strided: bad dtype");
  JEN1_HIP(hipGetLastError());
  return 0;
}
