// jen1_attn_small_forward / jen1_attn_small_backward: the attention core of the training path (AttentionBase.forward, math path:
// reference blocks.py:355-380) for SHORT sequences -- the transformer blocks of JEN-1 sit at the deep levels of the UNet, where a
// 1500-frame clip is 1 .. 24 positions and the text context 130 tokens.  As three + five launches of jen1_train_gemm / softmax per
// attention (208 launches per pass) these products are pure launch latency: a (batch element, head) is 24 x 130 x 64 multiply-adds.
// Here one workgroup owns one (batch element, head): Q, K, V (and dO) go into LDS once (as float32), the scores, the softmax, P V -- and in the
// backward pass dP, dS, dQ, dK, dV -- are computed from there in float32 on the vector units.  Results follow the GEMM path's
// roundings where they are visible to the rest of the pass: P is rounded to the activations' dtype before P V and is what the
// backward pass reads; dS stays float32 (the GEMM path rounds it once more).
#include "common.h"
#include "jen1_train.h"

namespace {

constexpr int ANT = 512;

struct AttnDev {
  const void* q; const void* k; const void* v; const void* d_o;
  void* o; void* p; void* dq; void* dk; void* dv;
  long long ldq, ldk, ldv, ldo, ldp, lddq, lddk, lddv;      // row pitches (elements)
  int B, H, Nq, Nk, d, causal;
  float scale;
  const int* causal_b;      // per batch element: causal or not (a pass that mixes both), or NULL: `causal` for all
  const float* kmask;       // [B][Nk] float32 multiplied into the rows of K and V (the padding mask of the context, blocks.py:431-434), or NULL
  const int* kv_row;        // which batch row of k / v batch element b READS (NULL: b): the unconditional half of a CFG pair shares ONE set of
                            // context rows (the fixed embedding, model.py:333), projected once instead of once per batch element
};

// rows [n][d] of one head from a [B][n][ld] tensor -> float32 LDS rows of pitch dp; 16-byte global vectors when the head's rows
// start on 16-byte boundaries
template <typename T>
__device__ __forceinline__ void stage_rows(float* dst, int dp, const T* src, long long ld, int n, int d, bool vec_ok,
                                           const float* row_scale = nullptr) {
  constexpr int V = 16 / (int)sizeof(T);
  if (vec_ok) {
    const int vpr = d / V;
    for (int e = threadIdx.x; e < n * vpr; e += ANT) {
      const int r = e / vpr, c = (e - r * vpr) * V;
      const uint4 w = *reinterpret_cast<const uint4*>(src + (long long)r * ld + c);
      const float sc = row_scale != nullptr ? row_scale[r] : 1.0f;
      T tmp[V];
      *reinterpret_cast<uint4*>(tmp) = w;
#pragma unroll
      for (int j = 0; j < V; j += 4)
        *reinterpret_cast<float4*>(dst + r * dp + c + j) =
            make_float4((float)(T)((float)tmp[j] * sc), (float)(T)((float)tmp[j + 1] * sc), (float)(T)((float)tmp[j + 2] * sc), (float)(T)((float)tmp[j + 3] * sc));
    }
  } else {
    for (int e = threadIdx.x; e < n * d; e += ANT) {
      const int r = e / d, c = e - r * d;
      const float sc = row_scale != nullptr ? row_scale[r] : 1.0f;
      dst[r * dp + c] = (float)(T)((float)src[(long long)r * ld + c] * sc);      // (rounded to T like the tensor product it replaces)
    }
  }
}

__device__ __forceinline__ bool rows_aligned(const void* p, long long ld, int d, int esz) {
  const int V = 16 / esz;
  return (d % V) == 0 && (ld % V) == 0 && ((unsigned long long)p & 15) == 0;
}

__device__ __forceinline__ float dot4(const float4& x, const float4& y) { return x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w; }
__device__ __forceinline__ void fma4(float4& acc, float s, const float4& y) {
  acc.x += s * y.x; acc.y += s * y.y; acc.z += s * y.z; acc.w += s * y.w;
}
template <typename T>
__device__ __forceinline__ void store4t(T* p, const float4& v, float s) {
  p[0] = (T)(v.x * s); p[1] = (T)(v.y * s); p[2] = (T)(v.z * s); p[3] = (T)(v.w * s);
}

// LDS layout (both kernels, float32): Q [Nq][dp] | K [Nk][dp] | V [Nk][dp] | (dO [Nq][dp]) | S [Nq][sp] | (dS [Nq][sp]).  d is a multiple
// of 4: every inner loop reads 16-byte vectors along d (rows dp = d + 4 floats apart: 16-byte aligned, 4 banks further per row)
__host__ __device__ inline int pitch_d(int d) { return d + 4; }
__host__ __device__ inline int pitch_s(int Nk) { return Nk | 1; }

template <typename T>
__global__ __launch_bounds__(ANT) void attn_small_fwd_kernel(const AttnDev a) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  const int z = blockIdx.x, b = z / a.H, h = z - b * a.H;
  const int Nq = a.Nq, Nk = a.Nk, d = a.d, d4 = d >> 2;
  const int dp = pitch_d(d), sp = pitch_s(Nk);
  float* Qs = reinterpret_cast<float*>(lds_raw);
  float* Ks = Qs + Nq * dp;
  float* Vs = Ks + Nk * dp;
  float* S = Vs + Nk * dp;
  const T* q = reinterpret_cast<const T*>(a.q) + (long long)b * Nq * a.ldq + h * d;
  const int bk = a.kv_row != nullptr ? a.kv_row[b] : b;
  const T* k = reinterpret_cast<const T*>(a.k) + (long long)bk * Nk * a.ldk + h * d;
  const T* v = reinterpret_cast<const T*>(a.v) + (long long)bk * Nk * a.ldv + h * d;
  stage_rows<T>(Qs, dp, q, a.ldq, Nq, d, rows_aligned(q, a.ldq, d, sizeof(T)));
  const float* km = a.kmask != nullptr ? a.kmask + (long long)b * Nk : nullptr;
  stage_rows<T>(Ks, dp, k, a.ldk, Nk, d, rows_aligned(k, a.ldk, d, sizeof(T)), km);
  stage_rows<T>(Vs, dp, v, a.ldv, Nk, d, rows_aligned(v, a.ldv, d, sizeof(T)), km);
  __syncthreads();
  // scores: consecutive threads take consecutive keys of one query
  for (int e = threadIdx.x; e < Nq * Nk; e += ANT) {
    const int i = e / Nk, j = e - i * Nk;
    const float4* qr = reinterpret_cast<const float4*>(Qs + i * dp);
    const float4* kr = reinterpret_cast<const float4*>(Ks + j * dp);
    float acc = 0.f;
    for (int c = 0; c < d4; ++c) acc += dot4(qr[c], kr[c]);
    S[i * sp + j] = acc * a.scale;
  }
  __syncthreads();
  // softmax over the kept keys, one wave per row; P is rounded to T (what P V multiplies and what the backward pass reads)
  const bool causal = a.causal_b != nullptr ? a.causal_b[b] != 0 : a.causal != 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T* P = reinterpret_cast<T*>(a.p) + (long long)z * Nq * a.ldp;
  for (int i = wave; i < Nq; i += ANT / 64) {
    const int lim = causal ? min(Nk, i + (Nk - Nq) + 1) : Nk;       // keys j < lim are kept (blocks.py:315-319)
    float m = -3.0e38f;
    for (int j = lane; j < lim; j += 64) m = fmaxf(m, S[i * sp + j]);
    m = wave_max(m);
    float zs = 0.f;
    for (int j = lane; j < lim; j += 64) zs += expf(S[i * sp + j] - m);
    zs = wave_sum(zs);
    const float inv = 1.0f / zs;
    for (int j = lane; j < (int)a.ldp; j += 64) {
      const T pv = (T)(j < lim ? expf(S[i * sp + j] - m) * inv : 0.f);
      P[(long long)i * a.ldp + j] = pv;
      if (j < Nk) S[i * sp + j] = (float)pv;
    }
  }
  __syncthreads();
  // O = P V: consecutive threads take consecutive groups of 4 channels of one query
  T* O = reinterpret_cast<T*>(a.o) + (long long)b * Nq * a.ldo + h * d;
  for (int e = threadIdx.x; e < Nq * d4; e += ANT) {
    const int i = e / d4, c = (e - i * d4) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < Nk; ++j) fma4(acc, S[i * sp + j], *reinterpret_cast<const float4*>(Vs + j * dp + c));
    store4t<T>(O + (long long)i * a.ldo + c, acc, 1.0f);
  }
}

template <typename T>
__global__ __launch_bounds__(ANT) void attn_small_bwd_kernel(const AttnDev a) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  const int z = blockIdx.x, b = z / a.H, h = z - b * a.H;
  const int Nq = a.Nq, Nk = a.Nk, d = a.d, d4 = d >> 2;
  const int dp = pitch_d(d), sp = pitch_s(Nk);
  float* Qs = reinterpret_cast<float*>(lds_raw);
  float* Ks = Qs + Nq * dp;
  float* Vs = Ks + Nk * dp;
  float* Gs = Vs + Nk * dp;                               // dO
  float* Pf = Gs + Nq * dp;
  float* dS = Pf + Nq * sp;
  const T* q = reinterpret_cast<const T*>(a.q) + (long long)b * Nq * a.ldq + h * d;
  const int bk = a.kv_row != nullptr ? a.kv_row[b] : b;
  const T* k = reinterpret_cast<const T*>(a.k) + (long long)bk * Nk * a.ldk + h * d;
  const T* v = reinterpret_cast<const T*>(a.v) + (long long)bk * Nk * a.ldv + h * d;
  const T* g = reinterpret_cast<const T*>(a.d_o) + (long long)b * Nq * a.ldo + h * d;
  const T* P = reinterpret_cast<const T*>(a.p) + (long long)z * Nq * a.ldp;
  stage_rows<T>(Qs, dp, q, a.ldq, Nq, d, rows_aligned(q, a.ldq, d, sizeof(T)));
  const float* km = a.kmask != nullptr ? a.kmask + (long long)b * Nk : nullptr;
  stage_rows<T>(Ks, dp, k, a.ldk, Nk, d, rows_aligned(k, a.ldk, d, sizeof(T)), km);
  stage_rows<T>(Vs, dp, v, a.ldv, Nk, d, rows_aligned(v, a.ldv, d, sizeof(T)), km);
  stage_rows<T>(Gs, dp, g, a.ldo, Nq, d, rows_aligned(g, a.ldo, d, sizeof(T)));
  for (int e = threadIdx.x; e < Nq * Nk; e += ANT) {
    const int i = e / Nk, j = e - i * Nk;
    Pf[i * sp + j] = (float)P[(long long)i * a.ldp + j];
  }
  __syncthreads();
  // dP = dO V^T
  for (int e = threadIdx.x; e < Nq * Nk; e += ANT) {
    const int i = e / Nk, j = e - i * Nk;
    const float4* gr = reinterpret_cast<const float4*>(Gs + i * dp);
    const float4* vr = reinterpret_cast<const float4*>(Vs + j * dp);
    float acc = 0.f;
    for (int c = 0; c < d4; ++c) acc += dot4(gr[c], vr[c]);
    dS[i * sp + j] = acc;
  }
  __syncthreads();
  // dS = P (dP - sum_j dP P), one wave per row
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = wave; i < Nq; i += ANT / 64) {
    float t = 0.f;
    for (int j = lane; j < Nk; j += 64) t += Pf[i * sp + j] * dS[i * sp + j];
    t = wave_sum(t);
    for (int j = lane; j < Nk; j += 64) dS[i * sp + j] = Pf[i * sp + j] * (dS[i * sp + j] - t);
  }
  __syncthreads();
  // dK = scale dS^T Q, dV = P^T dO (the long loop first: Nk d / 4 items), then dQ = scale dS K
  T* dK = reinterpret_cast<T*>(a.dk) + (long long)b * Nk * a.lddk + h * d;
  T* dV = reinterpret_cast<T*>(a.dv) + (long long)b * Nk * a.lddv + h * d;
  for (int e = threadIdx.x; e < Nk * d4; e += ANT) {
    const int j = e / d4, c = (e - j * d4) * 4;
    float4 ak = make_float4(0.f, 0.f, 0.f, 0.f), av = ak;
    for (int i = 0; i < Nq; ++i) {
      fma4(ak, dS[i * sp + j], *reinterpret_cast<const float4*>(Qs + i * dp + c));
      fma4(av, Pf[i * sp + j], *reinterpret_cast<const float4*>(Gs + i * dp + c));
    }
    const float mj = km != nullptr ? km[j] : 1.0f;         // d(k mask) / dk = mask
    store4t<T>(dK + (long long)j * a.lddk + c, ak, a.scale * mj);
    store4t<T>(dV + (long long)j * a.lddv + c, av, mj);
  }
  T* dQ = reinterpret_cast<T*>(a.dq) + (long long)b * Nq * a.lddq + h * d;
  for (int e = threadIdx.x; e < Nq * d4; e += ANT) {
    const int i = e / d4, c = (e - i * d4) * 4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < Nk; ++j) fma4(acc, dS[i * sp + j], *reinterpret_cast<const float4*>(Ks + j * dp + c));
    store4t<T>(dQ + (long long)i * a.lddq + c, acc, a.scale);
  }
}

size_t fwd_lds(int Nq, int Nk, int d) { return ((size_t)(Nq + 2 * Nk) * pitch_d(d) + (size_t)Nq * pitch_s(Nk)) * 4; }
size_t bwd_lds(int Nq, int Nk, int d) { return ((size_t)(2 * Nq + 2 * Nk) * pitch_d(d) + (size_t)2 * Nq * pitch_s(Nk)) * 4; }
constexpr size_t LDS_MAX = 160 * 1024;

int check_common(const char* who, int B, int H, int Nq, int Nk, int d, int dtype) {
  JEN1_CHECK(dtype == JEN1_F32 || dtype == JEN1_BF16, "%s: dtype must be JEN1_F32 or JEN1_BF16", who);
  JEN1_CHECK(B >= 1 && H >= 1 && Nq >= 1 && Nk >= 1 && d >= 1, "%s: B, H, Nq, Nk, d must be >= 1", who);
  JEN1_CHECK((long long)B * H <= 0x7fffffff, "%s: too many (batch element, head) pairs", who);
  return 0;
}

}  // namespace

extern "C" int jen1_attn_small_fits(int Nq, int Nk, int d, int dtype) {
  (void)dtype;
  return (Nq <= 64 && Nk <= 512 && d <= 128 && d % 4 == 0 && bwd_lds(Nq, Nk, d) <= LDS_MAX) ? 1 : 0;
}

extern "C" int jen1_attn_small_forward(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o,
                                       int64_t ldo, void* p, int64_t ldp, int B, int H, int Nq, int Nk, int d, float scale, int causal,
                                       const int32_t* causal_b, const float* kv_mask, int dtype, void* stream) {
  return jen1_attn_small_forward_rows(q, ldq, k, ldk, v, ldv, o, ldo, p, ldp, B, H, Nq, Nk, d, scale, causal, causal_b, kv_mask, nullptr, dtype, stream);
}

extern "C" int jen1_attn_small_forward_rows(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o,
                                            int64_t ldo, void* p, int64_t ldp, int B, int H, int Nq, int Nk, int d, float scale, int causal,
                                            const int32_t* causal_b, const float* kv_mask, const int32_t* kv_row, int dtype, void* stream) {
  if (check_common("jen1_attn_small_forward", B, H, Nq, Nk, d, dtype)) return 1;
  JEN1_CHECK(q && k && v && o && p, "jen1_attn_small_forward: NULL argument");
  JEN1_CHECK(ldp >= Nk, "jen1_attn_small_forward: ldp must be >= Nk");
  JEN1_CHECK(jen1_attn_small_fits(Nq, Nk, d, dtype), "jen1_attn_small_forward: Nq = %d, Nk = %d, d = %d do not fit one workgroup", Nq, Nk, d);
  AttnDev a = {};
  a.q = q; a.k = k; a.v = v; a.o = o; a.p = p;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.ldp = ldp;
  a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.d = d; a.causal = causal ? 1 : 0; a.scale = scale;
  a.causal_b = reinterpret_cast<const int*>(causal_b);
  a.kmask = kv_mask;
  a.kv_row = reinterpret_cast<const int*>(kv_row);
  const size_t lds = fwd_lds(Nq, Nk, d);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == JEN1_F32) {
    JEN1_MAX_LDS_ONCE(attn_small_fwd_kernel<float>, (int)LDS_MAX);
    hipLaunchKernelGGL(attn_small_fwd_kernel<float>, dim3(B * H), dim3(ANT), lds, s, a);
  } else {
    JEN1_MAX_LDS_ONCE(attn_small_fwd_kernel<bf16_t>, (int)LDS_MAX);
    hipLaunchKernelGGL(attn_small_fwd_kernel<bf16_t>, dim3(B * H), dim3(ANT), lds, s, a);
  }
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_attn_small_backward(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* p,
                                        int64_t ldp, const void* d_o, int64_t ldo, void* dq, int64_t lddq, void* dk, int64_t lddk,
                                        void* dv, int64_t lddv, int B, int H, int Nq, int Nk, int d, float scale, const float* kv_mask,
                                        int dtype, void* stream) {
  return jen1_attn_small_backward_rows(q, ldq, k, ldk, v, ldv, p, ldp, d_o, ldo, dq, lddq, dk, lddk, dv, lddv, B, H, Nq, Nk, d, scale, kv_mask, nullptr,
                                       dtype, stream);
}

extern "C" int jen1_attn_small_backward_rows(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* p,
                                             int64_t ldp, const void* d_o, int64_t ldo, void* dq, int64_t lddq, void* dk, int64_t lddk,
                                             void* dv, int64_t lddv, int B, int H, int Nq, int Nk, int d, float scale, const float* kv_mask,
                                             const int32_t* kv_row, int dtype, void* stream) {
  if (check_common("jen1_attn_small_backward", B, H, Nq, Nk, d, dtype)) return 1;
  JEN1_CHECK(q && k && v && p && d_o && dq && dk && dv, "jen1_attn_small_backward: NULL argument");
  JEN1_CHECK(ldp >= Nk, "jen1_attn_small_backward: ldp must be >= Nk");
  JEN1_CHECK(jen1_attn_small_fits(Nq, Nk, d, dtype), "jen1_attn_small_backward: Nq = %d, Nk = %d, d = %d do not fit one workgroup", Nq, Nk, d);
  AttnDev a = {};
  a.q = q; a.k = k; a.v = v; a.p = const_cast<void*>(p); a.d_o = d_o; a.dq = dq; a.dk = dk; a.dv = dv;
  a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.ldp = ldp; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
  a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.d = d; a.scale = scale;
  a.kmask = kv_mask;
  a.kv_row = reinterpret_cast<const int*>(kv_row);
  const size_t lds = bwd_lds(Nq, Nk, d);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == JEN1_F32) {
    JEN1_MAX_LDS_ONCE(attn_small_bwd_kernel<float>, (int)LDS_MAX);
    hipLaunchKernelGGL(attn_small_bwd_kernel<float>, dim3(B * H), dim3(ANT), lds, s, a);
  } else {
    JEN1_MAX_LDS_ONCE(attn_small_bwd_kernel<bf16_t>, (int)LDS_MAX);
    hipLaunchKernelGGL(attn_small_bwd_kernel<bf16_t>, dim3(B * H), dim3(ANT), lds, s, a);
  }
  JEN1_HIP(hipGetLastError());
  return 0;
}
