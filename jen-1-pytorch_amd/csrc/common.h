// Shared device/host helpers for libjen1_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "jen1_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define JEN1_WAVE 64
#define JEN1_FINE_GROUPS 32   // GroupNorm statistics are kept per 1/32 of the channel range

extern thread_local char g_jen1_err[512];
int jen1_set_error(const char* fmt, ...);

// stream_gemm.hip: launcher of jen1_conv_gemm's direct (weight-streaming) mode
int jen1_stream_gemm_launch(const jen1_conv_args& a, void* stream);
// tile_gemm.hip: launcher of the JEN1_CFG_T* tile configurations
int jen1_tile_gemm_launch(const jen1_conv_args& a, void* stream);

#define JEN1_CHECK(cond, ...)                  \
  do {                                         \
    if (!(cond)) return jen1_set_error(__VA_ARGS__); \
  } while (0)

#define JEN1_HIP(call)                                                                     \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) return jen1_set_error("%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: set once per (kernel, device ordinal).  ``mask``
// is a function-local static of the caller (one bit per device, ordinals >= 64 set it every time); launches are issued from one
// host thread per process in this library, so the mask needs no lock.
#define JEN1_MAX_LDS_ONCE(kern, bytes)                                                                      \
  do {                                                                                                      \
    static unsigned long long lds_mask_ = 0ull;                                                             \
    int dev_ = 0;                                                                                           \
    JEN1_HIP(hipGetDevice(&dev_));                                                                          \
    if (dev_ >= 64 || !((lds_mask_ >> dev_) & 1ull)) {                                                      \
      JEN1_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (bytes))); \
      if (dev_ < 64) lds_mask_ |= 1ull << dev_;                                                             \
    }                                                                                                       \
  } while (0)

// ---- 8-element fragments -------------------------------------------------------------------
struct f32x8 {
  float v[8];
};

// sum over the 64 lanes of a wave, every lane gets it: DPP row steps and the two v_permlane swaps of gfx950 -- no LDS crossbar
// (six ds_bpermute steps of __shfl_xor cost ~0.1 us each on a dependent chain)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<0xB1>(v);                    // lanes ^ 1
  v += dpp_mov<0x4E>(v);                    // lanes ^ 2
  v += dpp_mov<0x141>(v);                   // row_half_mirror
  v += dpp_mov<0x140>(v);                   // row_mirror
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);      // lanes ^ 16
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);   // lanes ^ 32
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ void load8(const float* p, float (&o)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
  o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&o)[8]) {
  const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = (float)a[i];
}
__device__ __forceinline__ void store8(float* p, const float (&o)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(o[4], o[5], o[6], o[7]);
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&o)[8]) {
  bf16x8 a;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (bf16_t)o[i];
  *reinterpret_cast<bf16x8*>(p) = a;
}
__device__ __forceinline__ void load4(const float* p, float (&o)[4]) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
}
__device__ __forceinline__ void load4(const bf16_t* p, float (&o)[4]) {
  const bf16x4 a = *reinterpret_cast<const bf16x4*>(p);
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (float)a[i];
}
__device__ __forceinline__ void store4(float* p, const float (&o)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
}
__device__ __forceinline__ void store4(bf16_t* p, const float (&o)[4]) {
  bf16x4 a;
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = (bf16_t)o[i];
  *reinterpret_cast<bf16x4*>(p) = a;
}

// ---- JEN1_FP8: OCP e4m3 operands of the matrix cores (gfx950's fp8; v_mfma_f32_16x16x32_fp8_fp8 reads 8 bytes of K per lane and
// operand with the same lane -> element map as the bf16 form).  Activations stay bf16 in memory; what is staged in LDS for the
// matrix cores -- activation tiles, Q / K / P / V^T -- is one byte per element.
struct fp8_t {
  unsigned char v;
};
static_assert(sizeof(fp8_t) == 1, "one byte per element");
#define JEN1_FP8_MAX 448.0f
#define JEN1_FP8_P_SCALE 256.0f      // softmax probabilities (<= 1) are stored as 256 p: 1 / Nk would be an e4m3 denormal
// two floats -> two e4m3 bytes in the low / high half of `old` (saturating: |x| > 448 would turn into NaN)
__device__ __forceinline__ unsigned jen1_pk_fp8(float a, float b, unsigned old, bool hi) {
  a = __builtin_amdgcn_fmed3f(a, -JEN1_FP8_MAX, JEN1_FP8_MAX);
  b = __builtin_amdgcn_fmed3f(b, -JEN1_FP8_MAX, JEN1_FP8_MAX);
  return hi ? (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)old, true) : (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)old, false);
}
__device__ __forceinline__ void store8(fp8_t* p, const float (&o)[8]) {
  unsigned lo = 0, hi = 0;
  lo = jen1_pk_fp8(o[0], o[1], lo, false);
  lo = jen1_pk_fp8(o[2], o[3], lo, true);
  hi = jen1_pk_fp8(o[4], o[5], hi, false);
  hi = jen1_pk_fp8(o[6], o[7], hi, true);
  *reinterpret_cast<uint2*>(p) = make_uint2(lo, hi);
}
template <typename T> __device__ __forceinline__ T to_elem(float x) { return (T)x; }
template <> __device__ __forceinline__ fp8_t to_elem<fp8_t>(float x) {
  fp8_t r;
  r.v = (unsigned char)(jen1_pk_fp8(x, x, 0u, false) & 0xffu);
  return r;
}

template <typename T>
struct VecOf;
template <>
struct VecOf<float> {
  typedef f32x8 type;
};
template <>
struct VecOf<bf16_t> {
  typedef bf16x8 type;
};
__device__ __forceinline__ void vec_to_float(const f32x8& f, float (&o)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = f.v[j];
}
__device__ __forceinline__ void vec_to_float(const bf16x8& f, float (&o)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (float)f[j];
}

__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// precise variant for the float32 parity mode (expf, IEEE division)
__device__ __forceinline__ float silu_precise(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// Touch the kernel-argument segment with one batch of scalar loads (one per 64-byte line, results discarded;
// never past the last line that holds an explicit argument).  hipcc sinks kernarg reads next to their uses,
// often behind uniform branches; on a cold scalar cache each of those is a serial memory round trip.  After
// this batch they all hit.
template <int NBYTES>
__device__ __forceinline__ void jen1_prefetch_kernarg() {
  const auto kp = __builtin_amdgcn_kernarg_segment_ptr();
  constexpr int FULL = (NBYTES + 63) / 64;
  constexpr int LINES = FULL >= 16 ? 16 : FULL >= 12 ? 12 : FULL >= 8 ? 8 : FULL >= 4 ? 4 : FULL >= 2 ? 2 : 0;
  unsigned t[16];
  if constexpr (LINES == 2) {
    asm volatile("s_load_dword %0, %2, 0x0\n\ts_load_dword %1, %2, 0x40\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(t[0]), "=&s"(t[1])
                 : "s"(kp)
                 : "memory");
  }
  if constexpr (LINES == 4) {
    asm volatile("s_load_dword %0, %4, 0x0\n\ts_load_dword %1, %4, 0x40\n\ts_load_dword %2, %4, 0x80\n\ts_load_dword %3, %4, 0xc0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(t[0]), "=&s"(t[1]), "=&s"(t[2]), "=&s"(t[3])
                 : "s"(kp)
                 : "memory");
  }
  if constexpr (LINES == 8) {
    asm volatile("s_load_dword %0, %8, 0x0\n\ts_load_dword %1, %8, 0x40\n\ts_load_dword %2, %8, 0x80\n\ts_load_dword %3, %8, 0xc0\n\ts_load_dword %4, %8, 0x100\n\ts_load_dword %5, %8, 0x140\n\ts_load_dword %6, %8, 0x180\n\ts_load_dword %7, %8, 0x1c0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(t[0]), "=&s"(t[1]), "=&s"(t[2]), "=&s"(t[3]), "=&s"(t[4]), "=&s"(t[5]), "=&s"(t[6]), "=&s"(t[7])
                 : "s"(kp)
                 : "memory");
  }
  if constexpr (LINES == 12) {
    asm volatile("s_load_dword %0, %12, 0x0\n\ts_load_dword %1, %12, 0x40\n\ts_load_dword %2, %12, 0x80\n\ts_load_dword %3, %12, 0xc0\n\ts_load_dword %4, %12, 0x100\n\ts_load_dword %5, %12, 0x140\n\ts_load_dword %6, %12, 0x180\n\ts_load_dword %7, %12, 0x1c0\n\ts_load_dword %8, %12, 0x200\n\ts_load_dword %9, %12, 0x240\n\ts_load_dword %10, %12, 0x280\n\ts_load_dword %11, %12, 0x2c0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(t[0]), "=&s"(t[1]), "=&s"(t[2]), "=&s"(t[3]), "=&s"(t[4]), "=&s"(t[5]), "=&s"(t[6]), "=&s"(t[7]), "=&s"(t[8]), "=&s"(t[9]), "=&s"(t[10]), "=&s"(t[11])
                 : "s"(kp)
                 : "memory");
  }
  if constexpr (LINES == 16) {
    asm volatile("s_load_dword %0, %16, 0x0\n\ts_load_dword %1, %16, 0x40\n\ts_load_dword %2, %16, 0x80\n\ts_load_dword %3, %16, 0xc0\n\ts_load_dword %4, %16, 0x100\n\ts_load_dword %5, %16, 0x140\n\ts_load_dword %6, %16, 0x180\n\ts_load_dword %7, %16, 0x1c0\n\ts_load_dword %8, %16, 0x200\n\ts_load_dword %9, %16, 0x240\n\ts_load_dword %10, %16, 0x280\n\ts_load_dword %11, %16, 0x2c0\n\ts_load_dword %12, %16, 0x300\n\ts_load_dword %13, %16, 0x340\n\ts_load_dword %14, %16, 0x380\n\ts_load_dword %15, %16, 0x3c0\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(t[0]), "=&s"(t[1]), "=&s"(t[2]), "=&s"(t[3]), "=&s"(t[4]), "=&s"(t[5]), "=&s"(t[6]), "=&s"(t[7]), "=&s"(t[8]), "=&s"(t[9]), "=&s"(t[10]), "=&s"(t[11]), "=&s"(t[12]), "=&s"(t[13]), "=&s"(t[14]), "=&s"(t[15])
                 : "s"(kp)
                 : "memory");
  }
}

template <typename T>
struct is_f32 {
  static constexpr bool value = false;
};
template <>
struct is_f32<float> {
  static constexpr bool value = true;
};
