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

#define JEN1_CHECK(cond, ...)                  \
  do {                                         \
    if (!(cond)) return jen1_set_error(__VA_ARGS__); \
  } while (0)

#define JEN1_HIP(call)                                                                     \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) return jen1_set_error("%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)

// ---- 8-element fragments -------------------------------------------------------------------
struct f32x8 {
  float v[8];
};

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

template <typename T>
struct is_f32 {
  static constexpr bool value = false;
};
template <>
struct is_f32<float> {
  static constexpr bool value = true;
};
