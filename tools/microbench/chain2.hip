// Can dependent kernels overlap their launch + weight prefetch with the predecessor when the dependency is
// enforced by an in-kernel spin on a device counter instead of stream order?  (MI355X, ROCm 7.2)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int NWG = 64;

// op i: prefetch "weights", wait for op i-1 (if spin), read act[i-1], write act[i], signal done[i]
template <bool SPIN>
__global__ __launch_bounds__(256) void op_kernel(int i, const float4* __restrict__ w, float* act, unsigned* done, unsigned* err) {
  // ---- independent prologue: 32 KB of weights per workgroup into registers
  const float4* wp = w + ((size_t)(i % 16) * NWG + blockIdx.x) * 2048 + threadIdx.x;
  float4 r[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) r[u] = wp[u * 256];
  // ---- dependency
  if (SPIN && i > 0) {
    if (threadIdx.x < 64) {
      const unsigned* d = done + (size_t)(i - 1) * 8;
      int spins = 0;
      bool ok = false;
      while (!ok) {
        unsigned v = (threadIdx.x < 8) ? __hip_atomic_load(d + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : NWG / 8;
        ok = __all(v >= NWG / 8);
        if (!ok) { __builtin_amdgcn_s_sleep(2); if (++spins > (1 << 20)) { if (threadIdx.x == 0) atomicAdd(err, 1u); break; } }
      }
    }
    __syncthreads();
  }
  // ---- dependent part: read the predecessor's output (sc1), combine, write (sc1)
  float in = 0.f;
  if (i > 0) in = __hip_atomic_load(act + (size_t)(i - 1) * 16384 + (blockIdx.x * 256 + threadIdx.x + 4096) % 16384, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < 8; ++u) s += r[u].x + r[u].y + r[u].z + r[u].w;
  const float out = in + 1.0f + s * 0.f;
  __hip_atomic_store(act + (size_t)i * 16384 + blockIdx.x * 256 + threadIdx.x, out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(done + (size_t)i * 8 + (blockIdx.x & 7), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int main() {
  const int N = 400;
  float4* w; float* act; unsigned *done, *err;
  CK(hipMalloc(&w, (size_t)16 * NWG * 2048 * 16)); CK(hipMemset(w, 0, (size_t)16 * NWG * 2048 * 16));
  CK(hipMalloc(&act, (size_t)N * 16384 * 4)); CK(hipMalloc(&done, (size_t)N * 8 * 4)); CK(hipMalloc(&err, 4));
  hipStream_t st[4];
  for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  auto reset = [&]() { CK(hipMemset(done, 0, (size_t)N * 8 * 4)); CK(hipMemset(err, 0, 4)); CK(hipMemset(act, 0, (size_t)N * 16384 * 4)); CK(hipDeviceSynchronize()); };
  auto check = [&](const char* name, double us) {
    float v; unsigned e;
    CK(hipMemcpy(&v, act + (size_t)(N - 1) * 16384 + 5, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost));
    printf("%-46s: %.2f us per op   (final value %.0f, expect %d; spin timeouts %u)\n", name, us / N, v, N, e);
  };
  auto now = []() { return std::chrono::high_resolution_clock::now(); };
  auto us_since = [&](auto t0) { return std::chrono::duration<double, std::micro>(now() - t0).count(); };

  // independent chains: k graphs (each a full stream-ordered chain over its own buffers) on k streams
  float* act2[4]; unsigned* done2[4];
  for (int j = 0; j < 4; ++j) { CK(hipMalloc(&act2[j], (size_t)N * 16384 * 4)); CK(hipMalloc(&done2[j], (size_t)N * 8 * 4)); CK(hipMemset(act2[j], 0, (size_t)N * 16384 * 4)); CK(hipMemset(done2[j], 0, (size_t)N * 8 * 4)); }
  hipGraph_t g[4]; hipGraphExec_t ge[4];
  for (int j = 0; j < 4; ++j) {
    CK(hipStreamBeginCapture(st[j], hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(op_kernel<false>, dim3(NWG), dim3(256), 0, st[j], i, w, act2[j], done2[j], err);
    CK(hipStreamEndCapture(st[j], &g[j])); CK(hipGraphInstantiate(&ge[j], g[j], nullptr, nullptr, 0));
  }
  for (int k : {1, 2, 3, 4}) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      auto t0 = now();
      for (int j = 0; j < k; ++j) CK(hipGraphLaunch(ge[j], st[j]));
      for (int j = 0; j < k; ++j) CK(hipStreamSynchronize(st[j]));
      double us = us_since(t0);
      if (rep == 2) printf("%d independent chains of %d ops on %d streams: %.1f us total, %.2f us per op per chain, %.2f us per op aggregate\n", k, N, k, us, us / N, us / N / k);
    }
  }
  return 0;
}
