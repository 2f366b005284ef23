// cross-XCD visibility without fences: which load/store flavours are coherent inside one persistent kernel?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned nwg, unsigned& epoch) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    ++epoch;
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = epoch * nwg;
    int spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) { __builtin_amdgcn_s_sleep(1); if (++spins > (1 << 22)) break; }
  }
  __syncthreads();
}

template <bool SC_ST, bool SC_LD, bool VEC>
__global__ __launch_bounds__(256) void persist(unsigned* ctr, float* buf, int iters, unsigned* errs) {
  unsigned epoch = 0;
  const unsigned nwg = gridDim.x;
  unsigned bad = 0;
  for (int it = 0; it < iters; ++it) {
    float* cur = buf + (size_t)(it & 1) * nwg * 1024;
    const float val = (float)(it * 1000 + (int)blockIdx.x);
    float* mine = cur + blockIdx.x * 1024 + threadIdx.x * 4;
    if (SC_ST) {
      typedef unsigned long long u64;
      const u64 w = ((u64)__float_as_uint(val) << 32) | __float_as_uint(val);
      __hip_atomic_store(reinterpret_cast<u64*>(mine), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(reinterpret_cast<u64*>(mine) + 1, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      *reinterpret_cast<float4*>(mine) = make_float4(val, val, val, val);
    }
    grid_barrier(ctr, nwg, epoch);
    for (int k = 1; k <= 3; k += 2) {
      const int nb = (blockIdx.x + k) % nwg;     // +1: next XCD, +3: another XCD
      const float* theirs = cur + nb * 1024 + threadIdx.x * 4;
      float4 v;
      if (SC_LD) {
        typedef unsigned long long u64;
        const u64 a = __hip_atomic_load(reinterpret_cast<const u64*>(theirs), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u64 b = __hip_atomic_load(reinterpret_cast<const u64*>(theirs) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v = make_float4(__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned)b), __uint_as_float((unsigned)(b >> 32)));
      } else {
        v = *reinterpret_cast<const float4*>(theirs);
      }
      const float want = (float)(it * 1000 + nb);
      if (v.x != want || v.y != want || v.z != want || v.w != want) ++bad;
    }
  }
  if (bad) atomicAdd(errs, bad);
}

int main() {
  unsigned *ctr, *errs;
  float *buf_n, *buf_u = nullptr, *buf_f = nullptr;
  const int nwg = 256, iters = 2000;
  const size_t bytes = (size_t)2 * nwg * 1024 * 4;
  CK(hipMalloc(&ctr, 4096)); CK(hipMalloc(&errs, 4096)); CK(hipMalloc(&buf_n, bytes));
  hipError_t eu = hipExtMallocWithFlags((void**)&buf_u, bytes, hipDeviceMallocUncached);
  printf("hipDeviceMallocUncached: %s\n", hipGetErrorString(eu));
  hipError_t ef = hipExtMallocWithFlags((void**)&buf_f, bytes, hipDeviceMallocFinegrained);
  printf("hipDeviceMallocFinegrained: %s\n", hipGetErrorString(ef));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct { const char* name; float* buf; int st, ld; } modes[] = {
    {"normal mem, plain st / plain ld", buf_n, 0, 0}, {"normal mem, sc1 st / sc1 ld", buf_n, 1, 1},
    {"normal mem, plain st / sc1 ld", buf_n, 0, 1},  {"normal mem, sc1 st / plain ld", buf_n, 1, 0},
    {"uncached mem, plain st / plain ld", buf_u, 0, 0}, {"finegrained mem, plain st / plain ld", buf_f, 0, 0},
    {"uncached mem, sc1 st / sc1 ld", buf_u, 1, 1},
  };
  for (auto& m : modes) {
    if (!m.buf) continue;
    float best = 1e9; unsigned e_tot = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(ctr, 0, 4096)); CK(hipMemset(errs, 0, 4096)); CK(hipMemset(m.buf, 0, bytes));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      if (m.st && m.ld) hipLaunchKernelGGL((persist<true, true, false>), dim3(nwg), dim3(256), 0, 0, ctr, m.buf, iters, errs);
      else if (!m.st && m.ld) hipLaunchKernelGGL((persist<false, true, false>), dim3(nwg), dim3(256), 0, 0, ctr, m.buf, iters, errs);
      else if (m.st && !m.ld) hipLaunchKernelGGL((persist<true, false, false>), dim3(nwg), dim3(256), 0, 0, ctr, m.buf, iters, errs);
      else hipLaunchKernelGGL((persist<false, false, false>), dim3(nwg), dim3(256), 0, 0, ctr, m.buf, iters, errs);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
      unsigned e; CK(hipMemcpy(&e, errs, 4, hipMemcpyDeviceToHost)); e_tot += e;
    }
    printf("%-40s: %.3f us / iter, mismatching reads = %u\n", m.name, best * 1000.f / iters, e_tot);
  }
  return 0;
}
