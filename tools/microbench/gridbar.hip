// microbenchmark: cost of a grid-wide barrier in a persistent kernel on MI355X
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

template <int MODE>   // 0: flat, fenced; 1: flat, no fences; 2: hierarchical fenced; 3: hierarchical no fences
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned* xctr, unsigned nwg, unsigned nx, unsigned xcc, unsigned& epoch) {
  __syncthreads();
  if (threadIdx.x == 0) {
    ++epoch;
    if (MODE == 0 || MODE == 2) __atomic_thread_fence(__ATOMIC_RELEASE);   // agent... (hip: system scope by default)
    if (MODE < 2) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = epoch * nwg;
      int spins = 0;
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) { __builtin_amdgcn_s_sleep(1); if (++spins > (1 << 22)) break; }
    } else {
      const unsigned old = __hip_atomic_fetch_add(xctr + xcc * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old == epoch * nx - 1) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = epoch * 8;
      int spins = 0;
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) { __builtin_amdgcn_s_sleep(1); if (++spins > (1 << 22)) break; }
    }
    if (MODE == 0 || MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

template <int MODE>
__global__ __launch_bounds__(256) void persist(unsigned* ctr, unsigned* xctr, float* buf, int iters, unsigned* xcc_out) {
  unsigned epoch = 0;
  const unsigned nwg = gridDim.x;
  const unsigned xcc = xcc_id();
  if (threadIdx.x == 0) xcc_out[blockIdx.x] = xcc;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    // a little dependent work: read what the neighbour WG wrote last iteration, write own
    const int nb = (blockIdx.x + 1) % nwg;
    acc += __builtin_nontemporal_load(buf + nb * 256 + threadIdx.x);
    __hip_atomic_store(buf + blockIdx.x * 256 + threadIdx.x, acc + 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    grid_barrier<MODE>(ctr, xctr, nwg, nwg / 8, xcc, epoch);
  }
  if (threadIdx.x == 0 && acc == -1.f) buf[0] = acc;
}

__global__ void tiny(float* buf) { if (threadIdx.x == 0 && blockIdx.x == 0) buf[0] += 1.f; }

int main() {
  unsigned *ctr, *xctr, *xcc_out;
  float* buf;
  CK(hipMalloc(&ctr, 4096)); CK(hipMalloc(&xctr, 4096 * 4)); CK(hipMalloc(&buf, 1024 * 256 * 4)); CK(hipMalloc(&xcc_out, 4096));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000;
  for (int nwg : {64, 128, 256, 512}) {
    for (int mode = 0; mode < 4; ++mode) {
      float best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(ctr, 0, 4096)); CK(hipMemset(xctr, 0, 4096 * 4)); CK(hipMemset(buf, 0, 1024 * 256 * 4));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        switch (mode) {
          case 0: hipLaunchKernelGGL(persist<0>, dim3(nwg), dim3(256), 0, 0, ctr, xctr, buf, iters, xcc_out); break;
          case 1: hipLaunchKernelGGL(persist<1>, dim3(nwg), dim3(256), 0, 0, ctr, xctr, buf, iters, xcc_out); break;
          case 2: hipLaunchKernelGGL(persist<2>, dim3(nwg), dim3(256), 0, 0, ctr, xctr, buf, iters, xcc_out); break;
          case 3: hipLaunchKernelGGL(persist<3>, dim3(nwg), dim3(256), 0, 0, ctr, xctr, buf, iters, xcc_out); break;
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      unsigned c; CK(hipMemcpy(&c, ctr, 4, hipMemcpyDeviceToHost));
      printf("nwg=%3d mode=%d (%s, %s): %.3f us per barrier+op   (ctr=%u)\n", nwg, mode, mode < 2 ? "flat" : "hier", (mode & 1) ? "no fence" : "fenced", best * 1000.f / iters, c);
    }
  }
  std::vector<unsigned> x(256);
  CK(hipMemcpy(x.data(), xcc_out, 256 * 4, hipMemcpyDeviceToHost));
  printf("xcc of wg 0..15:"); for (int i = 0; i < 16; ++i) printf(" %u", x[i]); printf("\n");
  // kernel-boundary reference: 2000 tiny dependent launches
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, 0, buf);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("stream launches of a tiny kernel: %.3f us each\n", ms * 1000.f / iters);
  return 0;
}
