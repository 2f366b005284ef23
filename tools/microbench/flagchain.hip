// microbenchmark: latency floor of one all-to-all exchange stage inside a persistent kernel on MI355X, two protocols
//   A  counters   payload with write-through (sc1) stores -> every storing wave drains -> one relaxed agent-scope atomic per
//                 workgroup on a sharded counter -> wave 0 polls the shards -> workgroup barrier -> sc1 loads of the payload
//                 (what csrc/deep_kernel.hip did up to round 2)
//   B  sentinel   the payload buffers are poisoned (all ones) before the launch; producers store 8-byte words atomically and do
//                 nothing else; every consumer WAVE re-loads its own vectors until no word is the sentinel (the data load IS the
//                 wait; no drain, no atomic, no barrier on the way)
// Every stage is a true dependency: what a workgroup stores in stage p is a function of everything it loaded in stage p - 1.
// hipcc --offload-arch=gfx950 -O3 flagchain.hip -o flagchain && ./flagchain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;
#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr int NT = 512, SHARDS = 32, SHW = 32;
constexpr int WPW = 16;            // 8-byte words a workgroup produces per stage (128 bytes: one 16-row M tile x 8 columns of bf16)
#ifndef VEC_N
#define VEC_N 4
#endif
constexpr int VEC = VEC_N;             // 16-byte vectors a thread consumes per stage

template <int MODE>
__global__ __launch_bounds__(NT) void chain(u64* buf, unsigned* ctr, int stages, int sleep_clk, u64* out) {
  const int tid = threadIdx.x, wg = blockIdx.x, nwg = gridDim.x;
  const int words = nwg * WPW;                       // payload of a stage
  u64 carry = 1;
  for (int p = 0; p < stages; ++p) {
    u64* cur = buf + (size_t)p * words;
    // ---- produce ----
    if (tid < WPW) __hip_atomic_store((gu64*)(u64)(cur + wg * WPW + tid), (carry & 0xffffffffull) | ((u64)(p + 1) << 32), RLX);
    if (MODE == 0) {
      if (tid < 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add((gu32*)(u64)(ctr + ((size_t)p * SHARDS + wg % SHARDS) * SHW), 1u, RLX);
      if (tid < 64) {
        for (int spins = 0; spins < (1 << 22); ++spins) {
          unsigned v = tid < SHARDS ? __hip_atomic_load((gu32*)(u64)(ctr + ((size_t)p * SHARDS + tid) * SHW), RLX) : 0u;
          for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
          if ((int)v >= nwg) break;
          __builtin_amdgcn_s_sleep(8);
        }
      }
      __syncthreads();
    }
    // ---- consume: VEC vectors of 2 words per thread, strided over the whole payload ----
    u64 w[VEC][2];
    for (int spins = 0;; ++spins) {
      bool bad = false;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const int idx = ((tid + i * NT) * 2) % words;
        w[i][0] = __hip_atomic_load((gu64*)(u64)(cur + idx), RLX);
        w[i][1] = __hip_atomic_load((gu64*)(u64)(cur + idx + 1), RLX);
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) bad |= (w[i][0] == ~0ull) | (w[i][1] == ~0ull);
      if (MODE == 0 || !__builtin_amdgcn_ballot_w64(bad) || spins > (1 << 20)) break;
      if (sleep_clk == 1) __builtin_amdgcn_s_sleep(2); else if (sleep_clk == 2) __builtin_amdgcn_s_sleep(8); else if (sleep_clk == 3) __builtin_amdgcn_s_sleep(32);
    }
    u64 s = 0;
#pragma unroll
    for (int i = 0; i < VEC; ++i) s += (w[i][0] & 0xffff) + (w[i][1] & 0xffff);
    carry = (carry + s) & 0xffff;
    if (MODE == 1) __syncthreads();                  // (the real unit has one barrier per stage behind its staging anyway)
  }
  if (tid == 0) out[wg] = carry;
}

int main() {
  const int stages = 400, maxwg = 256;
  u64 *buf, *out;
  unsigned* ctr;
  const size_t bbytes = (size_t)stages * maxwg * WPW * 8;
  CK(hipMalloc(&buf, bbytes)); CK(hipMalloc(&out, maxwg * 8)); CK(hipMalloc(&ctr, (size_t)stages * SHARDS * SHW * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int nwg : {64, 256}) {
    for (int mode = 0; mode < 5; ++mode) {
      float best = 1e9;
      for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemset(buf, 0xFF, bbytes)); CK(hipMemset(ctr, 0, (size_t)stages * SHARDS * SHW * 4));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(nwg), dim3(NT), 0, 0, buf, ctr, stages, 1, out);
        else hipLaunchKernelGGL(chain<1>, dim3(nwg), dim3(NT), 0, 0, buf, ctr, stages, mode - 1, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      u64 o[2]; CK(hipMemcpy(o, out, 16, hipMemcpyDeviceToHost));
      printf("nwg=%3d %s: %.3f us per stage   (carry %llu %llu)\n", nwg, mode == 0 ? "A counters + drain + barrier + load" : mode == 1 ? "B sentinel, no sleep" : mode == 2 ? "B sentinel, sleep 2x64" : mode == 3 ? "B sentinel, sleep 8x64" : "B sentinel, sleep 32x64", best * 1000.f / stages, o[0], o[1]);
    }
  }
  return 0;
}
