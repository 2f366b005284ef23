// Safety probe for an XCD-local hand-off with PLAIN stores (see groupchain.hip): can a consumer that polls with L1-bypassing (sc1)
// loads ever see the PREVIOUS launch's value of a word instead of the poison written between the launches?
//
// Per round e = 1..R:   launch 1  poison kernel: every word of X := all ones   (plain stores by workgroups on every XCD, as jen1_deep_poison)
//                       launch 2  exchange kernel: the producer (workgroup 0) waits ~15 us, then stores {e, i} to X with plain or
//                                 sc1 stores; the consumer (same XCD or another one) polls X with sc1 loads from the start and
//                                 classifies every word it sees: poison, the value of THIS round, or anything else (= stale)
// A stale count > 0 means the line of round e - 1 survived in that XCD's L2 across the launches and a data-is-the-flag consumer
// would have taken it for fresh data.
// hipcc --offload-arch=gfx950 -O3 stale_l2.hip -o stale_l2 && ./stale_l2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;
constexpr int WORDS = 4096;            // 32 KB exchanged per round

__global__ void poison(u64* x, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[i] = ~0ull;
}

__device__ __forceinline__ void st_plain(u64* p, u64 v) { asm volatile("global_store_dwordx2 %0, %1, off\n s_nop 1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_sc1(u64* p, u64 v) { asm volatile("global_store_dwordx2 %0, %1, off sc1\n s_nop 1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ u64 ld_sc1(const u64* p) {
  u64 v;
  asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// counts[0] stale words seen, [1] polls that saw poison, [2] time-outs, [3] words accepted
template <int PLAIN>
__global__ __launch_bounds__(512) void exchange(u64* x, int epoch, int consumer_wg, u64* counts, int delay_us) {
  const int wg = blockIdx.x, tid = threadIdx.x;
  if (wg == 0) {
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (u64)delay_us * 100) __builtin_amdgcn_s_sleep(8);
    for (int i = tid; i < WORDS; i += 512) {
      const u64 w = ((u64)epoch << 32) | (unsigned)i;
      if (PLAIN) st_plain(x + i, w); else st_sc1(x + i, w);
    }
  } else if (wg == consumer_wg) {
    u64 stale = 0, pois = 0, ok = 0, tmo = 0;
    for (int i = tid; i < WORDS; i += 512) {
      unsigned spins = 0;
      for (;;) {
        const u64 v = ld_sc1(x + i);
        if (v == ~0ull) { pois++; if (++spins > 400000u) { tmo++; break; } __builtin_amdgcn_s_sleep(2); continue; }
        if (v == (((u64)epoch << 32) | (unsigned)i)) ok++; else stale++;
        break;
      }
    }
    atomicAdd(&counts[0], stale); atomicAdd(&counts[1], pois); atomicAdd(&counts[2], tmo); atomicAdd(&counts[3], ok);
  }
}

int main() {
  u64 *x, *counts;
  CK(hipMalloc(&x, WORDS * 8)); CK(hipMalloc(&counts, 64));
  CK(hipMemset(x, 0, WORDS * 8));
  printf("%-10s %-6s | %8s %10s %9s %9s\n", "placement", "store", "stale", "poison", "time-outs", "accepted");
  for (int place = 0; place < 2; ++place)
    for (int plain = 1; plain >= 0; --plain) {
      if (place == 1 && plain == 1) continue;       // plain stores never reach another XCD in time: not a candidate
      CK(hipMemset(counts, 0, 64));
      const int consumer = place ? 1 : 8;           // workgroup b runs on XCD b % 8
      for (int e = 1; e <= 200; ++e) {
        hipLaunchKernelGGL(poison, dim3(256), dim3(256), 0, 0, x, WORDS);
        if (plain) hipLaunchKernelGGL(exchange<1>, dim3(256), dim3(512), 0, 0, x, e, consumer, counts, 15);
        else hipLaunchKernelGGL(exchange<0>, dim3(256), dim3(512), 0, 0, x, e, consumer, counts, 15);
      }
      CK(hipDeviceSynchronize());
      u64 h[4];
      CK(hipMemcpy(h, counts, 32, hipMemcpyDeviceToHost));
      printf("%-10s %-6s | %8llu %10llu %9llu %9llu\n", place ? "cross-XCD" : "same-XCD", plain ? "plain" : "sc1", h[0], h[1], h[2], h[3]);
    }
  return 0;
}
