// Hand-off latency probe for the data-is-the-flag protocol of csrc/deep_kernel.hip (tuning tool, not part of the product).
//
// Two workgroups of one launch play ping-pong with 64 eight-byte granules {tag, value}: the writer's 64 lanes store one granule each,
// the reader's 64 lanes poll their granule with L1-bypassing loads until every tag matches.  Measured: the one-way hand-off time for
//   placement  : partner on the same XCD (as read from HW_REG_XCC_ID) or on another one
//   store kind : plain (line stays in the XCD's L2), sc1 (write-through, line dropped), sc0 sc1
//   lines      : the same 512 B every round ("reuse") or fresh, zeroed 512 B regions ("fresh": what a tensor written once per
//                launch looks like to its consumer)
// Every spin is bounded; a time-out is reported as such (a plain store is not expected to reach another XCD at all).
//
//   hipcc --offload-arch=gfx950 -O3 -o scratch/xcd_handoff_probe tools/xcd_handoff_probe.hip && scratch/xcd_handoff_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int SK> __device__ __forceinline__ void store_g(u64* p, u64 v) {
  if (SK == 0) asm volatile("global_store_dwordx2 %0, %1, off\n s_nop 1" :: "v"(p), "v"(v) : "memory");
  else if (SK == 1) asm volatile("global_store_dwordx2 %0, %1, off sc1\n s_nop 1" :: "v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n s_nop 1" :: "v"(p), "v"(v) : "memory");
}
template <int LK> __device__ __forceinline__ u64 load_g(const u64* p) {
  u64 v;
  if (LK == 0) asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (LK == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

struct Args {
  u64* ping;          // [rounds or 1][64]
  u64* pong;
  int* xcc;           // [grid]
  int a, b;           // the two players
  int rounds;
  int fresh;          // 1: a new 512 B region per round
  u64* out;           // [0] ticks (100 MHz) of player a, [1] time-outs, [2] mismatching values
  int load_wgs;       // further workgroups streaming memory meanwhile (0: idle chip)
  float* load_buf; size_t load_n;
  int* stop;
};

template <int SK, int LK>
__global__ __launch_bounds__(64) void probe(Args A) {
  const int w = blockIdx.x, lane = threadIdx.x;
  if (lane == 0) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    A.xcc[w] = (int)(x & 0xf);
  }
  if (w != A.a && w != A.b) {
    if (w < A.load_wgs + 2 && A.load_buf) {
      // background traffic until the players are done
      float s = 0.f;
      size_t i = (size_t)w * 64 + lane;
      for (int it = 0; it < 1 << 20; ++it) {
        s += A.load_buf[i % A.load_n];
        i += (size_t)gridDim.x * 64;
        if ((it & 255) == 0 && __hip_atomic_load(A.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      }
      if (s == 1234.5f) A.load_buf[0] = s;
    }
    return;
  }
  const bool first = w == A.a;
  u64 timeouts = 0, bad = 0;
  const u64 t0 = __builtin_amdgcn_s_memrealtime();
  for (int r = 1; r <= A.rounds; ++r) {
    const size_t off = A.fresh ? (size_t)(r - 1) * 64 : 0;
    u64* mine = (first ? A.ping : A.pong) + off + lane;
    const u64* theirs = (first ? A.pong : A.ping) + off + lane;
    const u64 word = ((u64)r << 32) | (unsigned)(r * 64 + lane);
    if (first) store_g<SK>(mine, word);
    // wait for the partner's granules of this round
    unsigned spins = 0;
    bool ok;
    u64 v;
    do {
      v = load_g<LK>(theirs);
      ok = (v >> 32) == (u64)r;
    } while (!__all(ok) && ++spins < 200000u);
    if (!__all(ok)) { timeouts += 1; break; }
    bad += (unsigned)v != (unsigned)(r * 64 + lane);
    if (!first) store_g<SK>(mine, word);
  }
  const u64 t1 = __builtin_amdgcn_s_memrealtime();
  if (first) {
    if (lane == 0) A.out[0] = t1 - t0;
    __hip_atomic_store(A.stop, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (lane == 0) atomicAdd(&A.out[1], timeouts);
  if (bad) atomicAdd(&A.out[2], bad);
}

int main(int argc, char** argv) {
  const int grid = 256, rounds = 2000;
  u64 *ping, *pong, *out;
  int *xcc, *stop;
  float* load_buf;
  const size_t load_n = (size_t)64 << 20;
  CHECK(hipMalloc(&ping, (size_t)rounds * 64 * 8));
  CHECK(hipMalloc(&pong, (size_t)rounds * 64 * 8));
  CHECK(hipMalloc(&out, 64));
  CHECK(hipMalloc(&xcc, grid * 4));
  CHECK(hipMalloc(&stop, 64));
  CHECK(hipMalloc(&load_buf, load_n * 4));
  CHECK(hipMemset(load_buf, 0, load_n * 4));
  std::vector<int> hx(grid);
  // placement census first
  {
    Args A{ping, pong, xcc, 0, 1, 1, 0, out, 0, nullptr, 0, stop};
    CHECK(hipMemset(ping, 0, 64 * 8)); CHECK(hipMemset(pong, 0, 64 * 8)); CHECK(hipMemset(out, 0, 64)); CHECK(hipMemset(stop, 0, 4));
    hipLaunchKernelGGL((probe<1, 0>), dim3(grid), dim3(64), 0, 0, A);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(hx.data(), xcc, grid * 4, hipMemcpyDeviceToHost));
    printf("# XCC id of workgroups 0..15:");
    for (int i = 0; i < 16; ++i) printf(" %d", hx[i]);
    int rr = 1;
    for (int i = 0; i < grid; ++i) rr &= hx[i] == hx[i % 8];
    printf("   (workgroup b on the XCD of b %% 8 for all %d: %s)\n", grid, rr ? "yes" : "NO");
  }
  int same = -1, other = -1;
  for (int i = 1; i < grid && (same < 0 || other < 0); ++i) {
    if (hx[i] == hx[0] && same < 0) same = i;
    if (hx[i] != hx[0] && other < 0) other = i;
  }
  printf("# players: workgroup 0 with %d (same XCD) or %d (another XCD); %d rounds, one-way time = total / (2 rounds)\n", same, other, rounds);
  printf("%-10s %-8s %-8s %-6s %-6s | %10s %9s %6s\n", "placement", "store", "load", "lines", "load", "one-way us", "time-outs", "bad");
  const char* sk_name[3] = {"plain", "sc1", "sc0sc1"};
  const char* lk_name[3] = {"sc1", "sc0sc1", "sc0"};
  for (int busy = 0; busy < 2; ++busy)
    for (int place = 0; place < 2; ++place)
      for (int sk = 0; sk < 3; ++sk)
        for (int lk = 0; lk < 3; ++lk)
          for (int fresh = 0; fresh < 2; ++fresh) {
            Args A{ping, pong, xcc, 0, place ? other : same, rounds, fresh, out, busy ? 200 : 0, busy ? load_buf : nullptr, load_n, stop};
            CHECK(hipMemset(ping, 0, (size_t)rounds * 64 * 8)); CHECK(hipMemset(pong, 0, (size_t)rounds * 64 * 8));
            CHECK(hipMemset(out, 0, 64)); CHECK(hipMemset(stop, 0, 4));
            CHECK(hipDeviceSynchronize());
#define LAUNCH(S, L) hipLaunchKernelGGL((probe<S, L>), dim3(grid), dim3(64), 0, 0, A)
            switch (sk * 3 + lk) {
              case 0: LAUNCH(0, 0); break; case 1: LAUNCH(0, 1); break; case 2: LAUNCH(0, 2); break;
              case 3: LAUNCH(1, 0); break; case 4: LAUNCH(1, 1); break; case 5: LAUNCH(1, 2); break;
              case 6: LAUNCH(2, 0); break; case 7: LAUNCH(2, 1); break; default: LAUNCH(2, 2); break;
            }
            CHECK(hipDeviceSynchronize());
            u64 h[3];
            CHECK(hipMemcpy(h, out, 24, hipMemcpyDeviceToHost));
            printf("%-10s %-8s %-8s %-6s %-6s | %10.3f %9llu %6llu\n", place ? "cross-XCD" : "same-XCD", sk_name[sk], lk_name[lk], fresh ? "fresh" : "reuse",
                   busy ? "busy" : "idle", (double)h[0] * 0.01 / (2.0 * rounds), h[1], h[2]);
          }
  return 0;
}
