// microbenchmark: the exchange stage of csrc/deep_kernel.hip as it really is shaped -- GROUPS of workgroups (the M tiles of one batch
// group) exchange a tile among themselves: every member produces 1/G of a 16 KB tile, every member then consumes the WHOLE tile
// (2 x 16-byte vectors per thread, 512 threads) with the data-is-the-flag protocol.  Question: what does the stage cost when
//   spread   the members of a group sit on arbitrary XCDs (workgroup id / G: what the persistent launch does today), sc1 stores
//   local    the members of a group sit on ONE XCD (as read from HW_REG_XCC_ID), sc1 stores (the 16 readers of a line share an L2)
//   local+   the same with PLAIN stores (the line stays in that XCD's L2; only valid because every reader is on the writer's XCD)
// Every stage is a true dependency (what is stored in stage p depends on everything loaded in stage p - 1); spins are bounded.
// hipcc --offload-arch=gfx950 -O3 groupchain.hip -o groupchain && ./groupchain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr int NT = 512;
constexpr int TILE_WORDS = 2048;        // 16 KB tile of a group per stage
constexpr int VEC = 2;                  // 16-byte vectors a thread consumes per stage (512 x 2 x 16 B = the whole tile)

__device__ __forceinline__ void store_plain(u64* p, u64 v) { asm volatile("global_store_dwordx2 %0, %1, off\n s_nop 1" :: "v"(p), "v"(v) : "memory"); }

// role[wg] = group * G + member, or -1 (idle)
template <int PLAIN, int WIDE>
__global__ __launch_bounds__(NT) void chain(u64* buf, const int* role, int G, int ngroups, int stages, u64* out, unsigned* fail) {
  const int tid = threadIdx.x, wg = blockIdx.x;
  const int r = role[wg];
  if (r < 0) return;
  const int grp = r / G, mem = r % G;
  const int wpm = TILE_WORDS / G;                    // words a member produces per stage
  u64 carry = 1;
  for (int p = 0; p < stages; ++p) {
    u64* cur = buf + ((size_t)p * ngroups + grp) * TILE_WORDS;
    if (tid < wpm) {
      const u64 word = (carry & 0xffffffffull) | ((u64)(p + 1) << 32);
      if (PLAIN) store_plain(cur + mem * wpm + tid, word);
      else __hip_atomic_store((gu64*)(u64)(cur + mem * wpm + tid), word, RLX);
    }
    u64 w[VEC][2];
    unsigned spins = 0;
    for (;; ++spins) {
      bool bad = false;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const int idx = (tid + i * NT) * 2;
        if (WIDE) {
          // one 16-byte sc1 load per vector
          typedef unsigned __attribute__((ext_vector_type(4))) u32x4;
          u32x4 v;
          asm volatile("global_load_dwordx4 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(cur + idx) : "memory");
          w[i][0] = ((u64)v[1] << 32) | v[0];
          w[i][1] = ((u64)v[3] << 32) | v[2];
        } else {
          w[i][0] = __hip_atomic_load((gu64*)(u64)(cur + idx), RLX);
          w[i][1] = __hip_atomic_load((gu64*)(u64)(cur + idx + 1), RLX);
        }
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) bad |= (w[i][0] == ~0ull) | (w[i][1] == ~0ull);
      if (!__builtin_amdgcn_ballot_w64(bad)) break;
      if (spins > (1u << 18)) { if ((tid & 63) == 0) atomicAdd(fail, 1u); return; }
      __builtin_amdgcn_s_sleep(2);
    }
    u64 s = 0;
#pragma unroll
    for (int i = 0; i < VEC; ++i) s += (w[i][0] & 0xffff) + (w[i][1] & 0xffff);
    carry = (carry + s) & 0xffff;
    __syncthreads();                                 // (the real unit has one barrier per stage behind its staging anyway)
  }
  if (tid == 0) out[wg] = carry;
}

__global__ void census(int* xcc) {
  if (threadIdx.x == 0) {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    xcc[blockIdx.x] = (int)(x & 0xf);
  }
}

int main() {
  const int stages = 400, nwg = 256;
  u64 *buf, *out;
  int *role, *xcc;
  unsigned* fail;
  const size_t bbytes = (size_t)stages * 16 * TILE_WORDS * 8;
  CK(hipMalloc(&buf, bbytes)); CK(hipMalloc(&out, nwg * 8)); CK(hipMalloc(&role, nwg * 4)); CK(hipMalloc(&xcc, nwg * 4)); CK(hipMalloc(&fail, 4));
  hipLaunchKernelGGL(census, dim3(nwg), dim3(NT), 0, 0, xcc);
  int hx[256];
  CK(hipMemcpy(hx, xcc, nwg * 4, hipMemcpyDeviceToHost));
  int rr = 1;
  for (int i = 0; i < nwg; ++i) rr &= hx[i] == i % 8;
  printf("# workgroup b on XCD b %% 8: %s\n", rr ? "yes" : "NO (the local rows below are not local)");
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int G : {16, 32}) {
    for (int active : {128, 256}) {
      const int ngroups = active / G;
      for (int mode = 0; mode < 3; ++mode) {
        // spread: group = id / G over the first `active` workgroups;  local: the workgroups of XCD x are x, x + 8, ...: its groups are
        // consecutive runs of G of them (active / 8 workgroups per XCD take part)
        int hr[256];
        for (int i = 0; i < nwg; ++i) hr[i] = -1;
        if (mode == 0) {
          for (int i = 0; i < active; ++i) hr[i] = i;
        } else {
          const int per_xcd = active / 8;               // members per XCD
          if (per_xcd % G != 0 && G % per_xcd != 0) continue;
          if (per_xcd < G) { printf("G=%d active=%d local: a group does not fit one XCD's share, skipped\n", G, active); break; }
          int g = 0;
          for (int x = 0; x < 8; ++x)
            for (int j = 0; j < per_xcd; ++j) hr[x + 8 * j] = (x * (per_xcd / G) + j / G) * G + j % G, g = 0;
          (void)g;
        }
        CK(hipMemcpy(role, hr, nwg * 4, hipMemcpyHostToDevice));
        for (int wide = 0; wide < 2; ++wide) {
          float best = 1e9;
          unsigned hf = 0;
          for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(buf, 0xFF, bbytes)); CK(hipMemset(fail, 0, 4));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            if (mode == 2) { if (wide) hipLaunchKernelGGL((chain<1, 1>), dim3(nwg), dim3(NT), 0, 0, buf, role, G, ngroups, stages, out, fail);
                             else hipLaunchKernelGGL((chain<1, 0>), dim3(nwg), dim3(NT), 0, 0, buf, role, G, ngroups, stages, out, fail); }
            else { if (wide) hipLaunchKernelGGL((chain<0, 1>), dim3(nwg), dim3(NT), 0, 0, buf, role, G, ngroups, stages, out, fail);
                   else hipLaunchKernelGGL((chain<0, 0>), dim3(nwg), dim3(NT), 0, 0, buf, role, G, ngroups, stages, out, fail); }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
            unsigned f; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost)); hf += f;
          }
          printf("G=%2d active=%3d %-28s %s loads: %.3f us per stage%s\n", G, active,
                 mode == 0 ? "spread, sc1 stores" : mode == 1 ? "one XCD per group, sc1 stores" : "one XCD per group, plain", wide ? "16-B" : " 8-B", best * 1000.f / stages,
                 hf ? "   TIME-OUTS" : "");
        }
      }
    }
  }
  return 0;
}
