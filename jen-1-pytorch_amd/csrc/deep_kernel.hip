// Persistent deep-level kernel of the JEN-1 denoiser (gfx950 / MI355X).  C ABI: include/jen1_deep.h.
//
// The levels with T' <= 24 positions (down 3..8, bottleneck, up 0..5 of reference jen1/model/model.py:246-259,
// blocks jen1/model/blocks.py:540-830) are ~165 tiny dependent layers over 288 M parameters.  One launch of
// `nwg` resident 512-thread workgroups (one per CU) walks a list of phases; a unit of a phase is done by one
// workgroup, phase p starts when every unit of phase p-1 has arrived.  What a phase costs is the exchange
// (write-through stores -> sharded arrival counter -> poll -> L1-bypassing loads), so everything else is moved
// off that chain:
//   * the weight slice of a unit (K x 16 rows, up to 16 ring slots of 1 KiB per wave, 8 waves) and the
//     GroupNorm / FiLM parameters are requested BEFORE the dependency wait;
//   * GroupNorm(+FiLM)(+SiLU) (blocks.py:137-145, :509) is applied by the consumer while it stages the tile of
//     its batch elements in LDS (no norm_apply pass, no statistics arena): per-vector partial sums -> LDS ->
//     one quad per (batch element, group) adds them in a fixed order -> bit-reproducible, no float atomics;
//   * LayerNorm statistics of the deferred finish (blocks.py:427-429) are recomputed by the attention unit from
//     the rows themselves (16 lanes per row, DPP reduction);
//   * conv taps, the skip concat and a fused 1x1 shortcut are K segments over ONE staged tile (row-shifted views),
//     taps that only see zero padding are skipped together with their weights (2/3 of a k=3 conv at T' = 1).
// Inter-workgroup visibility follows cdna_hip_programming.md Guideline 16 R1: payload with sc1 (write-through)
// stores, every storing wave drains (asm s_waitcnt vmcnt(0)), one lane arrives with a relaxed agent-scope
// atomic; consumers poll relaxed, then read with sc1 (L1-bypassing) loads.  Every spin is bounded: a time-out
// raises the error word and releases every other waiter.
#include "common.h"
#include "jen1_deep.h"

namespace {

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int NT = JEN1_DEEP_THREADS;
constexpr int NW = NT / 64;
constexpr int SHARDS = JEN1_DEEP_SHARDS;
constexpr int SHW = JEN1_DEEP_SHARD_WORDS;
constexpr unsigned OOB = 0x80000000u;
constexpr int RSRC_FLAGS = 0x00020000;
constexpr int QCHUNK = 32;

__device__ __forceinline__ gu64* g64(const void* p) { return (gu64*)(u64)p; }
__device__ __forceinline__ gu32* g32(const void* p) { return (gu32*)(u64)p; }

// ---- 8-element vectors through agent-scope (sc1) accesses: data another workgroup produced in THIS launch --------
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> { u64 d[2]; };
template <> struct Raw8<float> { u64 d[4]; };

__device__ __forceinline__ void ld_live(Raw8<bf16_t>& r, const bf16_t* p) {
  r.d[0] = __hip_atomic_load(g64(p), RLX_AGENT);
  r.d[1] = __hip_atomic_load(g64(p) + 1, RLX_AGENT);
}
__device__ __forceinline__ void ld_live(Raw8<float>& r, const float* p) {
#pragma unroll
  for (int i = 0; i < 4; ++i) r.d[i] = __hip_atomic_load(g64(p) + i, RLX_AGENT);
}
__device__ __forceinline__ void ld_plain(Raw8<bf16_t>& r, const bf16_t* p) {
  const u32x4 v = *reinterpret_cast<const u32x4*>(p);
  r.d[0] = ((u64)v[1] << 32) | v[0];
  r.d[1] = ((u64)v[3] << 32) | v[2];
}
__device__ __forceinline__ void ld_plain(Raw8<float>& r, const float* p) {
  const u32x4 a = *reinterpret_cast<const u32x4*>(p);
  const u32x4 b = *reinterpret_cast<const u32x4*>(p + 4);
  r.d[0] = ((u64)a[1] << 32) | a[0];
  r.d[1] = ((u64)a[3] << 32) | a[2];
  r.d[2] = ((u64)b[1] << 32) | b[0];
  r.d[3] = ((u64)b[3] << 32) | b[2];
}
__device__ __forceinline__ void zero_raw(Raw8<bf16_t>& r) { r.d[0] = r.d[1] = 0; }
__device__ __forceinline__ void zero_raw(Raw8<float>& r) { r.d[0] = r.d[1] = r.d[2] = r.d[3] = 0; }
__device__ __forceinline__ void raw_to_float(const Raw8<bf16_t>& r, float (&o)[8]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const unsigned lo = (unsigned)r.d[i], hi = (unsigned)(r.d[i] >> 32);
    o[4 * i + 0] = __uint_as_float(lo << 16);
    o[4 * i + 1] = __uint_as_float(lo & 0xffff0000u);
    o[4 * i + 2] = __uint_as_float(hi << 16);
    o[4 * i + 3] = __uint_as_float(hi & 0xffff0000u);
  }
}
__device__ __forceinline__ void raw_to_float(const Raw8<float>& r, float (&o)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[2 * i] = __uint_as_float((unsigned)r.d[i]);
    o[2 * i + 1] = __uint_as_float((unsigned)(r.d[i] >> 32));
  }
}
// 4 consecutive output channels of one position, write-through
__device__ __forceinline__ void st_live4(bf16_t* p, const float (&v)[4]) {
  bf16x4 a;
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = (bf16_t)v[i];
  __hip_atomic_store(g64(p), __builtin_bit_cast(u64, a), RLX_AGENT);
}
__device__ __forceinline__ void st_live4(float* p, const float (&v)[4]) {
  __hip_atomic_store(g64(p), ((u64)__float_as_uint(v[1]) << 32) | __float_as_uint(v[0]), RLX_AGENT);
  __hip_atomic_store(g64(p) + 1, ((u64)__float_as_uint(v[3]) << 32) | __float_as_uint(v[2]), RLX_AGENT);
}
__device__ __forceinline__ void ld_live4(float (&o)[4], const bf16_t* p) {
  const u64 x = __hip_atomic_load(g64(p), RLX_AGENT);
  const unsigned lo = (unsigned)x, hi = (unsigned)(x >> 32);
  o[0] = __uint_as_float(lo << 16); o[1] = __uint_as_float(lo & 0xffff0000u);
  o[2] = __uint_as_float(hi << 16); o[3] = __uint_as_float(hi & 0xffff0000u);
}
__device__ __forceinline__ void ld_live4(float (&o)[4], const float* p) {
  const u64 a = __hip_atomic_load(g64(p), RLX_AGENT), b = __hip_atomic_load(g64(p) + 1, RLX_AGENT);
  o[0] = __uint_as_float((unsigned)a); o[1] = __uint_as_float((unsigned)(a >> 32));
  o[2] = __uint_as_float((unsigned)b); o[3] = __uint_as_float((unsigned)(b >> 32));
}
__device__ __forceinline__ void st_live8(bf16_t* p, const bf16_t* s) {   // 8 elements from LDS, write-through
  const u32x4 v = *reinterpret_cast<const u32x4*>(s);
  __hip_atomic_store(g64(p), ((u64)v[1] << 32) | v[0], RLX_AGENT);
  __hip_atomic_store(g64(p) + 1, ((u64)v[3] << 32) | v[2], RLX_AGENT);
}
__device__ __forceinline__ void st_live8(float* p, const float* s) {
  const u32x4 a = *reinterpret_cast<const u32x4*>(s);
  const u32x4 b = *reinterpret_cast<const u32x4*>(s + 4);
  __hip_atomic_store(g64(p), ((u64)a[1] << 32) | a[0], RLX_AGENT);
  __hip_atomic_store(g64(p) + 1, ((u64)a[3] << 32) | a[2], RLX_AGENT);
  __hip_atomic_store(g64(p) + 2, ((u64)b[1] << 32) | b[0], RLX_AGENT);
  __hip_atomic_store(g64(p) + 3, ((u64)b[3] << 32) | b[2], RLX_AGENT);
}

// ---- MFMA fragments ------------------------------------------------------------------------------------------------
template <typename T> struct DFrag;
template <> struct DFrag<bf16_t> { typedef bf16x8 type; };
template <> struct DFrag<float> { typedef f32x8 type; };
__device__ __forceinline__ void dmma(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void dmma(f32x4& acc, const f32x8& a, const f32x8& b) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], acc, 0, 0, 0);
}
__device__ __forceinline__ void dlds(bf16x8& f, const bf16_t* p) { f = *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ void dlds(f32x8& f, const float* p) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
  f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
}
// weight fragment through a buffer descriptor (out-of-range offsets return 0 and move no bytes); nt: used once per launch
__device__ __forceinline__ void wload(bf16x8& f, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  f = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 2));
}
__device__ __forceinline__ void wload(f32x8& f, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 2);
  const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(r, voff + 16u, soff, 2);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f.v[j] = __uint_as_float(lo[j]);
    f.v[4 + j] = __uint_as_float(hi[j]);
  }
}

template <int CTRL>
__device__ __forceinline__ float ddpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float quad_sum(float v) {       // fixed order: (a+b)+(c+d) in every lane of the quad
  v += ddpp<0xB1>(v);
  v += ddpp<0x4E>(v);
  return v;
}
__device__ __forceinline__ float row16_sum_d(float v) {
  v += ddpp<0xB1>(v);
  v += ddpp<0x4E>(v);
  v += ddpp<0x141>(v);
  v += ddpp<0x140>(v);
  return v;
}
__device__ __forceinline__ float row16_max_d(float v) {
  v = fmaxf(v, ddpp<0xB1>(v));
  v = fmaxf(v, ddpp<0x4E>(v));
  v = fmaxf(v, ddpp<0x141>(v));
  v = fmaxf(v, ddpp<0x140>(v));
  return v;
}

template <typename T> struct DeepCfg;
#ifndef JEN1_DEEP_MAXV_B
#define JEN1_DEEP_MAXV_B 3        // 8-channel vectors per thread of the normalised part and of the raw part of a staged tile (bf16)
#endif
#ifndef JEN1_DEEP_MAXV_F
#define JEN1_DEEP_MAXV_F 3        // same, float32 mode
#endif
#ifndef JEN1_DEEP_PF_B
#define JEN1_DEEP_PF_B 12         // weight ring slots per wave (1 KiB each in bf16, 2 KiB in float32)
#endif
#ifndef JEN1_DEEP_PF_F
#define JEN1_DEEP_PF_F 3
#endif
template <> struct DeepCfg<bf16_t> { static constexpr int MAXV = JEN1_DEEP_MAXV_B, PF = JEN1_DEEP_PF_B; };
template <> struct DeepCfg<float> { static constexpr int MAXV = JEN1_DEEP_MAXV_F, PF = JEN1_DEEP_PF_F; };

// ---- device-resident program: one blob per phase + a header array ----------------------------------------------------
//   blob  [0, sizeof(jen1_deep_phase))            the descriptor
//         [TAB_OFF, +32)                          int16 cnt[NW] (K chunks of wave w), int16 cnt_low[NW] (those below g_split)
//         [TAB_OFF + 32, ...)                     NW lists of MAXE entries {chunk index g, staged column | shift << 16}:
//                                                 the usable chunks of the flat (segment, chunk) list dealt round-robin
//   header {n_units, rot, kind, -}                what a workgroup needs to find its next unit without touching the blobs
constexpr int BLOB = JEN1_DEEP_BLOB_BYTES;
constexpr int TAB_OFF = 1024;
constexpr int MAXE = (BLOB - TAB_OFF - 32) / (8 * NW);
constexpr int HDR_BYTES = JEN1_DEEP_MAX_PHASES * 16;
constexpr int WS_OFF = HDR_BYTES + 2 * BLOB;           // LDS: headers | two descriptor slots | unit workspace
static_assert(sizeof(jen1_deep_phase) <= TAB_OFF, "descriptor must fit ahead of the chunk table");
static_assert(BLOB == NT * 8, "one 8-byte word per thread moves a blob");
static_assert(JEN1_DEEP_MAX_PHASES <= NT, "one header per thread at start-up");
struct Hdr {
  int n_units, rot, kind, pad;
};

// ---- synchronisation -------------------------------------------------------------------------------------------------
struct Sync {
  unsigned* base;      // counters: phase p, shard s at base[(p * SHARDS + s) * SHW]
  unsigned* err;       // error word
  bool dead;           // wave 0: a wait timed out somewhere: stop waiting, finish with whatever is there
  int p, wg, nwg;      // current phase / this workgroup
};

// Tuning builds only (-DJEN1_DEEP_PROFILE): thread 0 of every workgroup records the constant-rate 100 MHz counter at the
// stages of each unit it runs: dbg[(phase * nwg + wg) * 16 + stage]  (jen1_deep_debug_buffer sets the pointer)
#ifdef JEN1_DEEP_PROFILE
__device__ unsigned long long* g_deep_dbg = nullptr;
#define DK_STAMP(sy, i) do { if (threadIdx.x == 0 && g_deep_dbg) \
    g_deep_dbg[((size_t)(sy).p * (sy).nwg + (sy).wg) * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DK_STAMP(sy, i) do { } while (0)
#endif

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// every thread of the workgroup calls this; wave 0 polls (lanes 0..7 one shard each, lane 8 the error word).
// `dead` is wave 0's private knowledge (only it ever polls): once set, later waits fall through.
__device__ __forceinline__ void wait_phase(Sync& sy, int dep, int dep_units, int tid) {
  if (dep < 0) return;
  if (tid < 64 && !sy.dead) {
    const gu32* c = g32(sy.base + ((size_t)dep * SHARDS + (tid < SHARDS ? tid : 0)) * SHW);
    const gu32* e = g32(sy.err);
    unsigned spins = 0;
    for (;;) {
      unsigned v = 0, ev = 0;
      if (tid < SHARDS) v = __hip_atomic_load(c, RLX_AGENT);
      if (tid == SHARDS) ev = __hip_atomic_load(e, RLX_AGENT);
      const float tot = row16_sum_d((float)v);          // counts are small integers: exact in float
      const int total = __builtin_amdgcn_readfirstlane((int)tot);
      const int errv = __builtin_amdgcn_readlane((int)ev, SHARDS);
      if (total >= dep_units) break;
      if (errv != 0) { sy.dead = true; break; }
      if (++spins > (1u << 22)) {
        if (tid == 0) __hip_atomic_store(g32(sy.err), (unsigned)(dep + 1), RLX_AGENT);
        sy.dead = true;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  __syncthreads();
}

// called after a __syncthreads() that follows every storing wave's drain
__device__ __forceinline__ void arrive_phase(const Sync& sy, int tid) {
  if (tid == 0) __hip_atomic_fetch_add(g32(sy.base + ((size_t)sy.p * SHARDS + (sy.wg % SHARDS)) * SHW), 1u, RLX_AGENT);
}
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// next unit of this workgroup after (p, u): the same phase first (more units than workgroups), then later phases
__device__ __forceinline__ bool find_next(const Hdr* hdr, int n_phases, int wg, int nwg, int& p, int& u) {
  if (p >= 0) {
    const int nu = u + nwg;
    if (nu < hdr[p].n_units) { u = nu; return true; }
  }
  for (int q = p + 1; q < n_phases; ++q) {
    int u0 = wg - hdr[q].rot;
    u0 = u0 < 0 ? u0 + nwg : u0;
    if (u0 < hdr[q].n_units) { p = q; u = u0; return true; }
  }
  return false;
}

// =====================================================================================================================
// GEMM unit.  D: the phase's blob in LDS.
// =====================================================================================================================
template <typename T>
struct GemmWave {            // what prefill and the K loop share
  int mt, total;
  bool low_m;
};

template <typename T>
__device__ __forceinline__ GemmWave<T> gemm_wave(const unsigned char* D, int u, int wk) {
  const jen1_deep_phase* P = reinterpret_cast<const jen1_deep_phase*>(D);
  const int MT = P->MT;
  const int grp = (int)(((float)u + 0.5f) * (1.0f / (float)MT));      // u < 2^20: exact
  GemmWave<T> g;
  g.mt = u - grp * MT;
  g.low_m = g.mt < P->mt_split;
  const short* cnt = reinterpret_cast<const short*>(D + TAB_OFF);
  g.total = rfl(g.low_m ? cnt[NW + wk] : cnt[wk]);
  return g;
}

template <typename T, typename Frag>
__device__ __forceinline__ void gemm_issue(const unsigned char* D, const GemmWave<T>& g, int wk, int lane, int j, Frag& fa) {
  constexpr int ES = sizeof(T);
  constexpr unsigned BLK = 512 * ES;
  const jen1_deep_phase* P = reinterpret_cast<const jen1_deep_phase*>(D);
  const u64 wp = (u64)P->w;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void*>(((u64)(unsigned)rfl((int)(wp >> 32)) << 32) | (unsigned)rfl((int)wp)), 0, rfl((int)P->w_bytes), RSRC_FLAGS);
  const uint2 e = reinterpret_cast<const uint2*>(D + TAB_OFF + 32)[wk * MAXE + j];
  const unsigned voff = (e.x * (unsigned)P->MT + (unsigned)g.mt) * BLK + (unsigned)lane * (8u * ES);
  wload(fa, rw, voff, 0);
}

// fill the ring of the unit (called as early as the descriptor is known: before the previous unit's epilogue, or at unit start)
template <typename T, typename Frag, int PF>
__device__ __forceinline__ void gemm_prefill(const unsigned char* D, int u, int wk, int lane, Frag (&ra)[PF]) {
  const GemmWave<T> g = gemm_wave<T>(D, u, wk);
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    if (i < g.total) gemm_issue<T>(D, g, wk, lane, i, ra[i]);
  }
}

template <typename T, typename Frag, int PF, typename FPub, typename FPre>
__device__ __forceinline__ void gemm_unit(const unsigned char* D, int u, Sync& sy, bool need_wait, int dep_units, Frag (&ra)[PF],
                                          bool prefilled, FPub publish_next, FPre prefill_next, int tid) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr bool PRECISE = is_f32<T>::value;
  constexpr int MAXV = DeepCfg<T>::MAXV;
  const jen1_deep_phase* P = reinterpret_cast<const jen1_deep_phase*>(D);
  const int lane = tid & 63;
  const int wk = rfl(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  DK_STAMP(sy, 0);
  const GemmWave<T> gw = gemm_wave<T>(D, u, wk);
  if (!prefilled) gemm_prefill<T>(D, u, wk, lane, ra);
  DK_STAMP(sy, 7);

  // ---- descriptor fields into registers, once ---------------------------------------------------------------------------
  const int MT = P->MT, nb = P->nb, B = P->B, L_in = P->L_in, L_out = P->L_out, stride = P->stride, NF = P->NF;
  const int pitch = P->pitch, R = P->R, norm_C = P->norm_C, Ctot = P->Ctot, pro = P->pro_mode;
  const int gran = P->gran, cpg = P->gn_cpg, groups = P->gn_groups, nsrc = P->nsrc;
  const float inv_Lin = P->inv_Lin, inv_Lout = P->inv_Lout;
  const int grp = (int)(((float)u + 0.5f) * (1.0f / (float)MT));
  const int mt = gw.mt, b0 = grp * nb;
  const bool low_m = gw.low_m;
  jen1_deep_src src[JEN1_DEEP_MAX_SRC];
#pragma unroll
  for (int k = 0; k < JEN1_DEEP_MAX_SRC; ++k) src[k] = P->src[k];
  unsigned char* ws = smem + WS_OFF;
  T* tile = reinterpret_cast<T*>(ws);
  float2* part = reinterpret_cast<float2*>(ws + P->part_off);
  float2* stat = reinterpret_cast<float2*>(ws + P->stat_off);
  float* red = reinterpret_cast<float*>(ws + P->red_off);

  // ---- staging plan: normalised vectors (idx < Vn over the leading norm_C channels) and raw vectors (the rest) -------------
  const int VPRn = norm_C >> 3, VPRr = (Ctot - norm_C) >> 3;
  const int Vn = R * VPRn, Vr = R * VPRr;
  const float inv_vprn = VPRn ? 1.0f / (float)VPRn : 0.f, inv_vprr = VPRr ? 1.0f / (float)VPRr : 0.f;
  auto locate = [&](int row, int c, const T*& ap, float& sc, int& bl) -> bool {       // tile (row, channel c) -> source address
    bl = (int)(((float)row + 0.5f) * inv_Lin);
    const int t = row - bl * L_in;
    const void* xp = src[0].x;
    int ld = src[0].ld, coff = 0;
    sc = src[0].scale;
#pragma unroll
    for (int k = 1; k < JEN1_DEEP_MAX_SRC; ++k) {
      const bool use = k < nsrc && c >= src[k].coff;
      xp = use ? src[k].x : xp;
      ld = use ? src[k].ld : ld;
      coff = use ? src[k].coff : coff;
      sc = use ? src[k].scale : sc;
    }
    ap = reinterpret_cast<const T*>(xp) + ((size_t)((unsigned)((b0 + bl) * L_in + t) * (unsigned)ld) + (unsigned)(c - coff));
    return b0 + bl < B;
  };
  DK_STAMP(sy, 8);
  // GroupNorm (* FiLM) parameters of this thread's normalised vectors: requested before the dependency wait
  float p1[MAXV][8], p2[MAXV][8];
  int nrow[MAXV], nc[MAXV], nbl[MAXV];
  bool nokv[MAXV];
  const T* nap[MAXV];
  float nsc[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int idx = tid + i * NT;
    nrow[i] = (int)(((float)idx + 0.5f) * inv_vprn);
    nc[i] = (idx - nrow[i] * VPRn) * 8;
    nokv[i] = idx < Vn && locate(nrow[i], nc[i], nap[i], nsc[i], nbl[i]);
    if (nokv[i]) {
      const int b = b0 + nbl[i];
      const int fr = P->p_ld ? (P->film_step ? P->film_step[0] : (P->film_row ? P->film_row[b] : b)) : 0;
      const size_t po = (size_t)((unsigned)fr * (unsigned)P->p_ld) + (unsigned)nc[i];
      load8(P->p1 + po, p1[i]);
      load8(P->p2 + po, p2[i]);
    }
  }
  DK_STAMP(sy, 9);
  // epilogue operands that do not depend on other workgroups
  const bool epi = wk < NF;
  const int nfe = wk;                                   // the fragment this wave finishes
  const int m = mt * 16 + lg * 4;
  int ph = 0, co = m;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  bool okk = false;
  int yrow = 0;
  if (epi) {
    const int out_C = P->out_C, ps_f = P->ps_f;
    for (int k = 1; k < ps_f; ++k) ph += (m >= k * out_C) ? 1 : 0;
    co = m - ph * out_C;
    if (P->bias) bias4 = *reinterpret_cast<const f32x4*>(P->bias + co);
    const int n = nfe * 16 + li;
    const int ebl = (int)(((float)n + 0.5f) * inv_Lout);
    const int t = n - ebl * L_out;
    const int ty = t * ps_f + ph - P->ps_off;
    okk = n < nb * L_out && b0 + ebl < B && ty >= 0 && ty < P->L_y;
    yrow = okk ? (b0 + ebl) * P->y_brows + P->y_row0 + ty : 0;
  }
  const bool use_res = epi && okk && P->residual && (P->mt_split == 0 || low_m);
  const T* resp = reinterpret_cast<const T*>(P->residual) + ((size_t)((unsigned)yrow * (unsigned)P->ld_res) + (unsigned)co);
  const int act = P->act, y_f32 = P->y_f32, ld_y = P->ld_y;
  void* const yp = P->y;
  const float inv_count = P->inv_count, gn_eps = P->gn_eps, inv_groups = P->inv_groups;

  // ---- dependency ---------------------------------------------------------------------------------------------------
  DK_STAMP(sy, 1);
  if (need_wait) wait_phase(sy, sy.p - 1, dep_units, tid);
  DK_STAMP(sy, 2);

  // ---- stage the tile: every load first (sc1: another workgroup wrote the data in this launch) ----------------------------
  Raw8<T> xn[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    if (nokv[i]) ld_live(xn[i], nap[i]);
    else zero_raw(xn[i]);
  }
  float rres[4] = {0.f, 0.f, 0.f, 0.f};
  if (use_res) ld_live4(rres, resp);
  DK_STAMP(sy, 10);
  // zero row (conv padding) behind the R staged rows
  for (int i = tid; i < (Ctot >> 3); i += NT) {
    const float z8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    store8(tile + (size_t)R * pitch + i * 8, z8);
  }
  // raw vectors: straight into the tile, MAXV per thread in flight (one trip unless the tile is a long raw input, e.g. the
  // 94 rows a downsampling conv reads)
  for (int base = 0; base < Vr; base += MAXV * NT) {
    Raw8<T> xw[MAXV];
    int wrow[MAXV], wc[MAXV];
    bool wok[MAXV];
    float wsc[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int idx = base + tid + i * NT;
      wrow[i] = (int)(((float)idx + 0.5f) * inv_vprr);
      wc[i] = norm_C + (idx - wrow[i] * VPRr) * 8;
      const T* ap;
      int bl;
      wok[i] = idx < Vr && locate(wrow[i], wc[i], ap, wsc[i], bl);
      if (wok[i]) ld_live(xw[i], ap);
      else zero_raw(xw[i]);
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      if (wok[i]) {
        if (wsc[i] == 1.0f) {
          *reinterpret_cast<Raw8<T>*>(tile + (size_t)wrow[i] * pitch + wc[i]) = xw[i];
        } else {
          float x[8];
          raw_to_float(xw[i], x);
#pragma unroll
          for (int j = 0; j < 8; ++j) x[j] *= wsc[i];
          store8(tile + (size_t)wrow[i] * pitch + wc[i], x);
        }
      }
    }
  }
  DK_STAMP(sy, 11);
  if (norm_C) {
    // per-vector partial sums (granules of 8 / 4 / 2 / 1 channels; narrower than a vector only in tiny configurations)
    const int gpr = norm_C / gran;
    const int lg2 = gran >= 8 ? 3 : (gran >= 4 ? 2 : (gran >= 2 ? 1 : 0));
    float xf[MAXV][8];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      raw_to_float(xn[i], xf[i]);
      if (nokv[i]) {
        if (nsc[i] != 1.0f) {
#pragma unroll
          for (int j = 0; j < 8; ++j) xf[i][j] *= nsc[i];
        }
        float es[8], eq[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { es[j] = xf[i][j]; eq[j] = xf[i][j] * xf[i][j]; }
        if (gran >= 2) {
#pragma unroll
          for (int j = 0; j < 8; j += 2) { es[j] += es[j + 1]; eq[j] += eq[j + 1]; }
        }
        if (gran >= 4) {
#pragma unroll
          for (int j = 0; j < 8; j += 4) { es[j] += es[j + 2]; eq[j] += eq[j + 2]; }
        }
        if (gran >= 8) { es[0] += es[4]; eq[0] += eq[4]; }
        float2* pp = part + (size_t)nrow[i] * gpr + nc[i] / gran;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if ((j & (gran - 1)) == 0) pp[j >> lg2] = make_float2(es[j], eq[j]);
        }
      }
    }
    __syncthreads();
    DK_STAMP(sy, 12);
    // 16 lanes per (batch element of the unit, group): lane j adds elements j, j+16, ... then a DPP tree, fixed order
    const int pairs = nb * groups;
    for (int pid = tid >> 4; pid < pairs; pid += NT / 16) {
      const int bl = (int)(((float)pid + 0.5f) * inv_groups);
      const int g = pid - bl * groups;
      const int glo = g * cpg / gran;
      const int ghi = (g == groups - 1) ? gpr : (g + 1) * cpg / gran;       // the last group also takes padding channels
      const int ngr = ghi - glo;
      const int n_el = L_in * ngr;
      float s = 0.f, q = 0.f;
      if (b0 + bl < B) {
        const float inv_ngr = 1.0f / (float)ngr;
#pragma unroll 4
        for (int e = li; e < n_el; e += 16) {
          const int t = (int)(((float)e + 0.5f) * inv_ngr);
          const int gi = e - t * ngr;
          const float2 v = part[(size_t)(bl * L_in + t) * gpr + glo + gi];
          s += v.x;
          q += v.y;
        }
      }
      s = row16_sum_d(s);
      q = row16_sum_d(q);
      if (li == 0) {
        const float mean = s * inv_count;
        float var = q * inv_count - mean * mean;
        var = var < 0.f ? 0.f : var;
        const float rstd = PRECISE ? 1.0f / sqrtf(var + gn_eps) : rsqrtf(var + gn_eps);
        stat[pid] = make_float2(mean, rstd);
      }
    }
    __syncthreads();
    DK_STAMP(sy, 13);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      if (nokv[i]) {
        if (cpg >= 8) {
          int g = nc[i] / cpg;
          g = g < groups ? g : groups - 1;
          const float2 st = stat[nbl[i] * groups + g];
#pragma unroll
          for (int j = 0; j < 8; ++j) xf[i][j] = (xf[i][j] - st.x) * st.y * p1[i][j] + p2[i][j];
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            int g = (nc[i] + j) / cpg;
            g = g < groups ? g : groups - 1;
            const float2 st = stat[nbl[i] * groups + g];
            xf[i][j] = (xf[i][j] - st.x) * st.y * p1[i][j] + p2[i][j];
          }
        }
        if (pro == JEN1_PRO_GN_SILU) {
#pragma unroll
          for (int j = 0; j < 8; ++j) xf[i][j] = PRECISE ? silu_precise(xf[i][j]) : silu_f(xf[i][j]);
        }
        store8(tile + (size_t)nrow[i] * pitch + nc[i], xf[i]);
      }
    }
  }
  __syncthreads();
  DK_STAMP(sy, 3);

  // ---- K loop: weights from the ring, activation fragments from the staged tile ---------------------------------------------
  f32x4 acc[4];
#pragma unroll
  for (int nf = 0; nf < 4; ++nf) acc[nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int cbl[4], ct[4];
  bool cok[4];
#pragma unroll
  for (int nf = 0; nf < 4; ++nf) {
    const int n = nf * 16 + li;
    const int bl = (int)(((float)n + 0.5f) * inv_Lout);
    cbl[nf] = bl * L_in;
    ct[nf] = (n - bl * L_out) * stride;
    cok[nf] = nf < NF && n < nb * L_out && b0 + bl < B;
  }
  const uint2* ent = reinterpret_cast<const uint2*>(D + TAB_OFF + 32) + wk * MAXE;
  const int total = gw.total;
  auto consume = [&](int j, const typename DFrag<T>::type& fa) {
    const unsigned ey = ent[j].y;
    const int col = (int)(ey & 0xffffu) + lg * 8;
    const int sh = (int)(signed char)(ey >> 16);
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
      if (nf < NF) {
        const int tin = ct[nf] + sh;
        const bool ok = cok[nf] && tin >= 0 && tin < L_in;
        typename DFrag<T>::type fb;
        dlds(fb, tile + (size_t)(ok ? cbl[nf] + tin : R) * pitch + col);
        dmma(acc[nf], fa, fb);
      }
    }
  };
  for (int c = 0; c < total; c += PF) {
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      const int j = c + i;
      if (j < total) {
        consume(j, ra[i]);
        if (j + PF < total) gemm_issue<T>(D, gw, wk, lane, j + PF, ra[i]);
      }
    }
  }
  DK_STAMP(sy, 4);

  // ---- K reduction across the waves (fixed order), epilogue by the first NF waves; the next unit's descriptor is published
  // with the partial sums, its weight ring is requested as soon as this unit's stores have drained ------------------------------
#pragma unroll
  for (int nf = 0; nf < 4; ++nf) {
    if (nf < NF) *reinterpret_cast<float4*>(red + ((size_t)(wk * NF + nf) * 64 + lane) * 4) = make_float4(acc[nf][0], acc[nf][1], acc[nf][2], acc[nf][3]);
  }
  publish_next();
  __syncthreads();
  DK_STAMP(sy, 14);
  if (epi) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) {
      const float4 o = *reinterpret_cast<const float4*>(red + ((size_t)(w2 * NF + nfe) * 64 + lane) * 4);
      v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += bias4[r];
    if (act == JEN1_ACT_GELU && !low_m) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
    }
    if (okk) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += rres[r];
      const size_t off = (size_t)((unsigned)yrow * (unsigned)ld_y) + (unsigned)co;
      if (y_f32) st_live4(reinterpret_cast<float*>(yp) + off, v);
      else st_live4(reinterpret_cast<T*>(yp) + off, v);
    }
    DK_STAMP(sy, 15);
    drain_stores();
  } else {
    prefill_next();
  }
  __syncthreads();
  DK_STAMP(sy, 5);
  arrive_phase(sy, tid);
  if (epi) prefill_next();
  DK_STAMP(sy, 6);
}

// =====================================================================================================================
// attention unit: one (batch element, head, 32-query chunk); attention.hip's structure for 8 waves with the LayerNorm
// statistics of the deferred finish computed here (blocks.py:355-380, :427-429)
// =====================================================================================================================
template <typename T, typename FPub, typename FPre>
__device__ __forceinline__ void attn_unit(const unsigned char* D, int u, Sync& sy, bool need_wait, int dep_units, FPub publish_next,
                                          FPre prefill_next, int tid) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef typename DFrag<T>::type Frag;
  constexpr bool PRECISE = is_f32<T>::value;
  constexpr int MAXVA = (141 * 16 + NT - 1) / NT;      // K / V vectors per thread at the longest supported context
  const jen1_deep_phase* P = reinterpret_cast<const jen1_deep_phase*>(D);
  const int lane = tid & 63;
  const int wave = rfl(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  DK_STAMP(sy, 0);
  const int H = P->H, d = P->d, Nq = P->Nq, Nk = P->Nk, nqc = P->nqc;
  const int bh = (int)(((float)u + 0.5f) * P->inv_nqc);
  const int qc = u - bh * nqc;
  const int b = (int)(((float)bh + 0.5f) * P->inv_H), h = bh - b * H;
  const int q0 = qc * QCHUNK;
  const int nq = (Nq - q0 < QCHUNK) ? (Nq - q0) : QCHUNK;
  const int NKP = (Nk + 31) & ~31;
  const int DP = d < 32 ? 32 : d;
  const int DC = d < 16 ? 16 : d;
  const int dq = DP + 8, vt = NKP + 8, sp = NKP + 4;
  T* q_s = reinterpret_cast<T*>(smem + WS_OFF);                      // [32][dq]  (later the output tile)
  T* kv_s = q_s + QCHUNK * dq;                                      // K [NKP][dq], later V^T [DC][vt]
  const int kv_elems = (NKP * dq > DC * vt) ? NKP * dq : DC * vt;
  float* s_s = reinterpret_cast<float*>(kv_s + ((kv_elems + 7) & ~7));   // [32][sp]
  T* p_s = reinterpret_cast<T*>(s_s + ((QCHUNK * sp + 3) & ~3));      // [32][vt]
  float2* st_s = reinterpret_cast<float2*>(p_s + ((QCHUNK * vt + 7) & ~7));   // [max(Nk, 32)] LayerNorm (mean, rstd) per row

  const int ldq = P->ldq, ldkv = P->ldkv, log2_vpr = P->log2_vpr, ldo = P->ldo, ln_C = P->ln_C;
  const int vpr = 1 << log2_vpr;
  const int nkv = Nk * vpr, nqv = nq * vpr;
  const int hd = h * d;
  const T* qp = reinterpret_cast<const T*>(P->q);
  const T* kp_ = reinterpret_cast<const T*>(P->k);
  const T* vp_ = reinterpret_cast<const T*>(P->v);
  const T* xp_ = reinterpret_cast<const T*>(P->kv_extra);
  T* const outp = reinterpret_cast<T*>(P->out);
  const int fin_q = P->fin_q, fin_kv = P->fin_kv, kv_live = P->kv_live, causal = P->causal;
  const float scale = P->scale, ln_eps = P->ln_eps;
  const int q_off = P->q_off;

  // ---- operands that do not depend on other workgroups: cached text K/V, the finish vectors u / b ----------------------
  const int kvbase = (P->kv_row ? P->kv_row[b] : b) * Nk;
  int xr = (P->kv_extra && P->extra_row) ? P->extra_row[b] : -1;
  if (xr >= 0 && P->extra_step) xr = P->extra_step[0];
  Raw8<T> kraw[MAXVA], vraw[MAXVA], qraw;
  const T* kadr[MAXVA];
  const T* vadr[MAXVA];
  bool kvok[MAXVA];
  {
    const int ld_extra = P->ld_extra, kx_off = P->kx_off, vx_off = P->vx_off, k_off = P->k_off, v_off = P->v_off;
#pragma unroll
    for (int i = 0; i < MAXVA; ++i) {
      const int idx = tid + i * NT;
      kvok[i] = idx < nkv;
      const int r = idx >> log2_vpr, c = (idx & (vpr - 1)) * 8;
      const bool ex = (xr >= 0 && r == Nk - 1);
      kadr[i] = ex ? xp_ + ((size_t)((unsigned)xr * (unsigned)ld_extra) + (unsigned)(kx_off + hd + c))
                   : kp_ + ((size_t)((unsigned)(kvbase + r) * (unsigned)ldkv) + (unsigned)(k_off + hd + c));
      vadr[i] = ex ? xp_ + ((size_t)((unsigned)xr * (unsigned)ld_extra) + (unsigned)(vx_off + hd + c))
                   : vp_ + ((size_t)((unsigned)(kvbase + r) * (unsigned)ldkv) + (unsigned)(v_off + hd + c));
      if (!kv_live) {
        if (kvok[i]) { ld_plain(kraw[i], kadr[i]); ld_plain(vraw[i], vadr[i]); }
        else { zero_raw(kraw[i]); zero_raw(vraw[i]); }
      }
    }
  }
  float uk[8], bk[8], uv[8], bv[8], uq[8], bq[8];
  const int kr0 = tid >> log2_vpr, kc0 = (tid & (vpr - 1)) * 8;            // vector 0 of this thread (the only one with a K/V finish)
  if (fin_kv && tid < nkv) {
    load8(P->ln_u + P->k_off + hd + kc0, uk);
    load8(P->ln_b + P->k_off + hd + kc0, bk);
    load8(P->ln_u + P->v_off + hd + kc0, uv);
    load8(P->ln_b + P->v_off + hd + kc0, bv);
  }
  if (fin_q && tid < nqv) {
    load8(P->ln_u + q_off + hd + kc0, uq);
    load8(P->ln_b + q_off + hd + kc0, bq);
  }

  DK_STAMP(sy, 1);
  if (need_wait) wait_phase(sy, sy.p - 1, dep_units, tid);
  DK_STAMP(sy, 2);

  // ---- live operands -------------------------------------------------------------------------------------------------
  if (kv_live) {
#pragma unroll
    for (int i = 0; i < MAXVA; ++i) {
      if (kvok[i]) { ld_live(kraw[i], kadr[i]); ld_live(vraw[i], vadr[i]); }
      else { zero_raw(kraw[i]); zero_raw(vraw[i]); }
    }
  }
  if (tid < nqv) ld_live(qraw, qp + ((size_t)((unsigned)(b * Nq + q0 + kr0) * (unsigned)ldq) + (unsigned)(q_off + hd + kc0)));
  else zero_raw(qraw);
  // LayerNorm statistics of the rows the finish needs: 16 lanes per row over the ln_C leading columns of q's tensor
  if (fin_q || fin_kv) {
    const int rs0 = fin_kv ? 0 : q0, rsn = fin_kv ? Nk : nq;
    const int nvec = ln_C >> 3;
    const float inv_c = 1.0f / (float)ln_C;
    for (int r = tid >> 4; r < rsn; r += NT / 16) {
      const T* rowp = qp + (size_t)((unsigned)(b * Nq + rs0 + r) * (unsigned)ldq);
      float s = 0.f, q2 = 0.f;
#pragma unroll 4
      for (int vv = li; vv < nvec; vv += 16) {
        Raw8<T> x;
        ld_live(x, rowp + vv * 8);
        float f[8];
        raw_to_float(x, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s += f[j]; q2 += f[j] * f[j]; }
      }
      s = row16_sum_d(s);
      q2 = row16_sum_d(q2);
      if (li == 0) {
        const float mean = s * inv_c;
        float var = q2 * inv_c - mean * mean;
        var = var < 0.f ? 0.f : var;
        const float rstd = PRECISE ? 1.0f / sqrtf(var + ln_eps) : rsqrtf(var + ln_eps);
        st_s[r] = make_float2(mean, rstd);
      }
    }
  }
  // zero the padding the matrix cores will read
  {
    const float z8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (d < 32) {
      for (int i = tid; i < (QCHUNK + NKP) * (dq >> 3); i += NT) store8(q_s + i * 8, z8);
    } else {
      for (int i = tid; i < (QCHUNK - nq) * vpr; i += NT) store8(q_s + (nq + (i >> log2_vpr)) * dq + (i & (vpr - 1)) * 8, z8);
      for (int i = tid; i < (NKP - Nk) * vpr; i += NT) store8(kv_s + (Nk + (i >> log2_vpr)) * dq + (i & (vpr - 1)) * 8, z8);
    }
  }
  __syncthreads();
  auto finish = [&](float (&x)[8], const float2 st, const float (&uu)[8], const float (&bb)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (x[e] - st.x * uu[e]) * st.y + bb[e];
  };
  // ---- K, Q -> LDS ---------------------------------------------------------------------------------------------------
#pragma unroll
  for (int i = 0; i < MAXVA; ++i) {
    const int idx = tid + i * NT;
    if (idx < nkv) {
      const int r = idx >> log2_vpr, c = (idx & (vpr - 1)) * 8;
      float x[8];
      raw_to_float(kraw[i], x);
      if (i == 0 && fin_kv) finish(x, st_s[r], uk, bk);
      store8(kv_s + r * dq + c, x);
    }
  }
  if (tid < nqv) {
    float x[8];
    raw_to_float(qraw, x);
    if (fin_q) finish(x, st_s[fin_kv ? q0 + kr0 : kr0], uq, bq);
    store8(q_s + kr0 * dq + kc0, x);
  }
  __syncthreads();
  DK_STAMP(sy, 3);
  // ---- scores on the matrix cores: wave w takes key tiles w, w + NW, ... for both query tiles ------------------------------------
  {
    const int nkt = NKP >> 4;
    const int nqt = (nq + 15) >> 4;
    for (int kt = wave; kt < nkt; kt += NW) {
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      for (int c = 0; c < DP; c += 32) {
        Frag kb;
        dlds(kb, kv_s + (kt * 16 + li) * dq + c + lg * 8);
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
          if (qt < nqt) {
            Frag qa;
            dlds(qa, q_s + (qt * 16 + li) * dq + c + lg * 8);
            dmma(acc[qt], qa, kb);
          }
        }
      }
      const int j = kt * 16 + li;
      const float NEG = -3.402823466e+38f;
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        if (qt < nqt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int qi = qt * 16 + lg * 4 + r;
            const int lim = (q0 + qi) + (Nk - Nq);      // causal: keep j <= i + (Nk - Nq)  (blocks.py:315-319)
            s_s[qi * sp + j] = (causal && j > lim) ? NEG : acc[qt][r] * scale;
          }
        }
      }
    }
  }
  __syncthreads();
  // ---- V replaces K in LDS, transposed: V^T [d][keys] is the B operand of P V ---------------------------------------------------
  if (d < 16) {
    for (int i = tid; i < DC * vt; i += NT) kv_s[i] = (T)0.f;
    __syncthreads();
  } else {
    for (int i = tid; i < d * 32; i += NT) {
      const int c = i >> 5, j = Nk + (i & 31);
      if (j < NKP) kv_s[c * vt + j] = (T)0.f;
    }
  }
#pragma unroll
  for (int i = 0; i < MAXVA; ++i) {
    const int idx = tid + i * NT;
    if (idx < nkv) {
      const int r = idx >> log2_vpr, c = (idx & (vpr - 1)) * 8;
      float x[8];
      raw_to_float(vraw[i], x);
      if (i == 0 && fin_kv) finish(x, st_s[r], uv, bv);
#pragma unroll
      for (int e = 0; e < 8; ++e) kv_s[(c + e) * vt + r] = (T)x[e];
    }
  }
  // ---- softmax in float32 (blocks.py:367-371): 16 lanes per row, 32 rows per pass over the 8 waves ---------------------------------
  {
    const int rsel = lane >> 4;
    const int kpl = ((NKP >> 4) + 3) & ~3;
    const int j0 = li * kpl;
    const int r = wave * 4 + rsel;
    if (r < QCHUNK) {
      T* pr = p_s + r * vt;
      if (r >= nq) {
        for (int j = li; j < NKP; j += 16) pr[j] = (T)0.f;
      } else {
        const float* sr = s_s + r * sp + j0;
        float4 x[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) x[t] = (4 * t < kpl && j0 + 4 * t < NKP) ? *reinterpret_cast<const float4*>(sr + 4 * t) : make_float4(0.f, 0.f, 0.f, 0.f);
        float e[12];
#pragma unroll
        for (int t = 0; t < 3; ++t) { e[4 * t] = x[t].x; e[4 * t + 1] = x[t].y; e[4 * t + 2] = x[t].z; e[4 * t + 3] = x[t].w; }
        float mx = -3.402823466e+38f;
#pragma unroll
        for (int t = 0; t < 12; ++t) {
          const bool in = t < kpl && j0 + t < Nk;
          e[t] = in ? e[t] : -3.402823466e+38f;
          mx = fmaxf(mx, e[t]);
        }
        mx = row16_max_d(mx);
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 12; ++t) {
          const bool in = t < kpl && j0 + t < Nk;
          e[t] = in ? (PRECISE ? expf(e[t] - mx) : __expf(e[t] - mx)) : 0.f;
          sum += e[t];
        }
        sum = row16_sum_d(sum);
        const float inv = PRECISE ? 1.0f / sum : __builtin_amdgcn_rcpf(sum);
#pragma unroll
        for (int t = 0; t < 12; ++t) {
          if (t < kpl && j0 + t < NKP) pr[j0 + t] = (T)(e[t] * inv);
        }
      }
    }
  }
  __syncthreads();
  // ---- out = P V on the matrix cores; the tile goes through LDS (Q's place) so that it leaves as 16-byte write-through stores ----
  {
    const int nqt = (nq + 15) >> 4;
    const int nct = DC >> 4;
    for (int t = wave; t < nqt * nct; t += NW) {
      const int qt = t / nct, ct = t - qt * nct;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int j = 0; j < NKP; j += 32) {
        Frag pa, vb;
        dlds(pa, p_s + (qt * 16 + li) * vt + j + lg * 8);
        dlds(vb, kv_s + (ct * 16 + li) * vt + j + lg * 8);
        dmma(acc, pa, vb);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = qt * 16 + lg * 4 + r;
        q_s[qi * dq + ct * 16 + li] = (T)acc[r];
      }
    }
  }
  publish_next();
  __syncthreads();
  DK_STAMP(sy, 4);
  if (tid < nqv) {
    T* op = outp + ((size_t)((unsigned)(b * Nq + q0 + kr0) * (unsigned)ldo) + (unsigned)(hd + kc0));
    st_live8(op, q_s + kr0 * dq + kc0);
  }
  drain_stores();
  __syncthreads();
  DK_STAMP(sy, 5);
  arrive_phase(sy, tid);
  prefill_next();
  DK_STAMP(sy, 6);
}

// =====================================================================================================================
// statistics unit: GroupNorm fine-group (sum, sumsq) of one batch element of the chain's last tensor, for the
// launch-per-layer consumer that follows the persistent launch (same layout as jen1_conv_args.gn_stats*)
// =====================================================================================================================
template <typename T, typename FPub, typename FPre>
__device__ __forceinline__ void stats_unit(const unsigned char* D, int u, Sync& sy, bool need_wait, int dep_units, FPub publish_next,
                                           FPre prefill_next, int tid) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const jen1_deep_phase* P = reinterpret_cast<const jen1_deep_phase*>(D);
  const int b = u, L = P->sL, ld = P->sld, cpf = P->scpf, gran = P->sgran;
  const T* sx = reinterpret_cast<const T*>(P->sx);
  float* const sstats = P->sstats;
  DK_STAMP(sy, 0);
  DK_STAMP(sy, 1);
  if (need_wait) wait_phase(sy, sy.p - 1, dep_units, tid);
  DK_STAMP(sy, 2);
  const int VPR = ld >> 3;                       // power of two <= NT (checked on the host)
  const int vc = tid & (VPR - 1), r0 = tid / VPR, rstep = NT / VPR;
  const int sub = 8 / gran;                      // granules per vector (1, 2, 4 or 8)
  float s[8], q[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { s[k] = 0.f; q[k] = 0.f; }
  const T* xp = sx + (size_t)b * L * ld + vc * 8;
  for (int r = r0; r < L; r += rstep) {
    Raw8<T> x;
    ld_live(x, xp + (size_t)r * ld);
    float f[8];
    raw_to_float(x, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) { s[j] += f[j]; q[j] += f[j] * f[j]; }
  }
  if (gran >= 2) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) { s[j] += s[j + 1]; q[j] += q[j + 1]; }
  }
  if (gran >= 4) {
#pragma unroll
    for (int j = 0; j < 8; j += 4) { s[j] += s[j + 2]; q[j] += q[j + 2]; }
  }
  if (gran >= 8) { s[0] += s[4]; q[0] += q[4]; }
  const int lg2 = gran >= 8 ? 3 : (gran >= 4 ? 2 : (gran >= 2 ? 1 : 0));
  float2* part = reinterpret_cast<float2*>(smem + WS_OFF);          // [rstep][VPR * sub]
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if ((j & (gran - 1)) == 0) part[(size_t)r0 * (VPR * sub) + vc * sub + (j >> lg2)] = make_float2(s[j], q[j]);
  }
  publish_next();
  __syncthreads();
  DK_STAMP(sy, 3);
  // one quad per fine group: elements = rstep row groups x (cpf / gran) granules, fixed order
  const int gpf = cpf / gran;
  const int j4 = tid & 3;
  for (int fg = tid >> 2; fg < JEN1_FINE_GROUPS; fg += NT / 4) {
    const int n_el = rstep * gpf;
    float ss = 0.f, qq = 0.f;
    for (int e = j4; e < n_el; e += 4) {
      const int rr = e / gpf, gi = e - rr * gpf;
      const float2 v = part[(size_t)rr * (VPR * sub) + fg * gpf + gi];
      ss += v.x;
      qq += v.y;
    }
    ss = quad_sum(ss);
    qq = quad_sum(qq);
    if (j4 == 0) {
      sstats[(size_t)b * 64 + fg * 2] = ss;
      sstats[(size_t)b * 64 + fg * 2 + 1] = qq;
    }
  }
  DK_STAMP(sy, 4);
  drain_stores();
  __syncthreads();
  DK_STAMP(sy, 5);
  arrive_phase(sy, tid);
  prefill_next();
  DK_STAMP(sy, 6);
}

template <typename T>
__global__ __launch_bounds__(NT) void deep_kernel(const unsigned char* __restrict__ blobs, const int4* __restrict__ hdr_g, int n_phases,
                                                  unsigned* sync, int err_word) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef typename DFrag<T>::type Frag;
  constexpr int PF = DeepCfg<T>::PF;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wk = rfl(tid >> 6);
  const int wg = blockIdx.x, nwg = gridDim.x;
  Hdr* hdr = reinterpret_cast<Hdr*>(smem);
  if (tid < n_phases) reinterpret_cast<int4*>(smem)[tid] = hdr_g[tid];
  __syncthreads();
  Sync sy;
  sy.base = sync;
  sy.err = sync + err_word;
  sy.dead = false;
  sy.wg = wg;
  sy.nwg = nwg;
  int p = -1, u = 0;
  if (!find_next(hdr, n_phases, wg, nwg, p, u)) return;
  p = rfl(p);
  u = rfl(u);
  int slot = 0;
  reinterpret_cast<u64*>(smem + HDR_BYTES)[tid] = reinterpret_cast<const u64*>(blobs + (size_t)p * BLOB)[tid];
  __syncthreads();
  Frag ra[PF];
  bool prefilled = false;
  int waited = -1;
  for (;;) {
    int p2 = p, u2 = u;
    const bool more = find_next(hdr, n_phases, wg, nwg, p2, u2);
    p2 = rfl(p2);
    u2 = rfl(u2);
    const bool reload = more && p2 != p;
    u64 nx = 0;
    if (reload) nx = reinterpret_cast<const u64*>(blobs + (size_t)p2 * BLOB)[tid];       // in flight during the unit
    const unsigned char* D = smem + HDR_BYTES + slot * BLOB;
    unsigned char* Dn = smem + HDR_BYTES + (reload ? slot ^ 1 : slot) * BLOB;
    const bool next_gemm = more && hdr[p2].kind == JEN1_DEEP_GEMM;
    bool next_prefilled = false;
    // called by the unit right before one of its __syncthreads() / after it
    auto publish_next = [&]() { if (reload) reinterpret_cast<u64*>(Dn)[tid] = nx; };
    auto prefill_next = [&]() {
      if (next_gemm) {
        gemm_prefill<T>(Dn, u2, wk, lane, ra);
        next_prefilled = true;
      }
    };
    sy.p = p;
    const bool need_wait = waited != p && p > 0;
    waited = p;
    const int dep_units = p > 0 ? hdr[p - 1].n_units : 0;
    const int kind = hdr[p].kind;
    if (kind == JEN1_DEEP_GEMM) gemm_unit<T>(D, u, sy, need_wait, dep_units, ra, prefilled, publish_next, prefill_next, tid);
    else if (kind == JEN1_DEEP_ATTN) attn_unit<T>(D, u, sy, need_wait, dep_units, publish_next, prefill_next, tid);
    else stats_unit<T>(D, u, sy, need_wait, dep_units, publish_next, prefill_next, tid);
    if (!more) break;
    p = p2;
    u = u2;
    if (reload) slot ^= 1;
    prefilled = next_prefilled;
  }
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int align16i(int x) { return (x + 15) & ~15; }
constexpr int LDS_TOTAL = 160 * 1024;
constexpr int LDS_BUDGET = LDS_TOTAL - WS_OFF;      // what a unit may use behind the headers and the two descriptor slots

}  // namespace

extern "C" int jen1_deep_phase_size(void) { return (int)sizeof(jen1_deep_phase); }
extern "C" int jen1_deep_blob_bytes(void) { return BLOB; }

#ifdef JEN1_DEEP_PROFILE
// tuning builds: [n_phases][nwg][8] uint64 stamps (see DK_STAMP); not part of the product ABI
extern "C" int jen1_deep_debug_buffer(void* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_deep_dbg), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int jen1_deep_phase_conv(const jen1_conv_args* a, int nb_max, jen1_deep_phase* out) {
  JEN1_CHECK(a && out, "deep conv: null pointer");
  JEN1_CHECK(a->dtype == JEN1_F32 || a->dtype == JEN1_BF16, "deep conv: bad dtype");
  JEN1_CHECK(a->x0 && a->w && a->y, "deep conv: null tensor");
  JEN1_CHECK(a->pro_mode == JEN1_PRO_NONE || a->pro_mode == JEN1_PRO_GN || a->pro_mode == JEN1_PRO_GN_SILU,
             "deep conv: prologue %d is not supported by the persistent kernel", a->pro_mode);
  JEN1_CHECK(!a->ln_fold && !a->row_scale, "deep conv: LayerNorm-fold epilogue / row scale are not supported by the persistent kernel");
  JEN1_CHECK(a->c0 > 0 && a->c0 % 32 == 0 && a->c1 % 32 == 0 && a->M % 16 == 0, "deep conv: channels must be multiples of 32, M of 16");
  JEN1_CHECK(a->taps >= 1 && a->stride >= 1 && a->B >= 1 && a->L_in >= 1 && a->L_out >= 1, "deep conv: bad geometry");
  JEN1_CHECK(a->nseg >= 0 && a->nseg <= JEN1_DEEP_MAX_SRC - 2, "deep conv: at most %d extra K segments", JEN1_DEEP_MAX_SRC - 2);
  JEN1_CHECK(a->taps + a->nseg <= JEN1_DEEP_MAX_SEG, "deep conv: too many K segments");
  const int es = a->dtype == JEN1_F32 ? 4 : 2;
  const int maxv = a->dtype == JEN1_F32 ? JEN1_DEEP_MAXV_F : JEN1_DEEP_MAXV_B;
  jen1_deep_phase& p = *out;
  memset(&p, 0, sizeof(p));
  p.kind = JEN1_DEEP_GEMM;
  p.dtype = a->dtype;
  p.dep = -1;
  // ---- sources and segments ------------------------------------------------------------------------------------------------
  int ns = 0, coff = 0;
  p.src[ns++] = jen1_deep_src{a->x0, a->ld0, a->c0, 0, 1.0f};
  coff = a->c0;
  if (a->c1) {
    JEN1_CHECK(a->x1, "deep conv: c1 without x1");
    p.src[ns++] = jen1_deep_src{a->x1, a->ld1, a->c1, coff, a->src1_scale};
    coff += a->c1;
  }
  const int cmain = coff;
  int G = 0, nseg = 0;
  for (int tap = 0; tap < a->taps; ++tap) {
    G += cmain / 32;
    p.seg[nseg++] = jen1_deep_seg{0, tap - a->pad_left, G, 0};
  }
  for (int s = 0; s < a->nseg; ++s) {
    const jen1_conv_seg& e = a->seg[s];
    JEN1_CHECK(e.x && e.kch > 0 && e.ld >= 32 * e.kch, "deep conv: bad extra segment %d", s);
    p.src[ns++] = jen1_deep_src{e.x, e.ld, 32 * e.kch, coff, 1.0f};
    G += e.kch;
    p.seg[nseg++] = jen1_deep_seg{coff, e.shift, G, 0};
    coff += 32 * e.kch;
  }
  p.nsrc = ns; p.nseg = nseg; p.G = G; p.Ctot = coff;
  p.pitch = coff + 8;
  p.MT = a->M / 16;
  const int64_t wb = (int64_t)G * p.MT * 512 * es;
  JEN1_CHECK(wb < ((int64_t)1 << 31), "deep conv: packed weight too large for 31-bit offsets");
  p.w = a->w; p.w_bytes = (uint32_t)wb;
  p.mt_split = a->m_split / 16;
  p.g_split = a->m_split ? a->k_split : 0;
  JEN1_CHECK(a->m_split % 16 == 0 && p.g_split <= G, "deep conv: bad dual-range split");
  p.B = a->B; p.L_in = a->L_in; p.L_out = a->L_out; p.stride = a->stride;
  // ---- prologue ---------------------------------------------------------------------------------------------------------------
  p.pro_mode = a->pro_mode;
  if (a->pro_mode != JEN1_PRO_NONE) {
    JEN1_CHECK(a->gn_gamma && a->gn_beta && a->gn_groups >= 1 && a->gn_cpg >= 1 && a->gn_count >= 1, "deep conv: incomplete GroupNorm");
    p.norm_C = cmain;
    p.gn_groups = a->gn_groups;
    p.gn_cpg = a->gn_groups == 1 ? cmain : a->gn_cpg;
    p.gran = p.gn_cpg >= 8 ? 8 : p.gn_cpg;
    JEN1_CHECK((p.gn_cpg >= 8 && p.gn_cpg % 8 == 0) || p.gn_cpg == 4 || p.gn_cpg == 2 || p.gn_cpg == 1, "deep conv: group size %d", p.gn_cpg);
    JEN1_CHECK(a->gn_groups * p.gn_cpg <= cmain, "deep conv: groups exceed the channels");
    p.inv_count = 1.0f / (float)a->gn_count;
    p.gn_eps = a->gn_eps;
    p.inv_groups = 1.0f / (float)a->gn_groups;
    if (a->film) {
      // `film` is the FUSED table here: gamma * (scale + 1) at film_off + c, beta * (scale + 1) + shift at film_off + film_C + c
      JEN1_CHECK(a->film_C == cmain && a->film_ld >= a->film_off + 2 * a->film_C, "deep conv: bad FiLM table geometry");
      p.p1 = a->film + a->film_off;
      p.p2 = a->film + a->film_off + a->film_C;
      p.p_ld = a->film_ld;
      p.film_row = a->film_row; p.film_step = a->film_step;
    } else {
      p.p1 = a->gn_gamma; p.p2 = a->gn_beta; p.p_ld = 0;
    }
  }
  // ---- epilogue ----------------------------------------------------------------------------------------------------------------
  p.bias = a->bias; p.residual = a->residual; p.y = a->y;
  p.out_C = a->out_C; p.ps_f = a->ps_f < 1 ? 1 : a->ps_f; p.ps_off = a->ps_off; p.L_y = a->L_y; p.y_brows = a->y_brows;
  p.y_row0 = a->y_row0; p.ld_y = a->ld_y; p.ld_res = a->ld_res; p.act = a->act; p.y_f32 = a->y_f32;
  JEN1_CHECK(a->out_C % 4 == 0 && a->ld_y % 4 == 0 && (!a->residual || a->ld_res % 4 == 0), "deep conv: output channels / pitches must be multiples of 4");
  JEN1_CHECK((int64_t)a->B * a->L_in * (a->ld0 > a->ld1 ? a->ld0 : a->ld1) < ((int64_t)1 << 31) && (int64_t)a->B * a->y_brows * a->ld_y < ((int64_t)1 << 31),
             "deep conv: tensor too large");
  // ---- unit geometry: as many batch elements per unit as fit (fewer, fatter units re-read the weights less) --------------------------
  int nb = a->B;
  if (nb_max > 0 && nb > nb_max) nb = nb_max;
  for (;; --nb) {
    JEN1_CHECK(nb >= 1, "deep conv: one batch element (%d rows x %d channels) does not fit a unit", a->L_in, coff);
    const int cols = nb * a->L_out;
    const int NF = ceil_div(cols, 16);
    const int R = nb * a->L_in;
    if (NF > 4) continue;
    if ((int64_t)R * (p.norm_C / 8) > (int64_t)maxv * JEN1_DEEP_THREADS) continue;       // normalised vectors per thread
    const int tile_b = align16i((R + 1) * p.pitch * es);
    const int part_b = p.norm_C ? align16i(R * (p.norm_C / p.gran) * 8) : 0;
    const int stat_b = p.norm_C ? align16i(nb * p.gn_groups * 8) : 0;
    const int red_b = (JEN1_DEEP_THREADS / 64) * NF * 1024;
    const int tot = tile_b + stat_b + (part_b > red_b ? part_b : red_b);
    if (tot > LDS_BUDGET) continue;
    p.nb = nb; p.NF = NF; p.R = R;
    p.stat_off = tile_b;
    p.part_off = tile_b + stat_b;
    p.red_off = tile_b + stat_b;        // the partial sums are dead when the K reduction starts
    p.lds_bytes = tot;
    break;
  }
  p.groups_n = ceil_div(a->B, p.nb);
  p.n_units = p.MT * p.groups_n;
  JEN1_CHECK(p.n_units < (1 << 20), "deep conv: too many units");
  p.inv_vpr = 1.0f / (float)(coff / 8);
  p.inv_Lin = 1.0f / (float)a->L_in;
  p.inv_Lout = 1.0f / (float)a->L_out;
  return 0;
}

extern "C" int jen1_deep_phase_attention(const void* q, const void* k, const void* v, void* out_t, const int32_t* kv_row, const void* kv_extra,
                                         const int32_t* extra_row, const int32_t* extra_step, int ld_extra, int kx_off, int vx_off, int B,
                                         int H, int d, int Nq, int Nk, int ldq, int q_off, int ldkv, int k_off, int v_off, int ldo,
                                         int causal, float scale, const float* ln_u, const float* ln_b, int ln_C, float ln_eps,
                                         int finish_q, int finish_kv, int kv_live, int dtype, jen1_deep_phase* out) {
  JEN1_CHECK(q && k && v && out_t && out, "deep attention: null pointer");
  JEN1_CHECK(dtype == JEN1_F32 || dtype == JEN1_BF16, "deep attention: bad dtype");
  JEN1_CHECK(B >= 1 && H >= 1 && Nq >= 1 && Nk >= 1, "deep attention: bad sizes");
  JEN1_CHECK(d == 8 || d == 16 || d == 32 || d == 64 || d == 128, "deep attention: head dim %d must be 8, 16, 32, 64 or 128", d);
  JEN1_CHECK(Nk <= 192 && Nk * (d / 8) <= ((141 * 16 + JEN1_DEEP_THREADS - 1) / JEN1_DEEP_THREADS) * JEN1_DEEP_THREADS, "deep attention: Nk=%d d=%d outside the kernel's range", Nk, d);
  JEN1_CHECK(ldq % 8 == 0 && ldkv % 8 == 0 && ldo % 8 == 0 && q_off % 8 == 0 && k_off % 8 == 0 && v_off % 8 == 0 &&
             (!kv_extra || (ld_extra % 8 == 0 && kx_off % 8 == 0 && vx_off % 8 == 0)), "deep attention: offsets / strides must be multiples of 8 elements");
  const bool fin = finish_q || finish_kv;
  JEN1_CHECK(!fin || (ln_u && ln_b && ln_C >= 8 && ln_C % 8 == 0), "deep attention: a LayerNorm finish needs u, bias and ln_C");
  JEN1_CHECK(!finish_kv || (!kv_row && !kv_extra && Nk * (d / 8) <= JEN1_DEEP_THREADS && Nq == Nk && kv_live),
             "deep attention: the K/V finish is for self-attention over at most %d vectors", JEN1_DEEP_THREADS);
  const int es = dtype == JEN1_F32 ? 4 : 2;
  const int DP = d < 32 ? 32 : d, DC = d < 16 ? 16 : d;
  const int NKP = (Nk + 31) & ~31, dq = DP + 8, vt = NKP + 8, sp = NKP + 4;
  const int64_t kv_elems = (NKP * dq > DC * vt) ? NKP * dq : DC * vt;
  const int st_rows = Nk > QCHUNK ? Nk : QCHUNK;
  const int64_t lds = (int64_t)es * QCHUNK * dq + es * ((kv_elems + 7) & ~(int64_t)7) + 4 * (((int64_t)QCHUNK * sp + 3) & ~(int64_t)3) +
                      es * (((int64_t)QCHUNK * vt + 7) & ~(int64_t)7) + 8 * st_rows;
  JEN1_CHECK(lds <= LDS_BUDGET, "deep attention: Nk=%d d=%d needs %lld B of LDS", Nk, d, (long long)lds);
  jen1_deep_phase& p = *out;
  memset(&p, 0, sizeof(p));
  p.kind = JEN1_DEEP_ATTN;
  p.dtype = dtype;
  p.dep = -1;
  p.lds_bytes = align16i((int)lds);
  p.q = q; p.k = k; p.v = v; p.out = out_t; p.kv_row = kv_row; p.kv_extra = kv_extra; p.extra_row = extra_row; p.extra_step = extra_step;
  p.ln_u = ln_u; p.ln_b = ln_b;
  p.ld_extra = ld_extra; p.kx_off = kx_off; p.vx_off = vx_off; p.B = B; p.H = H; p.d = d; p.Nq = Nq; p.Nk = Nk; p.ldq = ldq; p.q_off = q_off;
  p.ldkv = ldkv; p.k_off = k_off; p.v_off = v_off; p.ldo = ldo; p.causal = causal;
  p.ln_C = ln_C; p.fin_q = finish_q; p.fin_kv = finish_kv; p.kv_live = kv_live;
  p.nqc = ceil_div(Nq, QCHUNK);
  int l2 = 0;
  while ((8 << l2) < d) ++l2;
  p.log2_vpr = l2;
  p.scale = scale; p.ln_eps = ln_eps; p.inv_H = 1.0f / (float)H; p.inv_nqc = 1.0f / (float)p.nqc;
  p.n_units = B * H * p.nqc;
  JEN1_CHECK(p.n_units < (1 << 20), "deep attention: too many units");
  return 0;
}

extern "C" int jen1_deep_phase_stats(const void* x, float* stats, int B, int L, int ld, int dtype, jen1_deep_phase* out) {
  JEN1_CHECK(x && stats && out && B >= 1 && L >= 1, "deep stats: bad arguments");
  JEN1_CHECK(dtype == JEN1_F32 || dtype == JEN1_BF16, "deep stats: bad dtype");
  JEN1_CHECK(ld >= 32 && (ld & (ld - 1)) == 0 && ld / 8 <= JEN1_DEEP_THREADS, "deep stats: row pitch %d must be a power of two in [32, 8192]", ld);
  jen1_deep_phase& p = *out;
  memset(&p, 0, sizeof(p));
  p.kind = JEN1_DEEP_STATS;
  p.dtype = dtype;
  p.dep = -1;
  p.sx = x; p.sstats = stats; p.sL = L; p.sld = ld;
  p.scpf = ld / JEN1_FINE_GROUPS;
  p.sgran = p.scpf >= 8 ? 8 : p.scpf;
  p.B = B;
  p.n_units = B;
  const int vpr = ld / 8, sub = 8 / p.sgran;
  p.lds_bytes = align16i((JEN1_DEEP_THREADS / vpr) * vpr * sub * 8);
  return 0;
}

extern "C" int jen1_deep_link(jen1_deep_phase* phases, int n_phases, int nwg, void* blobs, void* headers) {
  JEN1_CHECK(phases && blobs && headers && n_phases >= 1 && nwg >= 1, "deep link: bad arguments");
  JEN1_CHECK(n_phases <= JEN1_DEEP_MAX_PHASES, "deep link: %d phases (at most %d)", n_phases, JEN1_DEEP_MAX_PHASES);
  int lds = 0, rot = 0;
  unsigned char* bl = reinterpret_cast<unsigned char*>(blobs);
  int32_t* hd = reinterpret_cast<int32_t*>(headers);
  memset(bl, 0, (size_t)n_phases * BLOB);
  for (int p = 0; p < n_phases; ++p) {
    jen1_deep_phase& P = phases[p];
    P.dep = p - 1;
    P.dep_units = p ? phases[p - 1].n_units : 0;
    // successive phases start their units on successive workgroups (multiples of 8 keep a unit's XCD = its M tile mod 8):
    // a workgroup that just finished a unit is rarely the one the next phase waits for, so it has the time of a few
    // phases to pull the weight slice of its next unit, and the weight streams spread over all CUs
    P.rot = rot % nwg;
    rot += ((P.n_units + 7) / 8) * 8;
    lds = P.lds_bytes > lds ? P.lds_bytes : lds;
    unsigned char* b = bl + (size_t)p * BLOB;
    memcpy(b, &P, sizeof(P));
    hd[4 * p + 0] = P.n_units; hd[4 * p + 1] = P.rot; hd[4 * p + 2] = P.kind; hd[4 * p + 3] = 0;
    if (P.kind != JEN1_DEEP_GEMM) continue;
    // K chunks that can touch a real input row (a segment whose every row is conv padding for every position is skipped
    // together with its weights: exact), dealt round-robin to the waves
    int16_t* cnt = reinterpret_cast<int16_t*>(b + TAB_OFF);
    uint32_t* ent = reinterpret_cast<uint32_t*>(b + TAB_OFF + 32);
    const int tmax = (P.L_out - 1) * P.stride;
    int k = 0;
    for (int s = 0; s < P.nseg; ++s) {
      const int sb = s ? P.seg[s - 1].gend : 0, se = P.seg[s].gend, sh = P.seg[s].shift;
      if (!(tmax + sh >= 0 && sh < P.L_in)) continue;
      JEN1_CHECK(sh >= -128 && sh <= 127, "deep link: shift %d out of range", sh);
      for (int g = sb; g < se; ++g, ++k) {
        const int w = k % NW, j = k / NW;
        JEN1_CHECK(j < MAXE, "deep link: phase %d has more than %d K chunks per wave", p, MAXE);
        const int col = P.seg[s].coff + (g - sb) * 32;
        JEN1_CHECK(col < 65536, "deep link: staged column %d out of range", col);
        ent[(w * MAXE + j) * 2 + 0] = (uint32_t)g;
        ent[(w * MAXE + j) * 2 + 1] = (uint32_t)col | ((uint32_t)(uint8_t)(int8_t)sh << 16);
        cnt[w] = (int16_t)(j + 1);
        if (P.mt_split && g < P.g_split) cnt[NW + w] = (int16_t)(j + 1);
      }
    }
    if (!P.mt_split) for (int w = 0; w < NW; ++w) cnt[NW + w] = cnt[w];
  }
  return lds + WS_OFF;
}

extern "C" int64_t jen1_deep_sync_bytes(int n_phases) { return ((int64_t)n_phases * SHARDS * SHW + SHW) * 4; }
extern "C" int jen1_deep_error_word(int n_phases) { return n_phases * SHARDS * SHW; }

extern "C" int jen1_deep_num_workgroups(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
  return cus;
}

extern "C" int jen1_deep_run(const void* blobs_dev, const void* headers_dev, int n_phases, uint32_t* sync, int nwg, int lds_bytes, int dtype,
                             void* stream) {
  JEN1_CHECK(blobs_dev && headers_dev && sync && n_phases >= 1 && n_phases <= JEN1_DEEP_MAX_PHASES && nwg >= 1, "deep run: bad arguments");
  JEN1_CHECK(lds_bytes >= WS_OFF && lds_bytes <= LDS_TOTAL, "deep run: %d B of LDS", lds_bytes);
  JEN1_CHECK(dtype == JEN1_F32 || dtype == JEN1_BF16, "deep run: bad dtype");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int dev = 0;
  JEN1_HIP(hipGetDevice(&dev));
  static unsigned long long attr_set[2] = {0ull, 0ull};       // per dtype, one bit per device ordinal
  const void* fn = dtype == JEN1_F32 ? reinterpret_cast<const void*>(deep_kernel<float>) : reinterpret_cast<const void*>(deep_kernel<bf16_t>);
  if (dev >= 64 || !(attr_set[dtype == JEN1_F32 ? 0 : 1] >> dev & 1ull)) {
    JEN1_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL));
    if (dev < 64) attr_set[dtype == JEN1_F32 ? 0 : 1] |= 1ull << dev;
  }
  const int err_word = jen1_deep_error_word(n_phases);
  const unsigned char* bl = reinterpret_cast<const unsigned char*>(blobs_dev);
  const int4* hd = reinterpret_cast<const int4*>(headers_dev);
  if (dtype == JEN1_F32) hipLaunchKernelGGL(deep_kernel<float>, dim3(nwg), dim3(NT), (size_t)lds_bytes, s, bl, hd, n_phases, sync, err_word);
  else hipLaunchKernelGGL(deep_kernel<bf16_t>, dim3(nwg), dim3(NT), (size_t)lds_bytes, s, bl, hd, n_phases, sync, err_word);
  JEN1_HIP(hipGetLastError());
  return 0;
}
