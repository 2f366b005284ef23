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
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "jen1_deep.h"

namespace {

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int NT = JEN1_DEEP_THREADS;
constexpr int NW = NT / 64;
constexpr int SHARDS = JEN1_DEEP_SHARDS;
static_assert(SHARDS >= 1 && SHARDS <= 63, "one wave polls the shards and the error word");
constexpr int SHW = JEN1_DEEP_SHARD_WORDS;
constexpr unsigned OOB = 0x80000000u;
constexpr int RSRC_FLAGS = 0x00020000;
constexpr int QCHUNK = 32;
#ifndef JEN1_DEEP_ATTN_LIVE_VEC
#define JEN1_DEEP_ATTN_LIVE_VEC 2     // K / V vectors per thread of a self-attention (produced inside the launch: held across the polled loads)
#endif

__device__ __forceinline__ gu64* g64(const void* p) { return (gu64*)(u64)p; }
__device__ __forceinline__ gu32* g32(const void* p) { return (gu32*)(u64)p; }

// ---- JEN1_FP8 mode (BASELINE configs[4]: "fp8 MFMA attention path"; blocks.py:355-380, :402-407, :440-446) -------------
// Activations stay bf16 in HBM; what the matrix cores read is OCP e4m3 (gfx950's fp8): the packed weights (one float32 scale
// per output row, applied in the epilogue), the staged activation tile, and Q / K / P / V^T of the attention unit
// (v_mfma_f32_16x16x32_fp8_fp8: 8 bytes of K per lane and operand, same lane -> element map as the bf16 form).
template <typename T> struct Mode { typedef T G; };            // G: element type of activations in global memory
template <> struct Mode<fp8_t> { typedef bf16_t G; };
constexpr float P_SCALE = JEN1_FP8_P_SCALE;

// ---- 8-element vectors through agent-scope (sc1) accesses: data another workgroup produced in THIS launch --------
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> { u64 d[2]; };
template <> struct Raw8<float> { u64 d[4]; };

#ifdef JEN1_DEEP_EXP_PLAIN        // timing experiment only: plain (L1-cached) loads instead of agent-scope ones
#define JEN1_LIVE_LOAD(p) (*(p))
#else
#define JEN1_LIVE_LOAD(p) __hip_atomic_load(p, RLX_AGENT)
#endif
__device__ __forceinline__ void ld_live(Raw8<bf16_t>& r, const bf16_t* p) {
  r.d[0] = JEN1_LIVE_LOAD(g64(p));
  r.d[1] = JEN1_LIVE_LOAD(g64(p) + 1);
}
__device__ __forceinline__ void ld_live(Raw8<float>& r, const float* p) {
#pragma unroll
  for (int i = 0; i < 4; ++i) r.d[i] = JEN1_LIVE_LOAD(g64(p) + i);
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
__device__ __forceinline__ void float_to_raw(const float (&x)[8], Raw8<bf16_t>& r) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    bf16x4 a;
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = (bf16_t)x[4 * i + j];
    r.d[i] = __builtin_bit_cast(u64, a);
  }
}
__device__ __forceinline__ void float_to_raw(const float (&x)[8], Raw8<float>& r) {
#pragma unroll
  for (int i = 0; i < 4; ++i) r.d[i] = ((u64)__float_as_uint(x[2 * i + 1]) << 32) | __float_as_uint(x[2 * i]);
}
// a raw vector into the staged tile: a copy when the tile has the activations' type, a conversion in JEN1_FP8 mode
__device__ __forceinline__ void stage_raw(bf16_t* dst, const Raw8<bf16_t>& r) { *reinterpret_cast<Raw8<bf16_t>*>(dst) = r; }
__device__ __forceinline__ void stage_raw(float* dst, const Raw8<float>& r) { *reinterpret_cast<Raw8<float>*>(dst) = r; }
__device__ __forceinline__ void stage_raw(fp8_t* dst, const Raw8<bf16_t>& r) {
  float x[8];
  raw_to_float(r, x);
  store8(dst, x);
}
// The all-ones 8-byte word is RESERVED (it means "not stored yet", see Sync below).  A finite result never encodes as it; four bf16
// (two float32) NaNs with sign and every mantissa bit set would -- e.g. NaN weights whose payload propagates.  Every live store
// therefore breaks exactly that pattern (the lowest payload bit of the word's first element is cleared: still a NaN, no longer
// the sentinel): the consumer sees NaNs, as the reference's consumer would, and no data a producer can compute makes a consumer
// wait (include/jen1_deep.h "Reserved word").  Three vector instructions per store.
template <typename G>
__device__ __forceinline__ void st_word(G* p, int i, unsigned lo, unsigned hi) {
  lo -= ((lo & hi) == 0xffffffffu) ? 1u : 0u;
  __hip_atomic_store(g64(p) + i, ((u64)hi << 32) | lo, RLX_AGENT);
}
// 4 consecutive output channels of one position, write-through
__device__ __forceinline__ void st_live4(bf16_t* p, const float (&v)[4]) {
  bf16x4 a;
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = (bf16_t)v[i];
  const u32x2 w = __builtin_bit_cast(u32x2, a);
  st_word(p, 0, w[0], w[1]);
}
__device__ __forceinline__ void st_live4(float* p, const float (&v)[4]) {
  st_word(p, 0, __float_as_uint(v[0]), __float_as_uint(v[1]));
  st_word(p, 1, __float_as_uint(v[2]), __float_as_uint(v[3]));
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
  st_word(p, 0, v[0], v[1]);
  st_word(p, 1, v[2], v[3]);
}
__device__ __forceinline__ void st_live8(float* p, const float* s) {
  const u32x4 a = *reinterpret_cast<const u32x4*>(s);
  const u32x4 b = *reinterpret_cast<const u32x4*>(s + 4);
  st_word(p, 0, a[0], a[1]);
  st_word(p, 1, a[2], a[3]);
  st_word(p, 2, b[0], b[1]);
  st_word(p, 3, b[2], b[3]);
}

// ---- MFMA fragments ------------------------------------------------------------------------------------------------
template <typename T> struct DFrag;
template <> struct DFrag<bf16_t> { typedef bf16x8 type; };
template <> struct DFrag<float> { typedef f32x8 type; };
template <> struct DFrag<fp8_t> { typedef long type; };
__device__ __forceinline__ void dmma(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void dmma(f32x4& acc, const f32x8& a, const f32x8& b) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], acc, 0, 0, 0);
}
__device__ __forceinline__ void dmma(f32x4& acc, const long& a, const long& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void dlds(long& f, const fp8_t* p) { f = *reinterpret_cast<const long*>(p); }
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

__device__ __forceinline__ void wload(long& f, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  f = __builtin_bit_cast(long, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 2));
}
__device__ __forceinline__ void frag_zero_d(long& f) { f = 0; }
__device__ __forceinline__ void frag_zero_d(bf16x8& f) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = (bf16_t)0.f;
}
__device__ __forceinline__ void frag_zero_d(f32x8& f) {
#pragma unroll
  for (int j = 0; j < 8; ++j) f.v[j] = 0.f;
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

#ifndef JEN1_DEEP_POLL_SLEEP
#define JEN1_DEEP_POLL_SLEEP 2       // (0 / 1 / 3 / 8 / 20: 1136 / 1125 / 1119 / 1114 / 1126 us per launch) measured: several polls in flight per workgroup (LDS-DMA polls, 2 / 3 / 4 deep) slow the launch by
#endif                               // 4 / 9 / 12 % -- 256 pollers on the counter lines are in the producers' way; see DESIGN.md 4a
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
template <> struct DeepCfg<fp8_t> { static constexpr int MAXV = JEN1_DEEP_MAXV_B, PF = JEN1_DEEP_PF_B; };

// ---- device-resident program: one blob per phase + a header array ----------------------------------------------------
//   blob  [0, sizeof(jen1_deep_phase))            the descriptor
//         [TAB_OFF, +64)                          int16 cnt[NW] (K chunks of wave w), cnt_low[NW] (those below g_split),
//                                                 nruns[NW], nruns_low[NW]
//         [TAB_OFF + 64, ...)                     NW lists of MAXRUN runs {g0, n, col0, shift}: the usable chunks of the flat
//                                                 (segment, chunk) list dealt round-robin; inside a segment a wave's chunks
//                                                 are g0, g0 + NW, ... at staged columns col0, col0 + 32 NW, ...
//         [SLOT_OFF, ...)                         per wave: the flat chunk index g[SLOTS] and the staged offset (shift * pitch + column)
//                                                 off[SLOTS] of its first SLOTS chunks (the first round of the weight ring: no cursor
//                                                 walk in the unit or in the prefill), then per wave the cursor {run, left, g, col,
//                                                 shift} behind the ring's first round (CUR_OFF; PF chunks in, for layers with more)
//   header {n_units, rot, kind, -}                what a workgroup needs to find its next unit without touching the blobs
constexpr int BLOB = JEN1_DEEP_BLOB_BYTES;
constexpr int TAB_OFF = 1024;
constexpr int MAXRUN = 12;                             // runs (segment pieces) per wave
constexpr int SLOTS = 12;                              // ring slots tabulated per wave (>= PF of either dtype)
constexpr int SLOT_OFF = TAB_OFF + 64 + NW * MAXRUN * 16;
constexpr int CUR_OFF = SLOT_OFF + NW * SLOTS * 8;
static_assert(CUR_OFF + NW * 32 <= BLOB, "run / slot / cursor tables must fit the blob");
static_assert(JEN1_DEEP_PF_B <= SLOTS && JEN1_DEEP_PF_F <= SLOTS, "the slot table covers the ring");
constexpr int HDR_BYTES = JEN1_DEEP_MAX_PHASES * 16;
constexpr int TICKET_OFF = HDR_BYTES + 2 * BLOB;       // LDS: headers | two descriptor slots | the next ticket | unit workspace
constexpr int XW_OFF = TICKET_OFF + 16;                  // 8 x (sum, sumsq): lane sets of two waves meet here (groups of 1024 channels)
constexpr int WS_OFF = TICKET_OFF + 16 + 64;
static_assert(sizeof(jen1_deep_phase) <= TAB_OFF, "descriptor must fit ahead of the chunk table");
static_assert(BLOB == NT * 8, "one 8-byte word per thread moves a blob");
static_assert(JEN1_DEEP_MAX_PHASES <= NT, "one header per thread at start-up");
struct Hdr {
  int n_units, rot, kind, before;      // before: units of all earlier phases (the first ticket of this phase)
};

// ---- synchronisation: the data is its own flag ------------------------------------------------------------------------
// Every tensor a phase of the launch produces is POISONED before the launch (all bytes 0xFF: jen1_deep_poison, a node of the
// step's graph well ahead of the launch).  A producer stores its results as 8-byte single-copy-atomic write-through words and does
// nothing else: no drain, no arrival counter.  A consumer WAVE loads the vectors it needs with agent-scope (L1-bypassing) loads
// and repeats the loads until none of their 8-byte words is the sentinel: the load that finds the data complete is the load that
// delivers it.  (A finite activation never encodes as four bf16 / two float32 NaNs with all mantissa bits set.)  Measured on
// 256 workgroups (tools/microbench/flagchain.hip): 1.2 - 1.35 us per all-to-all stage against 3.05 us for
// stores -> drain -> counter -> poll -> barrier -> load, the protocol of round 2.  Every spin is bounded: a wave that gives up
// raises the error word (1 + phase), which releases every other waiter; results are garbage then and the host raises.
struct Sync {
  unsigned* err;       // error word
  bool dead;           // this wave: a wait timed out somewhere: stop waiting, finish with whatever is there
  int p, wg, nwg;      // current phase / this workgroup
#ifdef JEN1_DEEP_PROFILE
  unsigned long long tt[16];
#endif
};

// Tuning builds only (-DJEN1_DEEP_PROFILE): thread 0 of every workgroup records the constant-rate 100 MHz counter at the
// stages of each unit it runs: dbg[(phase * nwg + wg) * 16 + stage]  (jen1_deep_debug_buffer sets the pointer)
#ifdef JEN1_DEEP_PROFILE
__device__ unsigned long long* g_deep_dbg = nullptr;
// stamps stay in registers (s_memrealtime is an asynchronous scalar-memory request) and are written once per unit
#define DK_STAMP(sy, i) do { (sy).tt[(i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define DK_FLUSH(sy) do { if (threadIdx.x == 0 && g_deep_dbg) { \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) g_deep_dbg[((size_t)(sy).p * (sy).nwg + (sy).wg) * 16 + i_] = (sy).tt[i_]; } \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) (sy).tt[i_] = 0; } while (0)
#else
#define DK_STAMP(sy, i) do { } while (0)
#define DK_FLUSH(sy) do { } while (0)
#endif
#if defined(JEN1_DEEP_PROFILE) && defined(JEN1_DEEP_PROFILE_NORM)
#define DK_STAMPN(sy, i) DK_STAMP(sy, i)     // finer stamps inside the normalised staging part; they reuse the set-up slots 7..10
#else
#define DK_STAMPN(sy, i) do { } while (0)
#endif

__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

constexpr u64 POISON = ~0ull;
#ifndef JEN1_DEEP_POLL_LIMIT
#define JEN1_DEEP_POLL_LIMIT (1u << 17)      // failed polls of one wait before the wave gives up (~1.3 us each: ~170 ms; 2^15 = ~40 ms was
                                             // reached now and then by four full-model samplers sharing a GPU that had just been powered up)
#endif
__device__ __forceinline__ bool raw_bad(const Raw8<bf16_t>& r) { return (r.d[0] == POISON) | (r.d[1] == POISON); }
__device__ __forceinline__ bool raw_bad(const Raw8<float>& r) {
  return (r.d[0] == POISON) | (r.d[1] == POISON) | (r.d[2] == POISON) | (r.d[3] == POISON);
}
// 4 consecutive channels (a residual operand) as raw words
template <typename T> struct Raw4;
template <> struct Raw4<bf16_t> { u64 d[1]; };
template <> struct Raw4<float> { u64 d[2]; };
__device__ __forceinline__ void ld_live4r(Raw4<bf16_t>& r, const bf16_t* p) { r.d[0] = __hip_atomic_load(g64(p), RLX_AGENT); }
__device__ __forceinline__ void ld_live4r(Raw4<float>& r, const float* p) {
  r.d[0] = __hip_atomic_load(g64(p), RLX_AGENT);
  r.d[1] = __hip_atomic_load(g64(p) + 1, RLX_AGENT);
}
__device__ __forceinline__ bool raw_bad(const Raw4<bf16_t>& r) { return r.d[0] == POISON; }
__device__ __forceinline__ bool raw_bad(const Raw4<float>& r) { return (r.d[0] == POISON) | (r.d[1] == POISON); }
__device__ __forceinline__ void raw4_to_float(const Raw4<bf16_t>& r, float (&o)[4]) {
  const unsigned lo = (unsigned)r.d[0], hi = (unsigned)(r.d[0] >> 32);
  o[0] = __uint_as_float(lo << 16); o[1] = __uint_as_float(lo & 0xffff0000u);
  o[2] = __uint_as_float(hi << 16); o[3] = __uint_as_float(hi & 0xffff0000u);
}
__device__ __forceinline__ void raw4_to_float(const Raw4<float>& r, float (&o)[4]) {
  o[0] = __uint_as_float((unsigned)r.d[0]); o[1] = __uint_as_float((unsigned)(r.d[0] >> 32));
  o[2] = __uint_as_float((unsigned)r.d[1]); o[3] = __uint_as_float((unsigned)(r.d[1] >> 32));
}

// behind a round of loads of one wave: `bad` = this lane saw a sentinel word.  Returns true when the wave has to load again.
// (wave-uniform; the error word is looked at every 64th failed poll)
__device__ __forceinline__ bool poll_again(Sync& sy, bool bad, unsigned& spins) {
#ifdef JEN1_DEEP_EXP_NOWAIT      // timing experiment only: results are garbage
  return false;
#endif
  if (!__builtin_amdgcn_ballot_w64(bad) || sy.dead) return false;
  ++spins;
  if ((spins & 63u) == 0u) {
    const unsigned ev = __hip_atomic_load(g32(sy.err), RLX_AGENT);
    if (rfl((int)ev) != 0) { sy.dead = true; return false; }
  }
  if (spins > JEN1_DEEP_POLL_LIMIT) {
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(g32(sy.err), (unsigned)(sy.p + 1), RLX_AGENT);
    sy.dead = true;
    return false;
  }
#if JEN1_DEEP_POLL_SLEEP > 0
  __builtin_amdgcn_s_sleep(JEN1_DEEP_POLL_SLEEP);     // (units of 64 clocks) between polls
#endif
  return true;
}

// ---- scheduling: static ------------------------------------------------------------------------------------------------------------
// Unit u of phase p runs on workgroup (u + rot_p) mod nwg: every workgroup knows its units from the headers alone, requests the
// next unit's descriptor during the current one and its weight ring right behind it, and a workgroup that just finished a unit is
// rarely the one the next phase waits for.  Fastest (889 us per launch at B = 8, T = 1500 against 971 with tickets), but correct only
// while all nwg workgroups are resident together: the host uses it for ONE plan per device at a time (engine.py) and the ticket
// form below for every other persistent launch that may share the GPU with it.
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

// ---- scheduling: tickets -----------------------------------------------------------------------------------------------------------
// The units of the launch are numbered phase by phase; a workgroup takes the next number from one device counter (sync[0], zeroed
// with the poisoning) whenever it needs work.  Whoever holds the smallest unfinished ticket depends only on smaller tickets, which are
// finished or held by workgroups that are already running: the launch makes progress with ANY number of resident workgroups, so two
// persistent launches sharing the GPU (another stream, another process) slow each other down but cannot deadlock.  A workgroup keeps
// two tickets ahead of the unit it runs -- the next unit's descriptor and weight ring are requested a whole unit early -- and the
// counter's round trip rides on the unit in between.
__device__ __forceinline__ void ticket_decode(const Hdr* hdr, int n_phases, int t, int& p, int& u) {      // p: hint, tickets only grow
  while (p + 1 < n_phases && t >= hdr[p + 1].before) ++p;
  u = t - hdr[p].before;
}

// =====================================================================================================================
// GEMM unit.  D: the phase's blob in LDS.
//
// A wave retires an instruction every ~2 ns here (two waves per SIMD, dependent chains), so a unit is written for few
// dynamic instructions per wave:
//   * the staged tile gives every batch element zero halo rows, so a conv tap is a plain row offset: the LDS address of an
//     activation fragment is (per-lane column base) + (per-chunk SCALAR shift * pitch + channel offset), no validity tests;
//   * GroupNorm statistics never touch LDS: each (batch element, group) pair owns 2^lS consecutive lanes, a lane sums its own
//     vectors and a fixed DPP / shuffle tree finishes the pair (bit-reproducible); one barrier per staged tile;
//   * a thread owns ONE 8-channel column of the normalised part and one of the raw part: source, scale and (gamma, beta) / FiLM
//     parameters are per-thread constants; every address is computed BEFORE the dependency wait; vectors are processed
//     branch-free (vectors that do not exist read row 0 and land in a dummy slot);
//   * the K loop runs whole rounds of the PF ring slots straight-line (slots beyond the wave's chunks hold zero weights).
// =====================================================================================================================
// ---- descriptor fields without LDS round trips: lane i of two registers holds dwords i and 64 + i of the phase blob; a field is a
// v_readlane (scalar result), a pointer two of them.  (One ds_read pair per unit instead of one dependent LDS read per field.)
struct HotRegs {
  unsigned a, b;
};
static_assert(sizeof(jen1_deep_hot) <= 512, "the hot block must fit two dwords per lane");
__device__ __forceinline__ HotRegs hot_regs(const unsigned char* D, int lane) {
  const unsigned* w = reinterpret_cast<const unsigned*>(D);
  HotRegs r;
  r.a = w[lane];
  r.b = w[64 + lane];
  return r;
}
template <int OFF>
__device__ __forceinline__ int hot_i32(const HotRegs& r) {
  static_assert(OFF % 4 == 0 && OFF >= 0 && OFF < 512, "field offset");
  return OFF < 256 ? __builtin_amdgcn_readlane((int)r.a, (OFF / 4) & 63) : __builtin_amdgcn_readlane((int)r.b, (OFF / 4 - 64) & 63);
}
template <int OFF>
__device__ __forceinline__ float hot_f32(const HotRegs& r) { return __builtin_bit_cast(float, hot_i32<OFF>(r)); }
template <int OFF, typename PT>
__device__ __forceinline__ PT hot_ptr(const HotRegs& r) {
  const u64 v = (u64)(unsigned)hot_i32<OFF>(r) | ((u64)(unsigned)hot_i32<OFF + 4>(r) << 32);
  return reinterpret_cast<PT>(v);
}
#define HI(f) hot_i32<offsetof(jen1_deep_hot, f)>(hr)
#define HF(f) hot_f32<offsetof(jen1_deep_hot, f)>(hr)
#define HP(f, type) hot_ptr<offsetof(jen1_deep_hot, f), type>(hr)
#define HSRC_OFF(k, f) (offsetof(jen1_deep_hot, src) + (k) * sizeof(jen1_deep_src) + offsetof(jen1_deep_src, f))
// source k of the table as scalars
template <int K>
__device__ __forceinline__ jen1_deep_src hot_src(const HotRegs& hr) {
  jen1_deep_src s;
  s.x = hot_ptr<HSRC_OFF(K, x), const void*>(hr);
  s.ld = hot_i32<HSRC_OFF(K, ld)>(hr);
  s.C = hot_i32<HSRC_OFF(K, C)>(hr);
  s.coff = hot_i32<HSRC_OFF(K, coff)>(hr);
  s.scale = hot_f32<HSRC_OFF(K, scale)>(hr);
  return s;
}

// the same for any field of the phase (attention / statistics units): four registers cover the 1 KiB descriptor
struct PhaseRegs {
  unsigned w[4];
};
static_assert(sizeof(jen1_deep_phase) <= 1024, "the descriptor must fit four dwords per lane");
__device__ __forceinline__ PhaseRegs phase_regs(const unsigned char* D, int lane) {
  const unsigned* w = reinterpret_cast<const unsigned*>(D);
  PhaseRegs r;
#pragma unroll
  for (int k = 0; k < 4; ++k) r.w[k] = w[64 * k + lane];
  return r;
}
template <int OFF>
__device__ __forceinline__ int ph_i32(const PhaseRegs& r) {
  static_assert(OFF % 4 == 0 && OFF >= 0 && OFF < 1024, "field offset");
  return __builtin_amdgcn_readlane((int)r.w[OFF / 256], (OFF / 4) & 63);
}
template <int OFF>
__device__ __forceinline__ float ph_f32(const PhaseRegs& r) { return __builtin_bit_cast(float, ph_i32<OFF>(r)); }
template <int OFF, typename PT>
__device__ __forceinline__ PT ph_ptr(const PhaseRegs& r) {
  const u64 v = (u64)(unsigned)ph_i32<OFF>(r) | ((u64)(unsigned)ph_i32<OFF + 4>(r) << 32);
  return reinterpret_cast<PT>(v);
}
#define AI(f) ph_i32<offsetof(jen1_deep_phase, f)>(pr)
#define AF(f) ph_f32<offsetof(jen1_deep_phase, f)>(pr)
#define AP(f, type) ph_ptr<offsetof(jen1_deep_phase, f), type>(pr)

struct KRun {                // one run of a wave's K chunks: chunk j is flat chunk g0 + j * NW at staged column col0 + j * NW * 32
  int g0, n, col0, shift;
};

template <typename T>
struct GemmWave {            // what prefill and the K loop share (all scalar)
  int mt, total, nruns, MT, grp, chunk;
  bool low_m;
  const KRun* runs;          // this wave's run list in the LDS blob
  __amdgpu_buffer_rsrc_t rw;
};

template <typename T>
__device__ __forceinline__ GemmWave<T> gemm_wave(const unsigned char* D, const HotRegs& hr, int u, int wk) {
  GemmWave<T> g;
  g.MT = HI(MT);
  const int mrep = HI(mrep);
  const int MTg = g.MT >> (mrep > 4 ? 3 : (mrep > 2 ? 2 : (mrep > 1 ? 1 : 0)));      // units per batch group (mrep: 1, 2, 4, 8)
  const int grpc = (int)(((float)u + 0.5f) * (1.0f / (float)MTg));     // u < 2^20: exact; (batch group, column chunk)
  g.mt = (u - grpc * MTg) * mrep;                                       // the unit's first M tile
#ifndef JEN1_DEEP_CHUNKS      // (column chunks are a build option: their decode costs every GEMM unit ~50 ns = 8 us per launch)
  const int grp = grpc;
  g.chunk = 0;
#else
  const int grp = (int)(((float)grpc + 0.5f) * HF(inv_nchunks));        // (n_chunks = 1: grp = grpc, chunk = 0)
  g.chunk = grpc - grp * HI(n_chunks);
#endif
  g.grp = grp;
  g.low_m = g.mt < HI(mt_split);
  const short* cnt = reinterpret_cast<const short*>(D + TAB_OFF);
  g.total = rfl(g.low_m ? cnt[NW + wk] : cnt[wk]);
  g.nruns = rfl(g.low_m ? cnt[3 * NW + wk] : cnt[2 * NW + wk]);
  g.runs = reinterpret_cast<const KRun*>(D + TAB_OFF + 64) + wk * MAXRUN;
  g.rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(HP(w, const void*)), 0, HI(w_bytes), RSRC_FLAGS);
  return g;
}

// scalar cursor over a wave's runs
struct KCursor {
  int r, left, g, col, shift;
};
__device__ __forceinline__ void kc_load(KCursor& c, const KRun* runs, int r) {
  c.r = r;
  c.left = rfl(runs[r].n);
  c.g = rfl(runs[r].g0);
  c.col = rfl(runs[r].col0);
  c.shift = rfl(runs[r].shift);
}
__device__ __forceinline__ void kc_start(KCursor& c, const KRun* runs, int nruns) {
  c.r = 0; c.left = 0; c.g = 0; c.col = 0; c.shift = 0;
  if (nruns > 0) kc_load(c, runs, 0);
}
__device__ __forceinline__ void kc_next(KCursor& c, const KRun* runs, int nruns) {
  if (--c.left > 0) {
    c.g += NW;
    c.col += NW * 32;
  } else if (c.r + 1 < nruns) {
    kc_load(c, runs, c.r + 1);
  }
}

template <typename T, typename Frag>
__device__ __forceinline__ void gemm_issue(const GemmWave<T>& g, const KCursor& c, int lane, Frag& fa) {
  constexpr int ES = sizeof(T);
  constexpr unsigned BLK = 512 * ES;
#ifdef JEN1_DEEP_EXP_NOW          // timing experiment only: no weight traffic
  wload(fa, g.rw, OOB, 0);
#else
  wload(fa, g.rw, (unsigned)lane * (8u * ES), (unsigned)(c.g * g.MT + g.mt) * BLK);
#endif
}

// fill the ring of a unit as early as its descriptor is known (right behind the previous unit's arrival); slots beyond the wave's
// chunks are zeroed (the K loop runs whole rounds)
// the wave's tabulated first round: lane i < SLOTS holds chunk index / staged offset of slot i
struct SlotTab {
  int g, off;
};
__device__ __forceinline__ SlotTab slot_tab(const unsigned char* D, int wk, int lane) {
  const int* t = reinterpret_cast<const int*>(D + SLOT_OFF) + wk * (2 * SLOTS);
  SlotTab r;
  const int l = lane < SLOTS ? lane : 0;
  r.g = t[l];
  r.off = t[SLOTS + l];
  return r;
}
template <typename T, typename Frag>
__device__ __forceinline__ void gemm_issue_g(const GemmWave<T>& g, int chunk, int lane, Frag& fa) {
  constexpr int ES = sizeof(T);
  constexpr unsigned BLK = 512 * ES;
#ifdef JEN1_DEEP_EXP_NOW
  wload(fa, g.rw, OOB, 0);
#else
  wload(fa, g.rw, (unsigned)lane * (8u * ES), (unsigned)(chunk * g.MT + g.mt) * BLK);
#endif
}

// fill the ring of a unit as early as its descriptor is known (right behind the previous unit's arrival); slots beyond the wave's
// chunks are zeroed (the K loop runs whole rounds)
template <typename T, typename Frag, int PF>
__device__ __forceinline__ void gemm_prefill(const unsigned char* D, int u, int wk, int lane, Frag (&ra)[PF]) {
  const HotRegs hr = hot_regs(D, lane);
  const GemmWave<T> g = gemm_wave<T>(D, hr, u, wk);
  const SlotTab st = slot_tab(D, wk, lane);
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    if (i < g.total) gemm_issue_g<T>(g, __builtin_amdgcn_readlane(st.g, i), lane, ra[i]);
    else frag_zero_d(ra[i]);
  }
}

// K loop of one round: the first NS ring slots x NF fragments, straight-line
template <typename T, int NF, int NS, typename Frag, int PF>
__device__ __forceinline__ void k_round(f32x4 (&acc)[4], const Frag (&ra)[PF], const T* tile, const int (&cbase)[4], const int (&soff)[PF]) {
#pragma unroll
  for (int i = 0; i < (NS < PF ? NS : PF); ++i) {
    Frag fb[NF];
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) dlds(fb[nf], tile + cbase[nf] + soff[i]);
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) dmma(acc[nf], ra[i], fb[nf]);
  }
}
// a round of `left` chunks (slots beyond them hold zero weights: rounded up to 4 slots)
template <typename T, int NF, typename Frag, int PF>
__device__ __forceinline__ void k_round_n(f32x4 (&acc)[4], const Frag (&ra)[PF], const T* tile, const int (&cbase)[4], const int (&soff)[PF], int left) {
  if (left <= 4) k_round<T, NF, 4>(acc, ra, tile, cbase, soff);
  else if (left <= 8) k_round<T, NF, 8>(acc, ra, tile, cbase, soff);
  else k_round<T, NF, PF>(acc, ra, tile, cbase, soff);
}

// sum over the aligned group of 2^lS lanes (1 <= lS <= 6) that holds v, in a fixed order
// x[lane] + x[lane ^ 16] / x[lane ^ 32] without the LDS crossbar: v_permlane16_swap exchanges the odd rows of its first operand with
// the even rows of its second, v_permlane32_swap the upper half with the lower half; with both operands the same value the two
// results hold the value of the even / odd row (lower / upper half) of every row pair -- their sum is the butterfly step
__device__ __forceinline__ float xor16_add(float v) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_add(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// sum and sum of squares over the 2^lS lanes of a lane set (lS wave-uniform, 1..6), the two chains interleaved
__device__ __forceinline__ void lane_set_sum2(float& s, float& q, int lS) {
  s += ddpp<0xB1>(s); q += ddpp<0xB1>(q);                         // lanes ^ 1
  if (lS >= 2) { s += ddpp<0x4E>(s); q += ddpp<0x4E>(q); }        // lanes ^ 2
  if (lS >= 3) { s += ddpp<0x141>(s); q += ddpp<0x141>(q); }      // row_half_mirror: the other quad of the 8
  if (lS >= 4) { s += ddpp<0x140>(s); q += ddpp<0x140>(q); }      // row_mirror: the other half of the 16
  if (lS >= 5) { s = xor16_add(s); q = xor16_add(q); }
  if (lS >= 6) { s = xor32_add(s); q = xor32_add(q); }
}

template <typename T, typename Frag, int PF, typename FPub>
__device__ __forceinline__ void gemm_unit(const unsigned char* D, int u, Sync& sy, Frag (&ra)[PF], FPub publish_next, int tid) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr bool PRECISE = is_f32<T>::value;
  constexpr int MAXV = DeepCfg<T>::MAXV;
  typedef typename Mode<T>::G GT;                     // activations in global memory (T itself, bf16 in JEN1_FP8 mode)
  const jen1_deep_phase* P = reinterpret_cast<const jen1_deep_phase*>(D);
  const int lane = tid & 63;
  const int wk = rfl(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  DK_STAMP(sy, 0);
  const HotRegs hr = hot_regs(D, lane);
  GemmWave<T> gw = gemm_wave<T>(D, hr, u, wk);
  const int nb = HI(nb), B = HI(B), L_in = HI(L_in), L_out = HI(L_out), NF = HI(NF);
  const int pitch = HI(pitch), norm_C = HI(norm_C), Ctot = HI(Ctot), Lp = HI(Lp), Hb = HI(Hb);
  const int zrow = HI(zrow), Rtot = HI(Rtot);
  const int mrep = HI(mrep);                       // the unit finishes mrep M tiles from one staged tile
  const int b0 = gw.grp * nb;
#ifndef JEN1_DEEP_CHUNKS
  const int Lc = L_out, q0 = 0;
#else
  const int Lc = HI(Lc), q0 = gw.chunk * Lc;          // the unit's column chunk (Lc = L_out, q0 = 0 without chunking)
#endif
  const bool low_m = gw.low_m;
  unsigned char* ws = smem + WS_OFF;
  T* tile = reinterpret_cast<T*>(ws);
  float* red = reinterpret_cast<float*>(ws + HI(red_off));
  // per-thread landing place of vectors that do not exist (branch-free stores): the K-reduction scratch, unused while staging
  const int dummy_tile = HI(red_off) / (int)sizeof(T) + tid * 8;
  const float inv_Lin = HF(inv_Lin);
  const float inv_Lout = HF(inv_Lout);

  DK_STAMP(sy, 7);
  // ---- the normalised part: (batch element, group) pair of this lane set, column of this lane -----------------------------
  const int lS = HI(lS), lvpg = HI(lvpg), lgroups = HI(lgroups), cpg = HI(gn_cpg);
  const int pair = tid >> lS, wl = tid & ((1 << lS) - 1);
  const int nbl = pair >> lgroups, ngrp = pair & ((1 << lgroups) - 1);
  const int cn = ngrp * cpg + (wl & ((1 << lvpg) - 1)) * 8;       // first channel of the lane's column
  const int nt0 = wl >> lvpg, ntstep = (1 << lS) >> lvpg;           // first position, positions per trip
  const bool npair_ok = norm_C > 0 && nbl < nb && b0 + nbl < B;
  const jen1_deep_src s0 = hot_src<0>(hr), s1 = hot_src<1>(hr);
  const bool in1 = HI(nsrc) > 1 && s1.coff < norm_C && cn >= s1.coff;         // second normalised source (the skip)
  const GT* nbase = reinterpret_cast<const GT*>(in1 ? s1.x : s0.x) + (cn - (in1 ? s1.coff : 0)) +
                   (size_t)((unsigned)((npair_ok ? b0 + nbl : b0) * L_in) * (unsigned)(in1 ? s1.ld : s0.ld));
  const int nld = in1 ? s1.ld : s0.ld;
  const float nscale = in1 ? s1.scale : s0.scale;
  float p1[8], p2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { p1[j] = 0.f; p2[j] = 0.f; }
  if (norm_C) {
    // one table row per unit: gamma / beta (p_ld = 0), the sampler's per-step row, or the row of the unit's batch element
    // (units of a per-element table hold one batch element: jen1_deep_phase_conv)
    const int p_ld = HI(p_ld);
    const int* fstep = HP(film_step, const int*);
    const int* frow = HP(film_row, const int*);
    const int fr = p_ld ? (fstep ? fstep[0] : (frow ? frow[b0] : b0)) : 0;
    const size_t po = (size_t)((unsigned)fr * (unsigned)p_ld) + (unsigned)(npair_ok ? cn : 0);
    load8(HP(p1, const float*) + po, p1);
    load8(HP(p2, const float*) + po, p2);
  }
  const GT* nap[MAXV];
  int ntile[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int t = nt0 + i * ntstep;
    const bool ok = npair_ok && t < L_in;
    nap[i] = nbase + (size_t)((unsigned)(ok ? t : 0) * (unsigned)nld);
    ntile[i] = ok ? (nbl * Lp + Hb + t) * pitch + cn : dummy_tile;
  }
  DK_STAMP(sy, 8);
  // ---- the raw part: a thread owns one column, rows r0, r0 + rpr, ... of the unit's nb * L_in staged rows ----------------------
  const int Craw = Ctot - norm_C;
  const int lvr = HI(lvr);
  const int cr = Craw > 0 ? norm_C + (tid & ((1 << lvr) - 1)) * 8 : 0;      // (no raw part: the dummy loads stay inside source 0)
  const int rr0 = tid >> lvr, rpr = NT >> lvr;
  int rows_ok = (B - b0) * L_in;                       // staged rows below this belong to real batch elements
  rows_ok = rows_ok < nb * L_in ? rows_ok : nb * L_in;
  rows_ok = Craw > 0 ? rows_ok : 0;
  const GT* rbase;
  int rld, rsrc;
  float rscale;
  {
    const void* xp = s0.x;
    int ld = s0.ld, coff = 0;
    float sc = s0.scale;
    rsrc = 0;
    static_assert(JEN1_DEEP_MAX_SRC == 4, "the source scan below is written out for four sources");
    const int nsrc = HI(nsrc);
    auto pick = [&](const jen1_deep_src sk, int k) __attribute__((always_inline)) {
      const bool use = k < nsrc && cr >= sk.coff;
      rsrc = use ? k : rsrc;
      xp = use ? sk.x : xp;
      ld = use ? sk.ld : ld;
      coff = use ? sk.coff : coff;
      sc = use ? sk.scale : sc;
    };
    pick(s1, 1);
    pick(hot_src<2>(hr), 2);
    pick(hot_src<3>(hr), 3);
    rbase = reinterpret_cast<const GT*>(xp) + (cr - coff) + (size_t)((unsigned)(b0 * L_in) * (unsigned)ld);
    rld = ld;
    rscale = sc;
  }
  const int lpx = Lp - L_in;                            // halo rows per batch element
  const GT* wap[MAXV];
  int wtile[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int rr = rr0 + i * rpr;
    const bool ok = rr < rows_ok;
    const int rc = ok ? rr : 0;
    const int bl = (int)(((float)rc + 0.5f) * inv_Lin);
    wap[i] = rbase + (size_t)((unsigned)rc * (unsigned)rld);
    wtile[i] = ok ? (rc + bl * lpx + Hb) * pitch + cr : dummy_tile;
  }
  // vectors of this unit that exist for at least one lane (scalar): the branch-free slots beyond them are skipped altogether --
  // at L_in <= 6 most units own one or two of the MAXV slots, and the normalisation (two transcendentals per element, two waves
  // per SIMD) is what a unit spends its time on behind the wait: deep launch 891 -> 857 us together with V^T's own LDS region
  int nvn = norm_C > 0 ? (L_in + ntstep - 1) >> (lS - lvpg) : 0;         // (ntstep = 2^(lS - lvpg), rpr = NT >> lvr)
  nvn = nvn < MAXV ? nvn : MAXV;
  int nvr = (rows_ok + rpr - 1) >> (__builtin_ctz(NT) - lvr);
  nvr = nvr < MAXV ? nvr : MAXV;
  DK_STAMP(sy, 9);
  // ---- halo rows and the zero block: nobody else touches them, written before the wait ----------------------------------------
  {
    const int vpr = Ctot >> 3;
    const int hz = nb * lpx;                           // halo rows of the batch elements, then the zero block
    const int zr = hz + (Rtot - nb * Lp);
    const float inv_vprf = 1.0f / (float)vpr;
    const float inv_lpx = lpx ? 1.0f / (float)lpx : 0.f;
    for (int i = tid; i < zr * vpr; i += NT) {
      const int z = (int)(((float)i + 0.5f) * inv_vprf);
      const int c = (i - z * vpr) * 8;
      int row;
      if (z < hz) {
        const int bl = (int)(((float)z + 0.5f) * inv_lpx);
        const int k = z - bl * lpx;
        row = bl * Lp + (k < Hb ? k : L_in + k);
      } else {
        row = nb * Lp + (z - hz);
      }
      const float z8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      store8(tile + (size_t)row * pitch + c, z8);
    }
  }
  DK_STAMP(sy, 10);
  // ---- epilogue operands that do not depend on other workgroups (per M tile of the unit) ------------------------------------
  const bool epi = wk < NF;
  const int nfe = wk;                                   // the fragment this wave finishes
  int co = 0, yrow = 0;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 wsc4 = {1.f, 1.f, 1.f, 1.f};                  // JEN1_FP8: the scale of the tile's four output rows of this lane
  bool okk = false, use_res = false;
  const GT* resp = nullptr;
  auto epi_operands = [&](int mt) {
    const int m = mt * 16 + lg * 4;
    int ph = 0;
    co = m;
    if (epi) {
      const int out_C = HI(out_C), ps_f = HI(ps_f);
      for (int k = 1; k < ps_f; ++k) ph += (m >= k * out_C) ? 1 : 0;
      co = m - ph * out_C;
      const float* biasp = HP(bias, const float*);
      if (biasp) bias4 = *reinterpret_cast<const f32x4*>(biasp + co);
      if (sizeof(T) == 1) wsc4 = *reinterpret_cast<const f32x4*>(HP(wscale, const float*) + m);
      const int n = nfe * 16 + li;
      const int ebl = (int)(((float)n + 0.5f) * inv_Lout);
      const int t = n - ebl * Lc + q0;
      const int ty = t * ps_f + ph - HI(ps_off);
      okk = n < nb * Lc && t < L_out && b0 + ebl < B && ty >= 0 && ty < HI(L_y);
      yrow = okk ? (b0 + ebl) * HI(y_brows) + HI(y_row0) + ty : 0;
    }
    const GT* resb = HP(residual, const GT*);
    use_res = epi && okk && resb && (HI(mt_split) == 0 || low_m);
    resp = resb + ((size_t)((unsigned)yrow * (unsigned)HI(ld_res)) + (unsigned)co);
  };
  epi_operands(gw.mt);
  DK_STAMP(sy, 12);
  // ---- K loop: column base of this lane per fragment; scalar (shift * pitch + channel) offset per ring slot of the first round ---
  const int total = gw.total;
  int cbase[4];
  {
    const int stride = HI(stride);
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
      const int n = nf * 16 + li;
      const int bl = (int)(((float)n + 0.5f) * inv_Lout);
      const int t = n - bl * Lc + q0;
      const bool ok = nf < NF && n < nb * Lc && t < L_out && b0 + bl < B;
      cbase[nf] = (ok ? bl * Lp + Hb + t * stride : zrow) * pitch + lg * 8;
    }
  }
  int soff[PF];
  const SlotTab stab = slot_tab(D, wk, lane);
#pragma unroll
  for (int i = 0; i < PF; ++i) soff[i] = __builtin_amdgcn_readlane(stab.off, i);      // (slots beyond the last chunk hold zero weights)
  KCursor cc;                                          // behind the first round's chunks (used by layers with more than PF per wave)
  {
    const int* ct = reinterpret_cast<const int*>(D + CUR_OFF) + wk * 8;
    cc.r = rfl(ct[0]); cc.left = rfl(ct[1]); cc.g = rfl(ct[2]); cc.col = rfl(ct[3]); cc.shift = rfl(ct[4]);
  }

  // ---- dependency ---------------------------------------------------------------------------------------------------
  DK_STAMP(sy, 1);
  // which of this lane's loads read tensors produced inside the launch (bit k: source k, bit 8: the residual): those are repeated
  // until no word is the sentinel; tensors of earlier launches (the first phase's input) are plain data
  const int live = HI(live_mask);
  const bool nlive = norm_C > 0 && ((live >> (in1 ? 1 : 0)) & 1);
  const bool rlive = Craw > 0 && ((live >> rsrc) & 1);
  const bool reslive = use_res && ((live >> 8) & 1);

#ifndef JEN1_DEEP_EXP_NOSTAGE
  // ---- every load (sc1: another workgroup wrote the data in this launch), no branches; the wave repeats them until complete ---
  // (measured, not kept: two rounds of polled loads in flight half a round trip apart, to sample the producers' stores twice as
  // often -- deep launch 889 -> 997 / 1005 / 1024 / 1067 us at gaps of 4 / 8 / 16 / 24 x 64 clocks; a light pre-poll of ONE vector
  // per lane ahead of the full round -- the polled volume is what an exchange costs, tools/microbench/flagchain.hip: 1.22 / 1.35 /
  // 1.85 / 2.5 us per stage at 1 / 2 / 4 / 8 vectors per thread on 256 workgroups -- 857 -> 896 us: the serial extra round trip
  // costs more than the lighter polls save; re-requesting only the vectors that came back incomplete (wave-uniform flags per
  // vector: the raw part and the residual of a block's second conv are phases old) 860 -> 867 us: the ballots and branches per
  // vector cost more than the re-read of complete vectors)
  Raw8<GT> xn[MAXV], xw[MAXV];
  float rres[4] = {0.f, 0.f, 0.f, 0.f};
  {
    Raw4<GT> rr;
    unsigned spins = 0;
    bool bad;
    do {
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        if (i < nvn) ld_live(xn[i], nap[i]);
      }
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        if (i < nvr) ld_live(xw[i], wap[i]);
      }
      if (use_res) ld_live4r(rr, resp);
      bad = false;
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        if (i < nvn) bad |= nlive && raw_bad(xn[i]);
      }
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        if (i < nvr) bad |= rlive && raw_bad(xw[i]);
      }
      if (use_res) bad |= reslive && raw_bad(rr);
    } while (poll_again(sy, bad, spins));
    if (use_res) raw4_to_float(rr, rres);
  }
  DK_STAMP(sy, 2);
  // raw vectors: straight into the tile
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    if (i < nvr) {
      if (rscale != 1.0f) {
        float x[8];
        raw_to_float(xw[i], x);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] *= rscale;
        float_to_raw(x, xw[i]);
      }
      stage_raw(tile + wtile[i], xw[i]);
    }
  }
  // further trips of a long raw input (e.g. the 94 rows a downsampling conv reads)
  for (int rbeg = MAXV * rpr; rbeg < rows_ok; rbeg += MAXV * rpr) {
    Raw8<GT> xv[MAXV];
    int vt[MAXV];
    unsigned spins = 0;
    bool bad;
    do {
      bad = false;
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int rr = rbeg + rr0 + i * rpr;
        const bool ok = rr < rows_ok;
        const int rc = ok ? rr : 0;
        const int bl = (int)(((float)rc + 0.5f) * inv_Lin);
        vt[i] = ok ? (rc + bl * lpx + Hb) * pitch + cr : dummy_tile;
        ld_live(xv[i], rbase + (size_t)((unsigned)rc * (unsigned)rld));
      }
#pragma unroll
      for (int i = 0; i < MAXV; ++i) bad |= rlive && raw_bad(xv[i]);
    } while (poll_again(sy, bad, spins));
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      if (rscale != 1.0f) {
        float x[8];
        raw_to_float(xv[i], x);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] *= rscale;
        float_to_raw(x, xv[i]);
      }
      stage_raw(tile + vt[i], xv[i]);
    }
  }
  DK_STAMP(sy, 11);
  if (norm_C) {
    // GroupNorm (+FiLM) (+SiLU): the lane sums its own vectors, the lane set finishes the pair, no LDS, no barrier
    float xf[MAXV][8];
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      if (i >= nvn) continue;
      raw_to_float(xn[i], xf[i]);
      const float nm = ntile[i] != dummy_tile ? nscale : 0.f;           // scale of the source, 0 for vectors that do not exist
#pragma unroll
      for (int j = 0; j < 8; ++j) xf[i][j] *= nm;
#ifndef JEN1_DEEP_EXP_NOSUMS      // timing experiment only (results are garbage): what ANY scheme that ships the statistics with the data could save
      const float a0 = (xf[i][0] + xf[i][1]) + (xf[i][2] + xf[i][3]), a1 = (xf[i][4] + xf[i][5]) + (xf[i][6] + xf[i][7]);
      const float c0 = (xf[i][0] * xf[i][0] + xf[i][1] * xf[i][1]) + (xf[i][2] * xf[i][2] + xf[i][3] * xf[i][3]);
      const float c1 = (xf[i][4] * xf[i][4] + xf[i][5] * xf[i][5]) + (xf[i][6] * xf[i][6] + xf[i][7] * xf[i][7]);
      s += a0 + a1;
      q += c0 + c1;
#endif
    }
    // long rows (more than MAXV vectors per lane: e.g. 36 positions x 512 channels of one batch element): the further trips are
    // summed here and read again below for the normalisation (their second read is an L2 hit)
    const int ntrips = HI(ntrips);
    for (int trip = 1; trip < ntrips; ++trip) {
      Raw8<GT> xt[MAXV];
      float mt_[MAXV];
      unsigned spins = 0;
      bool bad;
      do {
        bad = false;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
          const int t = nt0 + (trip * MAXV + i) * ntstep;
          const bool ok = npair_ok && t < L_in;
          mt_[i] = ok ? nscale : 0.f;
          ld_live(xt[i], nbase + (size_t)((unsigned)(ok ? t : 0) * (unsigned)nld));
        }
#pragma unroll
        for (int i = 0; i < MAXV; ++i) bad |= nlive && raw_bad(xt[i]);
      } while (poll_again(sy, bad, spins));      // (the second read of these rows below finds them complete)
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        float x[8];
        raw_to_float(xt[i], x);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] *= mt_[i];
        s += ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
        q += ((x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3])) + ((x[4] * x[4] + x[5] * x[5]) + (x[6] * x[6] + x[7] * x[7]));
      }
    }
    DK_STAMPN(sy, 7);
#ifndef JEN1_DEEP_EXP_NOSUMS
    lane_set_sum2(s, q, lS < 6 ? lS : 6);
#else
    s = 0.5f; q = 1.0f;
#endif
    if (lS == 7) {
      // a group of 1024 channels (LayerNorm over the channels of ONE position, folded single-position self-attention: engine.py
      // "s1q2") is 128 columns: the pair's two waves exchange their sums through LDS and add them in a fixed order
      float2* xw = reinterpret_cast<float2*>(smem + XW_OFF);
      if (lane == 0) xw[wk] = make_float2(s, q);
      __syncthreads();
      const float2 e = xw[wk & ~1], o = xw[wk | 1];
      s = e.x + o.x;
      q = e.y + o.y;
    }
    const float inv_count = HF(inv_count);
    const float gn_eps = HF(gn_eps);
    const float mean = s * inv_count;
    float var = q * inv_count - mean * mean;
    var = var < 0.f ? 0.f : var;
    const float rstd = PRECISE ? 1.0f / sqrtf(var + gn_eps) : rsqrtf(var + gn_eps);
    const bool silu = HI(pro_mode) == JEN1_PRO_GN_SILU;
    DK_STAMPN(sy, 8);
    // bf16 / fp8 staging: the affine map as ONE fused multiply-add per element (a = rstd * gamma', b = beta' - mean * a)
    float pa[8], pb[8];
    if (!PRECISE) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { pa[j] = rstd * p1[j]; pb[j] = p2[j] - mean * pa[j]; }
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      if (i >= nvn) continue;
#ifndef JEN1_DEEP_EXP_NONORM      // timing experiment only (results are garbage): what a consumer would save if producers stored the normalised activation
#pragma unroll
      for (int j = 0; j < 8; ++j) xf[i][j] = PRECISE ? (xf[i][j] - mean) * rstd * p1[j] + p2[j] : xf[i][j] * pa[j] + pb[j];
      if (silu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) xf[i][j] = PRECISE ? silu_precise(xf[i][j]) : silu_f(xf[i][j]);
      }
#else
      xf[i][0] += mean + rstd + pa[0] + pb[0];
#endif
      store8(tile + ntile[i], xf[i]);
    }
    DK_STAMPN(sy, 9);
    for (int trip = 1; trip < ntrips; ++trip) {
      Raw8<GT> xt[MAXV];
      int tt[MAXV];
      float mt_[MAXV];
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        const int t = nt0 + (trip * MAXV + i) * ntstep;
        const bool ok = npair_ok && t < L_in;
        mt_[i] = ok ? nscale : 0.f;
        tt[i] = ok ? (nbl * Lp + Hb + t) * pitch + cn : dummy_tile;
        ld_live(xt[i], nbase + (size_t)((unsigned)(ok ? t : 0) * (unsigned)nld));
      }
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        float x[8];
        raw_to_float(xt[i], x);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = PRECISE ? (x[j] * mt_[i] - mean) * rstd * p1[j] + p2[j] : (x[j] * mt_[i]) * pa[j] + pb[j];
        if (silu) {
#pragma unroll
          for (int j = 0; j < 8; ++j) x[j] = PRECISE ? silu_precise(x[j]) : silu_f(x[j]);
        }
        store8(tile + tt[i], x);
      }
    }
  }
#else
  float rres[4] = {0.f, 0.f, 0.f, 0.f};
#endif
  DK_STAMP(sy, 13);
  __syncthreads();
  DK_STAMP(sy, 3);

  // ---- per M tile of the unit: K loop, K reduction across the waves, epilogue.  The first tile is the straight path; further
  // tiles (mrep > 1: phases with more M tiles x batch groups than workgroups) reuse the staged tile -----------------------------
  const int red_floats = HI(red_bytes) >> 2;
  // K loop: rounds of the PF ring slots; a slot is refilled behind its use when the wave has more chunks
  auto k_loop = [&](f32x4 (&acc)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) acc[nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifndef JEN1_DEEP_EXP_NOK
    KCursor ic = cc;                                   // already behind the first round's chunks when there are more
    for (int c0 = 0; c0 < total; c0 += PF) {
      const int left = total - c0;
      if (NF == 1) k_round_n<T, 1>(acc, ra, tile, cbase, soff, left);
      else if (NF == 2) k_round_n<T, 2>(acc, ra, tile, cbase, soff, left);
      else if (NF == 3) k_round_n<T, 3>(acc, ra, tile, cbase, soff, left);
      else k_round_n<T, 4>(acc, ra, tile, cbase, soff, left);
      if (c0 + PF < total) {
        // next round: request its chunks (slots beyond the end: zero), note their staged offsets
#pragma unroll
        for (int i = 0; i < PF; ++i) {
          if (c0 + PF + i < total) {
            gemm_issue<T>(gw, ic, lane, ra[i]);
            soff[i] = ic.shift * pitch + ic.col;
            if (c0 + PF + i + 1 < total) kc_next(ic, gw.runs, gw.nruns);
          } else {
            frag_zero_d(ra[i]);
          }
        }
      }
    }
#endif
  };
  // the next tile of the unit: the same chunks, the next 1 KiB fragment of each; requested behind a K loop
  auto ring_next_tile = [&]() __attribute__((always_inline)) {
    gw.mt += 1;
#pragma unroll
    for (int i = 0; i < PF; ++i) {
      if (i < total) {
        gemm_issue_g<T>(gw, __builtin_amdgcn_readlane(stab.g, i), lane, ra[i]);
        soff[i] = __builtin_amdgcn_readlane(stab.off, i);
      } else {
        frag_zero_d(ra[i]);
      }
    }
  };
  auto put_partial = [&](const f32x4 (&acc)[4], float* redj) __attribute__((always_inline)) {
#pragma unroll
    for (int nf = 0; nf < 4; ++nf) {
      if (nf < NF) *reinterpret_cast<float4*>(redj + ((size_t)(wk * NF + nf) * 64 + lane) * 4) = make_float4(acc[nf][0], acc[nf][1], acc[nf][2], acc[nf][3]);
    }
  };
  // K reduction across the waves in a fixed order, epilogue by the first NF waves
  auto epilogue = [&](const float* redj) __attribute__((always_inline)) {
    float4 o[NW];
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) o[w2] = *reinterpret_cast<const float4*>(redj + ((size_t)(w2 * NF + nfe) * 64 + lane) * 4);
    float v[4];
    // as a tree: ((0+1)+(2+3)) + ((4+5)+(6+7))
#pragma unroll
    for (int st = 1; st < NW; st <<= 1) {
#pragma unroll
      for (int w2 = 0; w2 + st < NW; w2 += 2 * st) {
        o[w2].x += o[w2 + st].x; o[w2].y += o[w2 + st].y; o[w2].z += o[w2 + st].z; o[w2].w += o[w2 + st].w;
      }
    }
    if (sizeof(T) == 1) { o[0].x *= wsc4[0]; o[0].y *= wsc4[1]; o[0].z *= wsc4[2]; o[0].w *= wsc4[3]; }
    v[0] = o[0].x + bias4[0]; v[1] = o[0].y + bias4[1]; v[2] = o[0].z + bias4[2]; v[3] = o[0].w + bias4[3];
    if (HI(act) == JEN1_ACT_GELU && !low_m) {
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
    }
#ifdef JEN1_DEEP_EXP_NOEPI
    if (okk && v[0] == 1234.5f) {
#else
    if (okk) {
#endif
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += rres[r];
      const size_t off = (size_t)((unsigned)yrow * (unsigned)HI(ld_y)) + (unsigned)co;
      if (HI(y_f32)) st_live4(HP(y, float*) + off, v);
      else st_live4(HP(y, GT*) + off, v);
    }
  };
  {
    f32x4 acc[4];
    k_loop(acc);
    if (mrep > 1) ring_next_tile();
    DK_STAMP(sy, 4);
    put_partial(acc, red);
    publish_next();                                      // the next unit's descriptor rides on this barrier
    __syncthreads();
    DK_STAMP(sy, 14);
    if (epi) epilogue(red);
  }
  for (int j = 1; j < mrep; ++j) {
    // consecutive tiles alternate between two scratch areas: one barrier per tile
    epi_operands(gw.mt);
#pragma unroll
    for (int r = 0; r < 4; ++r) rres[r] = 0.f;
    if (use_res) {
      Raw4<GT> rr;
      unsigned spins = 0;
      do {
        ld_live4r(rr, resp);
      } while (poll_again(sy, ((live >> 8) & 1) && raw_bad(rr), spins));
      raw4_to_float(rr, rres);
    }
    f32x4 acc[4];
    k_loop(acc);
    if (j + 1 < mrep) ring_next_tile();
    float* redj = red + (j & 1) * red_floats;
    put_partial(acc, redj);
    __syncthreads();
    if (epi) epilogue(redj);
  }
  DK_STAMP(sy, 15);
  __syncthreads();          // (LDS: the next unit stages over this one's tile and reduction scratch; the stores need no drain)
  DK_STAMP(sy, 5);
  DK_STAMP(sy, 6);
}

// =====================================================================================================================
// attention unit: one (batch element, head, 32-query chunk); attention.hip's structure for 8 waves with the LayerNorm
// statistics of the deferred finish computed here (blocks.py:355-380, :427-429)
// =====================================================================================================================
template <typename T, typename FPub>
__device__ __forceinline__ void attn_unit(const unsigned char* D, int u, Sync& sy, FPub publish_next, int tid) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef typename DFrag<T>::type Frag;
  typedef typename Mode<T>::G GT;                      // q / k / v / out in global memory (bf16 in JEN1_FP8 mode)
  constexpr bool PRECISE = is_f32<T>::value;
  constexpr bool F8 = sizeof(T) == 1;
  constexpr float PS = F8 ? P_SCALE : 1.0f;            // the probabilities are stored as PS * p (see P_SCALE)
  constexpr int MAXVA = (141 * 16 + NT - 1) / NT;      // K / V vectors per thread at the longest supported context
  const jen1_deep_phase* P = reinterpret_cast<const jen1_deep_phase*>(D);
  const int lane = tid & 63;
  const int wave = rfl(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  DK_STAMP(sy, 0);
  const PhaseRegs pr = phase_regs(D, lane);
  const int H = AI(H), d = AI(d), Nq = AI(Nq), Nk = AI(Nk), nqc = AI(nqc);
  const int bh = (int)(((float)u + 0.5f) * AF(inv_nqc));
  const int qc = u - bh * nqc;
  const int b = (int)(((float)bh + 0.5f) * AF(inv_H)), h = bh - b * H;
  const int q0 = qc * QCHUNK;
  const int nq = (Nq - q0 < QCHUNK) ? (Nq - q0) : QCHUNK;
  const int NKP = (Nk + 31) & ~31;
  const int DP = d < 32 ? 32 : d;
  const int DC = d < 16 ? 16 : d;
  const int dq = DP + 8, vt = NKP + 8, sp = NKP + 4;
  T* q_s = reinterpret_cast<T*>(smem + WS_OFF);                      // [32][dq]  (later the output tile, in the activations' type)
  GT* o_s = reinterpret_cast<GT*>(smem + WS_OFF);
  T* kv_s = reinterpret_cast<T*>(o_s + QCHUNK * dq);                // K [NKP][dq]
  // V^T [DC][vt]: a region of its own when the phase's LDS allows (vsep: staged together with K -- for cross-attention BEFORE the
  // dependency wait, its K / V are cached text projections), else in K's place once the scores are done
  const bool vsep = AI(vsep) != 0;
  T* vt_s = vsep ? kv_s + ((NKP * dq + 15) & ~15) : kv_s;
  const int kv_elems = vsep ? ((NKP * dq + 15) & ~15) + DC * vt : ((NKP * dq > DC * vt) ? NKP * dq : DC * vt);
  float* s_s = reinterpret_cast<float*>(kv_s + ((kv_elems + 15) & ~15));   // [32][sp]
  T* p_s = reinterpret_cast<T*>(s_s + ((QCHUNK * sp + 3) & ~3));      // [32][vt]
  float2* st_s = reinterpret_cast<float2*>(p_s + ((QCHUNK * vt + 7) & ~7));   // [max(Nk, 32)] LayerNorm (mean, rstd) per row

  const int ldq = AI(ldq), ldkv = AI(ldkv), log2_vpr = AI(log2_vpr), ldo = AI(ldo), ln_C = AI(ln_C);
  const int vpr = 1 << log2_vpr;
  const int nkv = Nk * vpr, nqv = nq * vpr;
  const int hd = h * d;
  const GT* qp = reinterpret_cast<const GT*>(AP(q, const void*));
  const GT* kp_ = reinterpret_cast<const GT*>(AP(k, const void*));
  const GT* vp_ = reinterpret_cast<const GT*>(AP(v, const void*));
  const GT* xp_ = reinterpret_cast<const GT*>(AP(kv_extra, const void*));
  GT* const outp = reinterpret_cast<GT*>(AP(out, void*));
  const int fin_q = AI(fin_q), fin_kv = AI(fin_kv), causal = AI(causal);
  const int kv_live = AI(kv_live) & 1;                 // K / V were produced inside this launch (self-attention)
  const bool q_live = (AI(kv_live) >> 1) & 1;          // ... and so was q's tensor (always, unless the unit is the first phase)
  const float scale = AF(scale), ln_eps = AF(ln_eps);
  const int q_off = AI(q_off);

  // ---- operands that do not depend on other workgroups: cached text K/V, the finish vectors u / b ----------------------
  const int kvbase = (AP(kv_row, const int32_t*) ? AP(kv_row, const int32_t*)[b] : b) * Nk;
  int xr = (AP(kv_extra, const void*) && AP(extra_row, const int32_t*)) ? AP(extra_row, const int32_t*)[b] : -1;
  if (xr >= 0 && AP(extra_step, const int32_t*)) xr = AP(extra_step, const int32_t*)[0];
  const int ld_extra = AI(ld_extra), kx_off = AI(kx_off), vx_off = AI(vx_off), k_off = AI(k_off), v_off = AI(v_off);
  // K (which = 0) or V (which = 1) vector `idx` of the unit's head: row idx >> log2_vpr, 8 channels from (idx & (vpr - 1)) * 8
  auto kv_ptr = [&](int idx, int which) __attribute__((always_inline)) -> const GT* {
    const int r = idx >> log2_vpr, c = (idx & (vpr - 1)) * 8;
    const bool ex = (xr >= 0 && r == Nk - 1);
    const GT* base = ex ? xp_ : (which ? vp_ : kp_);
    const unsigned row = ex ? (unsigned)xr * (unsigned)ld_extra : (unsigned)(kvbase + r) * (unsigned)ldkv;
    const unsigned col = (unsigned)((ex ? (which ? vx_off : kx_off) : (which ? v_off : k_off)) + hd + c);
    return base + ((size_t)row + col);
  };
  float uk[8], bk[8], uv[8], bv[8], uq[8], bq[8];
  const int kr0 = tid >> log2_vpr, kc0 = (tid & (vpr - 1)) * 8;            // vector 0 of this thread (the only one with a K/V finish)
  if (fin_kv && tid < nkv) {
    load8(AP(ln_u, const float*) + k_off + hd + kc0, uk);
    load8(AP(ln_b, const float*) + k_off + hd + kc0, bk);
    load8(AP(ln_u, const float*) + v_off + hd + kc0, uv);
    load8(AP(ln_b, const float*) + v_off + hd + kc0, bv);
  }
  if (fin_q && tid < nqv) {
    load8(AP(ln_u, const float*) + q_off + hd + kc0, uq);
    load8(AP(ln_b, const float*) + q_off + hd + kc0, bq);
  }
  // ---- LDS the matrix cores read but nobody stores: Q rows >= nq, K rows >= Nk, V^T keys >= Nk (LDS is free: the previous unit
  // ended on a barrier) ------------------------------------------------------------------------------------------------------------
  {
    const float z8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (d < 32) {
      for (int i = tid; i < QCHUNK * (dq >> 3); i += NT) store8(q_s + i * 8, z8);
      for (int i = tid; i < NKP * (dq >> 3); i += NT) store8(kv_s + i * 8, z8);
      __syncthreads();
    } else {
      for (int i = tid; i < (QCHUNK - nq) * vpr; i += NT) store8(q_s + (nq + (i >> log2_vpr)) * dq + (i & (vpr - 1)) * 8, z8);
      for (int i = tid; i < (NKP - Nk) * vpr; i += NT) store8(kv_s + (Nk + (i >> log2_vpr)) * dq + (i & (vpr - 1)) * 8, z8);
    }
  }
  auto zero_vt_pad = [&]() __attribute__((always_inline)) {
    if (d < 16) {
      for (int i = tid; i < DC * vt; i += NT) vt_s[i] = to_elem<T>(0.f);
      __syncthreads();                                  // (the whole region is cleared there; the real columns follow)
    } else {
      for (int i = tid; i < d * 32; i += NT) {
        const int c = i >> 5, j = Nk + (i & 31);
        if (j < NKP) vt_s[c * vt + (j ^ (((c >> 3) & 3) << 3))] = to_elem<T>(0.f);
      }
    }
  };
  auto finish = [&](float (&x)[8], const float2 st, const float (&uu)[8], const float (&bb)[8]) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (x[e] - st.x * uu[e]) * st.y + bb[e];
  };
  // one K vector into its LDS row / one V vector into V^T (fin0: vector 0 of a thread takes the deferred LayerNorm finish from st_s)
  auto put_k = [&](const Raw8<GT>& raw, int idx, bool fin0) __attribute__((always_inline)) {
    const int r = idx >> log2_vpr, c = (idx & (vpr - 1)) * 8;
    float x[8];
    raw_to_float(raw, x);
    if (fin0) finish(x, st_s[r], uk, bk);
    store8(kv_s + r * dq + c, x);
  };
  auto put_vt = [&](const Raw8<GT>& raw, int idx, bool fin0) __attribute__((always_inline)) {
    const int r = idx >> log2_vpr, c = (idx & (vpr - 1)) * 8;
    float x[8];
    raw_to_float(raw, x);
    if (fin0) finish(x, st_s[r], uv, bv);
    // the 8-key blocks of a row are permuted by the row's channel block (XOR inside each 32-key group): the 16 channel blocks a
    // wave transposes at once would otherwise land on two LDS banks
    const int rs = r ^ (((c >> 3) & 3) << 3);
#pragma unroll
    for (int e = 0; e < 8; ++e) vt_s[(c + e) * vt + rs] = to_elem<T>(x[e]);
  };
  // the cached K / V of a cross-attention (not produced in this launch): plain loads, staged right away
  auto stage_cached = [&](int which) __attribute__((always_inline)) {
    Raw8<GT> raw[MAXVA];
#pragma unroll
    for (int i = 0; i < MAXVA; ++i) {
      if (tid + i * NT < nkv) ld_plain(raw[i], kv_ptr(tid + i * NT, which));
    }
#pragma unroll
    for (int i = 0; i < MAXVA; ++i) {
      if (tid + i * NT < nkv) {
        if (which) put_vt(raw[i], tid + i * NT, false);
        else put_k(raw[i], tid + i * NT, false);
      }
    }
  };
  if (vsep) zero_vt_pad();
  if (!kv_live) {                                     // cross-attention: K (and V^T with a region of its own) are in LDS before the wait ends
    stage_cached(0);
    if (vsep) stage_cached(1);
  }
  // self-attention: K / V of at most MAXVL vectors per thread arrive with the polled loads
  constexpr int MAXVL = JEN1_DEEP_ATTN_LIVE_VEC;
  Raw8<GT> kl[MAXVL], vl[MAXVL], qraw;
  // LayerNorm statistics of the rows the finish needs: 16 lanes per row over the ln_C leading columns of q's tensor; the first
  // LNB vectors per lane ride in the same round of polled loads as q / k / v (bf16: a whole 1024-column row)
  constexpr int LNB = sizeof(GT) == 2 ? 4 : 2;
  const bool fin = fin_q || fin_kv;
  const int rs0 = fin_kv ? 0 : q0, rsn = fin ? (fin_kv ? Nk : nq) : 0;
  const int nvec = ln_C >> 3;
  const int lrow = tid >> 4;
  const bool lhas = lrow < rsn;
  const GT* lrowp = qp + (size_t)((unsigned)(b * Nq + rs0 + (lhas ? lrow : 0)) * (unsigned)ldq);
  Raw8<GT> lx[LNB];

  DK_STAMP(sy, 1);

#ifdef JEN1_DEEP_EXP_NOATTN
  if (sy.wg == 100000) {
#else
  {
#endif
  // ---- live operands: repeated by the wave until no word is the sentinel ---------------------------------------------------------
  {
    unsigned spins = 0;
    bool bad;
    do {
      bad = false;
      if (kv_live) {
#pragma unroll
        for (int i = 0; i < MAXVL; ++i) {
          if (tid + i * NT < nkv) { ld_live(kl[i], kv_ptr(tid + i * NT, 0)); ld_live(vl[i], kv_ptr(tid + i * NT, 1)); }
          else { zero_raw(kl[i]); zero_raw(vl[i]); }
        }
      }
      if (tid < nqv) ld_live(qraw, qp + ((size_t)((unsigned)(b * Nq + q0 + kr0) * (unsigned)ldq) + (unsigned)(q_off + hd + kc0)));
      else zero_raw(qraw);
      if (lhas) {
#pragma unroll
        for (int k = 0; k < LNB; ++k) {
          if (li + 16 * k < nvec) ld_live(lx[k], lrowp + (li + 16 * k) * 8);
          else zero_raw(lx[k]);
        }
      }
      if (kv_live) {
#pragma unroll
        for (int i = 0; i < MAXVL; ++i) bad |= raw_bad(kl[i]) | raw_bad(vl[i]);
      }
      bad |= q_live && raw_bad(qraw);
      if (lhas) {
#pragma unroll
        for (int k = 0; k < LNB; ++k) bad |= q_live && raw_bad(lx[k]);
      }
    } while (poll_again(sy, bad, spins));
  }
  DK_STAMP(sy, 2);
  if (fin) {
    const float inv_c = 1.0f / (float)ln_C;
    for (int r = lrow; r < rsn; r += NT / 16) {
      const GT* rowp = qp + (size_t)((unsigned)(b * Nq + rs0 + r) * (unsigned)ldq);
      float s = 0.f, q2 = 0.f;
      for (int v0 = li; v0 < nvec; v0 += 16 * LNB) {
        Raw8<GT> x[LNB];
        if (r == lrow && v0 == li) {
#pragma unroll
          for (int k = 0; k < LNB; ++k) x[k] = lx[k];                  // the chunk that came with the first round
        } else {                                                       // (long rows in float32 / more than 32 rows: further rounds)
          unsigned spins = 0;
          bool bad;
          do {
            bad = false;
#pragma unroll
            for (int k = 0; k < LNB; ++k) {
              if (v0 + 16 * k < nvec) ld_live(x[k], rowp + (v0 + 16 * k) * 8);
              else zero_raw(x[k]);
            }
#pragma unroll
            for (int k = 0; k < LNB; ++k) bad |= q_live && raw_bad(x[k]);
          } while (poll_again(sy, bad, spins));
        }
#pragma unroll
        for (int k = 0; k < LNB; ++k) {
          float f[8];
          raw_to_float(x[k], f);
#pragma unroll
          for (int j = 0; j < 8; ++j) { s += f[j]; q2 += f[j] * f[j]; }
        }
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
    __syncthreads();                                                   // the finish below reads other threads' row statistics
  }
  // ---- K (and V^T) of a self-attention, Q -> LDS -----------------------------------------------------------------------------------
  if (kv_live) {
#pragma unroll
    for (int i = 0; i < MAXVL; ++i) {
      if (tid + i * NT < nkv) {
        put_k(kl[i], tid + i * NT, i == 0 && fin_kv);
        if (vsep) put_vt(vl[i], tid + i * NT, i == 0 && fin_kv);
      }
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
  DK_STAMP(sy, 10);
  // ---- without a region of its own V replaces K in LDS now, transposed: V^T [d][keys] is the B operand of P V --------------------------
  if (!vsep) {
    zero_vt_pad();
    if (kv_live) {
#pragma unroll
      for (int i = 0; i < MAXVL; ++i) {
        if (tid + i * NT < nkv) put_vt(vl[i], tid + i * NT, i == 0 && fin_kv);
      }
    } else {
      stage_cached(1);                                 // (read again: an L2 hit; the float32 mode's large contexts come here)
    }
  }
  DK_STAMP(sy, 11);
  // ---- softmax in float32 (blocks.py:367-371): 16 lanes per row, 32 rows per pass over the 8 waves ---------------------------------
  {
    const int rsel = lane >> 4;
    const int kpl = ((NKP >> 4) + 3) & ~3;
    const int j0 = li * kpl;
    const int r = wave * 4 + rsel;
    if (r < QCHUNK) {
      T* pr = p_s + r * vt;
      if (r >= nq) {
        for (int j = li; j < NKP; j += 16) pr[j] = to_elem<T>(0.f);
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
        const float inv = (PRECISE ? 1.0f / sum : __builtin_amdgcn_rcpf(sum)) * PS;
#pragma unroll
        for (int t = 0; t < 12; ++t) {
          if (t < kpl && j0 + t < NKP) pr[j0 + t] = to_elem<T>(e[t] * inv);
        }
      }
    }
  }
  __syncthreads();
  DK_STAMP(sy, 12);
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
        dlds(vb, vt_s + (ct * 16 + li) * vt + ((j + lg * 8) ^ ((((ct * 16 + li) >> 3) & 3) << 3)));
        dmma(acc, pa, vb);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qi = qt * 16 + lg * 4 + r;
        o_s[qi * dq + ct * 16 + li] = (GT)(F8 ? acc[r] * (1.0f / P_SCALE) : acc[r]);
      }
    }
  }
  publish_next();
  __syncthreads();
  DK_STAMP(sy, 4);
  if (tid < nqv) {
    GT* op = outp + ((size_t)((unsigned)(b * Nq + q0 + kr0) * (unsigned)ldo) + (unsigned)(hd + kc0));
    st_live8(op, o_s + kr0 * dq + kc0);
  }
  }
  publish_next();
  __syncthreads();
  DK_STAMP(sy, 5);
  DK_STAMP(sy, 6);
}

// =====================================================================================================================
// statistics unit: GroupNorm fine-group (sum, sumsq) of one batch element of the chain's last tensor, for the
// launch-per-layer consumer that follows the persistent launch (same layout as jen1_conv_args.gn_stats*)
// =====================================================================================================================
template <typename T, typename FPub>
__device__ __forceinline__ void stats_unit(const unsigned char* D, int u, Sync& sy, FPub publish_next, int tid) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const jen1_deep_phase* P = reinterpret_cast<const jen1_deep_phase*>(D);
  const int b = u, L = P->sL, ld = P->sld, cpf = P->scpf, gran = P->sgran;
  typedef typename Mode<T>::G GT;
  const GT* sx = reinterpret_cast<const GT*>(P->sx);
  float* const sstats = P->sstats;
  DK_STAMP(sy, 0);
  DK_STAMP(sy, 1);
  DK_STAMP(sy, 2);
  const int VPR = ld >> 3;                       // power of two <= NT (checked on the host)
  const int vc = tid & (VPR - 1), r0 = tid / VPR, rstep = NT / VPR;
  const int sub = 8 / gran;                      // granules per vector (1, 2, 4 or 8)
  float s[8], q[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { s[k] = 0.f; q[k] = 0.f; }
  const GT* xp = sx + (size_t)b * L * ld + vc * 8;
  for (int r = r0; r < L; r += rstep) {
    Raw8<GT> x;
    unsigned spins = 0;
    do {
      ld_live(x, xp + (size_t)r * ld);
    } while (poll_again(sy, raw_bad(x), spins));      // (the chain's last tensor: always produced inside the launch)
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
  __syncthreads();
  DK_STAMP(sy, 5);
  DK_STAMP(sy, 6);
}

// =====================================================================================================================
// tile unit (JEN1_DEEP_TILE): a layer of a LONG level (include/jen1_deep.h).  All output channels of a block x a tile of tb positions
// of one batch element.  Same algebra and LDS structure as tile_gemm.hip (the activation tile + conv halo is staged once, taps are
// row-shifted views, weights come through a register ring with scalar offsets), but inside the persistent launch:
//   * the 8 waves form a (BM / 32) x (8 / (BM / 32)) grid: a wave owns 32 output rows (two M tiles) and half (BM = 128) or all
//     (BM = 256) of the tile's 16-position fragments -- an activation fragment read from LDS feeds two matrix instructions and is
//     read by 4 waves, not 8 (the loop is LDS-bandwidth bound: with one M tile per wave it took 2.4 us per 48 positions);
//   * the weight ring (24 fragments per wave: all of K at 128 channels x 3 taps) and gamma / beta / FiLM are requested before the
//     dependency wait; deeper K streams through the same ring from L2;
//   * GroupNorm statistics arrive as the producers' per-tile partial sums (8-byte (sum, sumsq) words, poisoned like every tensor of
//     the launch): a thread adds the partials of ONE statistics group over the tiles in ascending order, the lanes / waves holding the
//     same group are merged by a fixed shuffle tree and a fixed LDS pass -- bit-reproducible;
//   * the unit's own output partials leave as one 8-byte word per statistics group, written last.
// T: element type of activations, weights and the staged tile (bf16 or float; the JEN1_FP8 mode runs these units in bf16).
// =====================================================================================================================
#define TI(f) ph_i32<offsetof(jen1_deep_phase, f)>(pr)
#define TF(f) ph_f32<offsetof(jen1_deep_phase, f)>(pr)
#define TP(f, type) ph_ptr<offsetof(jen1_deep_phase, f), type>(pr)
#ifndef JEN1_TILE_RING_B
#define JEN1_TILE_RING_B 24
#endif
#ifndef JEN1_TILE_RING_F
#define JEN1_TILE_RING_F 8
#endif
template <typename T> struct TileCfg {             // bf16: 24 weight fragments of 4 registers (12 k-steps x 2 M tiles), 4 staging vectors of 4 registers
  static constexpr int RING = JEN1_TILE_RING_B, VB = 4;
};
template <> struct TileCfg<float> {                // float32: fragments and vectors are twice as wide
  static constexpr int RING = JEN1_TILE_RING_F, VB = 2;
};
constexpr int TILE_SP = 4;             // statistics partials (8 bytes each) per thread, source and polled round
constexpr int TILE_MF = 2;             // 16-row M tiles per wave

template <typename T, typename FPub>
__device__ __forceinline__ void tile_unit(const unsigned char* D, int u, Sync& sy, typename DFrag<T>::type (&ring)[TileCfg<T>::RING], FPub publish_next,
                                          int tid) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef typename DFrag<T>::type Frag;
  constexpr bool PRECISE = is_f32<T>::value;
  constexpr unsigned ES = sizeof(T);
  constexpr unsigned BLK = 512 * ES;
  constexpr int MF = TILE_MF;
  constexpr int TILE_VB = TileCfg<T>::VB;             // staging vectors per thread per polled batch
  constexpr int PFT = TileCfg<T>::RING / MF;          // k-steps in the ring
  const int lane = tid & 63;
  const int wv = rfl(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  DK_STAMP(sy, 0);
  const PhaseRegs pr = phase_regs(D, lane);
  // ---- unit -> (M block, batch element, tile) ------------------------------------------------------------------------------------
  const int B = TI(h.B), tiles_t = TI(tl.tiles_t), tb = TI(tl.tb);
  const int bt = B * tiles_t;
  const int mblk = (int)(((float)u + 0.5f) * TF(tl.inv_bt));
  const int rem = u - mblk * bt;
  const int b = (int)(((float)rem + 0.5f) * TF(tl.inv_tiles_t));
  const int tt = rem - b * tiles_t;
  const int L_in = TI(h.L_in), L_out = TI(h.L_out), stride = TI(h.stride), taps = TI(tl.taps), pad_left = TI(tl.pad_left);
  const int t0 = tt * tb;
  const int cmain = TI(tl.cmain), call = TI(h.Ctot), ldsld = TI(h.pitch), kch = TI(tl.kch), KS = TI(tl.KS), NF = TI(tl.NF);
  const int rows_in = TI(tl.rows_in);
  const int tin0 = t0 * stride - pad_left;
  const int MT = TI(h.MT), mtb = TI(tl.BM) >> 4;
  // wave grid: WM = BM / 32 waves along M, WN = 8 / WM along the positions
  const int WM = mtb >> 1;
  const int wm = wv & (WM - 1), wn = WM == NW ? 0 : (wv >> 2);
  const int NFh = WM == NW ? NF : ((NF + 1) >> 1);    // fragments of the first wave column
  const int nf0 = wn * NFh;                           // this wave's first fragment and how many it owns
  const int nfw = (NF - nf0) < NFh ? (NF - nf0) : NFh;
  const int mt0 = mblk * mtb + wm * MF;               // this wave's first 16-row tile of M
  unsigned char* ws = smem + WS_OFF;
  T* tile = reinterpret_cast<T*>(ws);
  float* tabA = reinterpret_cast<float*>(ws + TI(tl.tab_off));
  float* tabS = tabA + cmain;
  float* red = reinterpret_cast<float*>(ws + TI(tl.red_off));     // [2 sources][NW][32][2] partial merge, later [NW * MF][4] output partials

  // ---- (1) weight ring (filled behind the first polled round, below) ----------------------------------------------------------------------
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(TP(h.w, const void*)), 0, TI(h.w_bytes), RSRC_FLAGS);
  unsigned voffA[MF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) voffA[mf] = (nfw > 0 && mt0 + mf < MT) ? (unsigned)(mt0 + mf) * BLK + (unsigned)lane * (8u * ES) : OOB;
  const unsigned stepA = (unsigned)MT * BLK;
  unsigned soffA = 0;
  int issuedA = 0;
  auto issueA = [&](int slot) __attribute__((always_inline)) {
    const bool in = issuedA < KS;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) wload(ring[slot * MF + mf], rw, in ? voffA[mf] : OOB, in ? soffA : 0u);
    soffA += stepA;
    ++issuedA;
  };
  const int pro_mode = TI(h.pro_mode);
  const bool gn = pro_mode == JEN1_PRO_GN || pro_mode == JEN1_PRO_GN_SILU;
  const bool do_silu = pro_mode == JEN1_PRO_GN_SILU;
  // ---- (3) addresses of the staging vectors --------------------------------------------------------------------------------------------
  typedef T GT;
  const int vpr = call >> 3;
  const float inv_vpr = TF(h.inv_vpr);
  const int nvec = rows_in * vpr;
  const int live = TI(h.live_mask);
  const int nsrc = TI(h.nsrc);
  const void* sx1 = TP(h.src[1].x, const void*); const int sld1 = TI(h.src[1].ld), sco1 = TI(h.src[1].coff);
  const void* sx2 = TP(h.src[2].x, const void*); const int sld2 = TI(h.src[2].ld), sco2 = TI(h.src[2].coff);
  const void* sx3 = TP(h.src[3].x, const void*); const int sld3 = TI(h.src[3].ld), sco3 = TI(h.src[3].coff);
  const void* sx0 = TP(h.src[0].x, const void*); const int sld0 = TI(h.src[0].ld);
  const float sc1 = TF(h.src[1].scale);
  const bool two_main = nsrc > 1 && sco1 < cmain;      // the second source is normalised / tapped like the first (the skip)
  // vector v of the tile: row v / vpr, 8 channels from (v % vpr) * 8: global address; LDS element offset, or -1 - offset for a row
  // of the zero padding; whether the tensor is written inside this launch
  auto vec_of = [&](int v, const GT*& gp, bool& lv, int& c) __attribute__((always_inline)) -> int {
    const int vv = v < nvec ? v : 0;
    const int row = (int)(((float)vv + 0.5f) * inv_vpr);
    c = (vv - row * vpr) * 8;
    const int tin = tin0 + row;
    const bool ok = tin >= 0 && tin < L_in;
    const void* xp = sx0;
    int ld = sld0, coff = 0, k = 0;
    if (nsrc > 1 && c >= sco1) { xp = sx1; ld = sld1; coff = sco1; k = 1; }
    if (nsrc > 2 && c >= sco2) { xp = sx2; ld = sld2; coff = sco2; k = 2; }
    if (nsrc > 3 && c >= sco3) { xp = sx3; ld = sld3; coff = sco3; k = 3; }
    lv = ok && v < nvec && ((live >> k) & 1);
    gp = reinterpret_cast<const GT*>(xp) + ((unsigned)(b * L_in + (ok ? tin : 0)) * (unsigned)ld + (unsigned)(c - coff));
    const int to = row * ldsld + c;
    return ok ? to : -1 - to;
  };
  struct Batch {
    Raw8<GT> x[TILE_VB];
    const GT* gp[TILE_VB];
    int to[TILE_VB], c[TILE_VB];
    bool lv[TILE_VB];
  };
  auto prep_batch = [&](Batch& bt_, int v0) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < TILE_VB; ++k) {
      if (v0 + k * NT < nvec) bt_.to[k] = vec_of(v0 + k * NT + tid, bt_.gp[k], bt_.lv[k], bt_.c[k]);
    }
  };
  auto load_batch = [&](Batch& bt_, int v0) __attribute__((always_inline)) -> bool {      // one round of loads; true = a sentinel was seen
    bool bad = false;
#pragma unroll
    for (int k = 0; k < TILE_VB; ++k) {
      if (v0 + k * NT < nvec) {
        ld_live(bt_.x[k], bt_.gp[k]);
        bad |= bt_.lv[k] && raw_bad(bt_.x[k]);
      }
    }
    return bad;
  };
  Batch cur;
  prep_batch(cur, 0);

  // ---- (4) statistics partials of the batch element + the first staging batch: one polled round ------------------------------------------
  const int st_live = TI(tl.st_live);
  const int nt0 = TI(tl.st_tiles[0]), nt1 = TI(tl.st_tiles[1]);
  const int nfg0 = TI(tl.st_nfg[0]), nfg1 = TI(tl.st_nfg[1]);
  const int np0 = gn ? (nt0 ? nt0 : 1) * nfg0 : 0;                            // 8-byte pairs of source 0 / 1 for this batch element
  const int np1 = (gn && two_main) ? (nt1 ? nt1 : 1) * nfg1 : 0;
  const gu64* q0 = g64(TP(tl.st[0], const float*)) + (size_t)b * (nt0 ? np0 : 32);
  const gu64* q1 = np1 ? g64(TP(tl.st[1], const float*)) + (size_t)b * (nt1 ? np1 : 32) : q0;
  float s0 = 0.f, qq0 = 0.f, s1 = 0.f, qq1 = 0.f;                             // this thread's share of its statistics group
  DK_STAMP(sy, 1);
  float g1 = 0.f, g2 = 0.f;                            // p1 / p2 of channel tid (cmain <= NT is checked on the host)
  f32x4 bias4[MF];
  {
    u64 w0[TILE_SP], w1[TILE_SP];
    // the first polled round is issued BEFORE everything that no other workgroup writes (weight ring, gamma / beta / FiLM, bias): those
    // requests ride behind it instead of delaying it, and their latency hides behind the wait
    auto issue_poll = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int k = 0; k < TILE_SP; ++k) {
        const int i = tid + k * NT;
        if (k * NT < np0) w0[k] = i < np0 ? __hip_atomic_load(q0 + i, RLX_AGENT) : 0ull;
        if (k * NT < np1) w1[k] = i < np1 ? __hip_atomic_load(q1 + i, RLX_AGENT) : 0ull;
      }
#pragma unroll
      for (int k = 0; k < TILE_VB; ++k) {
        if (k * NT < nvec) ld_live(cur.x[k], cur.gp[k]);
      }
    };
    auto check_poll = [&]() __attribute__((always_inline)) -> bool {
      bool bad = false;
#pragma unroll
      for (int k = 0; k < TILE_SP; ++k) {
        if (k * NT < np0) bad |= (st_live & 1) && w0[k] == POISON;
        if (k * NT < np1) bad |= (st_live & 2) && w1[k] == POISON;
      }
#pragma unroll
      for (int k = 0; k < TILE_VB; ++k) {
        if (k * NT < nvec) bad |= cur.lv[k] && raw_bad(cur.x[k]);
      }
      return bad;
    };
    issue_poll();
    // ---- weight ring: the first PFT k-steps of this wave's M tiles; gamma / beta or the fused GroupNorm-FiLM row; bias ----------------------
#pragma unroll
    for (int s_ = 0; s_ < PFT; ++s_) issueA(s_);
    if (gn && tid < cmain) {
      const int p_ld = TI(h.p_ld);
      const int* fstep = TP(h.film_step, const int*);
      const int* frow = TP(h.film_row, const int*);
      const int fr = p_ld ? (fstep ? fstep[0] : (frow ? frow[b] : b)) : 0;
      const size_t po = (size_t)((unsigned)fr * (unsigned)p_ld) + (unsigned)tid;
      g1 = TP(h.p1, const float*)[po];
      g2 = TP(h.p2, const float*)[po];
    }
    {
      const float* biasp = TP(h.bias, const float*);
      const int out_C0 = TI(h.out_C), ps_f0 = TI(h.ps_f);
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        const int m = (mt0 + mf) * 16 + lg * 4;
        int ph = 0;
        for (int k = 1; k < ps_f0; ++k) ph += (m >= k * out_C0) ? 1 : 0;
        bias4[mf] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (biasp && nfw > 0 && mt0 + mf < MT) bias4[mf] = *reinterpret_cast<const f32x4*>(biasp + (m - ph * out_C0));
      }
    }
    DK_STAMP(sy, 7);
    unsigned spins = 0;
    bool bad = check_poll();
    while (poll_again(sy, bad, spins)) {
      issue_poll();
      bad = check_poll();
    }
#pragma unroll
    for (int k = 0; k < TILE_SP; ++k) {
      if (k * NT < np0) { s0 += __uint_as_float((unsigned)w0[k]); qq0 += __uint_as_float((unsigned)(w0[k] >> 32)); }
      if (k * NT < np1) { s1 += __uint_as_float((unsigned)w1[k]); qq1 += __uint_as_float((unsigned)(w1[k] >> 32)); }
    }
    // batch elements with more than TILE_SP * NT partials per source (very long inputs): further rounds
    for (int i0 = TILE_SP * NT; i0 < np0 || i0 < np1; i0 += NT) {
      u64 a0 = 0, a1 = 0;
      unsigned sp2 = 0;
      do {
        bad = false;
        const int i = i0 + tid;
        if (i0 < np0) { a0 = i < np0 ? __hip_atomic_load(q0 + i, RLX_AGENT) : 0ull; bad |= (st_live & 1) && a0 == POISON; }
        if (i0 < np1) { a1 = i < np1 ? __hip_atomic_load(q1 + i, RLX_AGENT) : 0ull; bad |= (st_live & 2) && a1 == POISON; }
      } while (poll_again(sy, bad, sp2));
      s0 += __uint_as_float((unsigned)a0); qq0 += __uint_as_float((unsigned)(a0 >> 32));
      s1 += __uint_as_float((unsigned)a1); qq1 += __uint_as_float((unsigned)(a1 >> 32));
    }
  }
  DK_STAMP(sy, 2);
  // ---- (5) group sums -> affine tables  y = silu?(A[c] x + S[c])  (blocks.py:137-145) ---------------------------------------------------
  if (gn) {
    // threads t, t + nfg, ... hold the same statistics group (NT is a multiple of nfg): merge inside the wave (fixed tree), the
    // waves' sums meet in LDS and every channel adds them in wave order
    auto wave_merge = [&](float v, int nfg) __attribute__((always_inline)) -> float {
      v += __shfl_xor(v, 32);                       // (nfg is 8, 16 or 32: lanes l, l + nfg, ... of a wave hold group l mod nfg)
      if (nfg <= 16) v += __shfl_xor(v, 16);
      if (nfg <= 8) v += __shfl_xor(v, 8);
      return v;
    };
    s0 = wave_merge(s0, nfg0); qq0 = wave_merge(qq0, nfg0);
    if (lane < nfg0) *reinterpret_cast<float2*>(red + (wv * 32 + lane) * 2) = make_float2(s0, qq0);
    if (np1) {
      s1 = wave_merge(s1, nfg1); qq1 = wave_merge(qq1, nfg1);
      if (lane < nfg1) *reinterpret_cast<float2*>(red + NW * 64 + (wv * 32 + lane) * 2) = make_float2(s1, qq1);
    }
    __syncthreads();
    if (tid < cmain) {
      const int c = tid;
      const int cpg = TI(h.gn_cpg), groups = TI(h.gn_groups);
      const int gch = (int)(((float)c + 0.5f) * TF(tl.inv_cpg));
      const int g = gch < groups ? gch : groups - 1;
      const int lo0 = g * cpg;
      const bool k1 = np1 > 0 && lo0 >= sco1;                                  // the group lies in the second source
      const int lo = lo0 - (k1 ? sco1 : 0);
      const float icps = k1 ? TF(tl.inv_cps[1]) : TF(tl.inv_cps[0]);
      const int nfg = k1 ? nfg1 : nfg0;
      const int f0 = (int)(((float)lo + 0.5f) * icps);
      int cnt = (int)(((float)cpg + 0.5f) * icps);
      cnt = cnt < 1 ? 1 : cnt;
      const float* rk = red + (k1 ? NW * 64 : 0);
      float s = 0.f, q = 0.f;
      for (int f = f0; f < f0 + cnt && f < nfg; ++f) {
        float fs = 0.f, fq = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < NW; ++w2) {
          const float2 e2 = *reinterpret_cast<const float2*>(rk + (w2 * 32 + f) * 2);
          fs += e2.x; fq += e2.y;
        }
        s += fs; q += fq;
      }
      const float sc = k1 ? sc1 : 1.0f;
      s *= sc;
      q *= sc * sc;
      const float inv_count = TF(h.inv_count);
      const float mean = s * inv_count;
      float var = q * inv_count - mean * mean;
      var = var < 0.f ? 0.f : var;
      const float rstd = PRECISE ? 1.0f / sqrtf(var + TF(h.gn_eps)) : rsqrtf(var + TF(h.gn_eps));
      const float A = rstd * g1;
      tabA[c] = A * sc;
      tabS[c] = g2 - mean * A;
    }
    __syncthreads();
  }
  DK_STAMP(sy, 8);
  // ---- (6) stage the tile: prologue applied once, zero padding applied after it (most layers are ONE batch of vectors: it came with
  // the statistics; longer tiles poll their further batches one after the other) ------------------------------------------------------
  for (int v0 = 0; v0 < nvec; v0 += NT * TILE_VB) {
    if (v0 > 0) {
      prep_batch(cur, v0);
      unsigned spins = 0;
      bool bad;
      do {
        bad = load_batch(cur, v0);
      } while (poll_again(sy, bad, spins));
    }
#pragma unroll
    for (int k = 0; k < TILE_VB; ++k) {
      const int v = v0 + k * NT + tid;
      if (v0 + k * NT >= nvec) continue;
      if (v >= nvec) continue;
      const int c = cur.c[k];
      float x[8];
      raw_to_float(cur.x[k], x);
      int to = cur.to[k];
      if (to >= 0) {
        if (c < cmain) {
          if (gn) {
            const float4 a0 = *reinterpret_cast<const float4*>(tabA + c), a1 = *reinterpret_cast<const float4*>(tabA + c + 4);
            const float4 e0 = *reinterpret_cast<const float4*>(tabS + c), e1 = *reinterpret_cast<const float4*>(tabS + c + 4);
            x[0] = x[0] * a0.x + e0.x; x[1] = x[1] * a0.y + e0.y; x[2] = x[2] * a0.z + e0.z; x[3] = x[3] * a0.w + e0.w;
            x[4] = x[4] * a1.x + e1.x; x[5] = x[5] * a1.y + e1.y; x[6] = x[6] * a1.z + e1.z; x[7] = x[7] * a1.w + e1.w;
            if (do_silu) {
#pragma unroll
              for (int j = 0; j < 8; ++j) x[j] = PRECISE ? silu_precise(x[j]) : silu_f(x[j]);
            }
          } else if (two_main && c >= sco1 && sc1 != 1.0f) {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] *= sc1;
          }
        }
      } else {
        to = -1 - to;
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = 0.f;
      }
      store8(tile + to, x);
    }
  }
  DK_STAMP(sy, 11);
  // ---- (7) .. (9): MFMA loop and epilogue, specialised by the number of fragments this wave owns ---------------------------------------------
  const int out_C = TI(h.out_C), ps_f = TI(h.ps_f), ps_off = TI(h.ps_off), L_y = TI(h.L_y), y_brows = TI(h.y_brows), y_row0 = TI(h.y_row0);
  const GT* resb = TP(h.residual, const GT*);
  const bool reslive = resb && ((live >> 8) & 1);
  const int ld_res = TI(h.ld_res);
  float* const out_part = TP(tl.out_part, float*);
  auto body = [&](auto nfc) __attribute__((always_inline)) {
    constexpr int NFW = decltype(nfc)::value;
    int ldsrow[NFW];
#pragma unroll
    for (int nf = 0; nf < NFW; ++nf) {
      const int n = (nf0 + nf) * 16 + li;
      ldsrow[nf] = ((n < tb ? n : 0) * stride) * ldsld + lg * 8;
    }
    __syncthreads();
    DK_STAMP(sy, 3);
    // MFMA loop: weights from the ring, activations from row-shifted views of the LDS tile; k-steps in (tap, chunk) order, behind the
    // last tap the extra chunks follow at the centre row, columns cmain + 32 j
    f32x4 acc[MF][NFW];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf) acc[mf][nf] = bias4[mf];                 // the accumulators start from the bias
    {
      int c_tap = 0, c_kc = 0;
      bool in_extra = false;
      const bool xs = call > cmain;
      auto kstep = [&](int slot) __attribute__((always_inline)) {
        const T* bp = tile + c_tap * ldsld + c_kc * 32;
        Frag bfr[NFW];
#pragma unroll
        for (int nf = 0; nf < NFW; ++nf) dlds(bfr[nf], bp + ldsrow[nf]);
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
          for (int nf = 0; nf < NFW; ++nf) dmma(acc[mf][nf], ring[slot * MF + mf], bfr[nf]);
        if (++c_kc == kch && !in_extra) {
          c_kc = 0;
          if (++c_tap == taps && xs) { in_extra = true; c_tap = pad_left; c_kc = kch; }
        }
#ifndef JEN1_TILE_NO_SCHED_BARRIER
        __builtin_amdgcn_sched_barrier(0);      // (left alone the scheduler hoists the LDS reads of the whole round: ~190 registers beside the ring)
#endif
      };
      int ks = 0;
      for (; ks + PFT <= KS; ks += PFT) {               // whole rounds of the ring, straight-line
        const bool refill = ks + PFT < KS;
#pragma unroll
        for (int s_ = 0; s_ < PFT; ++s_) {
          kstep(s_);
          if (refill) issueA(s_);
        }
      }
      if (ks < KS) {                                      // the tail round
#pragma unroll
        for (int s_ = 0; s_ < PFT; ++s_) {
          if (ks + s_ < KS) kstep(s_);
        }
      }
    }
    DK_STAMP(sy, 4);
    // epilogue: bias, residual, sub-pixel row mapping, write-through stores, partial sums of the next GroupNorm
    // output rows of this lane (-1: none), the residual (nothing of the epilogue lives across the loop: registers); offsets stay 32-bit (the host checks
    // the tensor sizes) so that an address is a scalar base + one register
    int co_[MF];
    int yrow_[MF][NFW];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const int m = (mt0 + mf) * 16 + lg * 4;
      int ph = 0;
      for (int k = 1; k < ps_f; ++k) ph += (m >= k * out_C) ? 1 : 0;
      co_[mf] = m - ph * out_C;
#pragma unroll
      for (int nf = 0; nf < NFW; ++nf) {
        const int n = (nf0 + nf) * 16 + li;
        const int q = t0 + n;
        const int ty = q * ps_f + ph - ps_off;
        const bool ok = mt0 + mf < MT && n < tb && q < L_out && ty >= 0 && ty < L_y;
        yrow_[mf][nf] = ok ? b * y_brows + y_row0 + ty : -1;
      }
    }
    Raw4<GT> rr[MF][NFW];
    auto load_res = [&]() __attribute__((always_inline)) -> bool {
      bool bad = false;
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int nf = 0; nf < NFW; ++nf) {
          if (yrow_[mf][nf] >= 0) {
            const unsigned off = (unsigned)yrow_[mf][nf] * (unsigned)ld_res + (unsigned)co_[mf];
            ld_live4r(rr[mf][nf], resb + off);
            bad |= reslive && raw_bad(rr[mf][nf]);
          }
        }
      return bad;
    };
    bool rbad = false;
    if (resb) rbad = load_res();
    if (resb) {
      unsigned spins = 0;
      while (poll_again(sy, rbad, spins)) rbad = load_res();
    }
    DK_STAMP(sy, 9);
    const int ld_y = TI(h.ld_y), y_f32 = TI(h.y_f32);
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      float gs = 0.f, gq = 0.f;
      {
#pragma unroll
        for (int nf = 0; nf < NFW; ++nf) {
          if (yrow_[mf][nf] >= 0) {
            float v[4] = {acc[mf][nf][0], acc[mf][nf][1], acc[mf][nf][2], acc[mf][nf][3]};
            if (resb) {
              float r4[4];
              raw4_to_float(rr[mf][nf], r4);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] += r4[r];
            }
            const unsigned off = (unsigned)yrow_[mf][nf] * (unsigned)ld_y + (unsigned)co_[mf];
            if (y_f32) st_live4(TP(h.y, float*) + off, v);
            else st_live4(TP(h.y, GT*) + off, v);
            gs += (v[0] + v[1]) + (v[2] + v[3]);
            gq += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
          }
        }
      }
      if (mf == MF - 1) DK_STAMP(sy, 10);
      if (out_part) {
        // the 16 rows of an M tile lie in ONE statistics group (out_cps is a multiple of 16): all 64 lanes, fixed tree
        gs = row16_sum_d(gs); gq = row16_sum_d(gq);
        auto rl = [](float v, int l) __attribute__((always_inline)) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); };
        const float ts = (rl(gs, 0) + rl(gs, 16)) + (rl(gs, 32) + rl(gs, 48));       // the four rows of the wave, fixed order, no LDS round trips
        const float tq = (rl(gq, 0) + rl(gq, 16)) + (rl(gq, 32) + rl(gq, 48));
        const float fgo = mt0 + mf < MT ? (float)(int)(((float)__builtin_amdgcn_readlane(co_[mf], 0) + 0.5f) * TF(tl.inv_out_cps)) : -1.0f;
        if (lane == 0) *reinterpret_cast<float4*>(red + (wv * MF + mf) * 4) = make_float4(ts, tq, fgo, 0.f);
      }
    }
  };
  if (nfw <= 0) {                                         // (a wave column without fragments: the tile's tail)
    __syncthreads();
    if (out_part && lane == 0) {
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) red[(wv * MF + mf) * 4 + 2] = -1.0f;
    }
  } else if (nfw == 1) body(std::integral_constant<int, 1>());
#ifdef JEN1_TILE_NFW4
  else if (nfw == 2) body(std::integral_constant<int, 2>());
  else if (nfw == 3) body(std::integral_constant<int, 3>());
  else body(std::integral_constant<int, 4>());
#else
  else body(std::integral_constant<int, 2>());       // (the host keeps a wave at two fragments: tb <= 64 at BM = 128, tb <= 32 at BM = 256)
#endif
  DK_STAMP(sy, 12);
  publish_next();
  __syncthreads();
  DK_STAMP(sy, 13);
  if (out_part) {
    const int onfg = TI(tl.out_nfg), mblocks = TI(tl.mblocks);
    if (tid < onfg) {
      float s = 0.f, q = 0.f;
      float4 e4[NW * MF];
#pragma unroll
      for (int e = 0; e < NW * MF; ++e) e4[e] = *reinterpret_cast<const float4*>(red + e * 4);      // (all reads first: no dependent LDS chain)
#pragma unroll
      for (int e = 0; e < NW * MF; ++e) {
        const bool mine = e4[e].z == (float)tid;
        s += mine ? e4[e].x : 0.f;
        q += mine ? e4[e].y : 0.f;
      }
      const size_t idx = ((size_t)b * (size_t)(tiles_t * mblocks) + (size_t)(tt * mblocks + mblk)) * (size_t)onfg + (size_t)tid;
      st_word(reinterpret_cast<float*>(out_part) + 2 * idx, 0, __float_as_uint(s), __float_as_uint(q));      // (sum, sumsq) partial: same reserved word
    }
  }
  DK_STAMP(sy, 14);
  DK_STAMP(sy, 15);
  __syncthreads();            // (LDS: the next unit stages over this one's tile and scratch)
  DK_STAMP(sy, 5);
  DK_STAMP(sy, 6);
}

// KM: the unit kinds compiled into this instance (bit k: kind k).  A program of tile phases and a program of GEMM / attention phases
// are different kernels: each keeps its own register allocation (one kernel with every unit kind spills in the GEMM unit).
constexpr int KM_DEEP = (1 << JEN1_DEEP_GEMM) | (1 << JEN1_DEEP_ATTN) | (1 << JEN1_DEEP_STATS);
constexpr int KM_TILE = (1 << JEN1_DEEP_TILE) | (1 << JEN1_DEEP_STATS);
template <typename T, bool TK, int KM>      // TK: units by ticket (any number of resident workgroups) instead of the static unit -> workgroup map
__global__ __launch_bounds__(NT) void deep_kernel(const unsigned char* __restrict__ blobs, const int4* __restrict__ hdr_g, int n_phases,
                                                  unsigned* sync, unsigned* err) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef typename DFrag<T>::type Frag;
  constexpr int PF = KM == KM_TILE ? TileCfg<T>::RING : DeepCfg<T>::PF;      // the weight ring of the kernel's units
  const int tid = threadIdx.x;
  const int lane = tid & 63, wk = rfl(tid >> 6);
  const int wg = blockIdx.x, nwg = gridDim.x;
  Hdr* hdr = reinterpret_cast<Hdr*>(smem);
  if (tid < n_phases) reinterpret_cast<int4*>(smem)[tid] = hdr_g[tid];
  __syncthreads();
  Sync sy;
  (void)sync;
  sy.err = err;
  sy.dead = false;
  sy.wg = wg;
  sy.nwg = nwg;
#ifdef JEN1_DEEP_PROFILE
#pragma unroll
  for (int i_ = 0; i_ < 16; ++i_) sy.tt[i_] = 0;
#endif
  if constexpr (!TK) {
  int p = -1, u = 0;
  if (!find_next(hdr, n_phases, wg, nwg, p, u)) return;
  p = rfl(p);
  u = rfl(u);
  int slot = 0;
  reinterpret_cast<u64*>(smem + HDR_BYTES)[tid] = reinterpret_cast<const u64*>(blobs + (size_t)p * BLOB)[tid];
  __syncthreads();
  Frag ra[PF];
  if constexpr ((KM >> JEN1_DEEP_GEMM) & 1) {
    if (hdr[p].kind == JEN1_DEEP_GEMM) gemm_prefill<T>(smem + HDR_BYTES, u, wk, lane, ra);
  }
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
    // called by the unit right before one of its __syncthreads()
    auto publish_next = [&]() { if (reload) reinterpret_cast<u64*>(Dn)[tid] = nx; };
    sy.p = p;
    const int kind = hdr[p].kind;
    if (((KM >> JEN1_DEEP_GEMM) & 1) && kind == JEN1_DEEP_GEMM) {
      if constexpr ((KM >> JEN1_DEEP_GEMM) & 1) gemm_unit<T>(D, u, sy, ra, publish_next, tid);
    } else if (((KM >> JEN1_DEEP_ATTN) & 1) && kind == JEN1_DEEP_ATTN) {
      if constexpr ((KM >> JEN1_DEEP_ATTN) & 1) attn_unit<T>(D, u, sy, publish_next, tid);
    } else if (((KM >> JEN1_DEEP_TILE) & 1) && kind == JEN1_DEEP_TILE) {
      if constexpr (KM == KM_TILE) tile_unit<T>(D, u, sy, ra, publish_next, tid);
    } else {
      stats_unit<T>(D, u, sy, publish_next, tid);
    }
    DK_FLUSH(sy);
    if (!more) break;
    // the next unit's weight ring: requested right behind this unit's arrival, long before its dependency wait ends
#ifndef JEN1_DEEP_EXP_NOPRE
    if constexpr ((KM >> JEN1_DEEP_GEMM) & 1) {
      if (hdr[p2].kind == JEN1_DEEP_GEMM) gemm_prefill<T>(Dn, u2, wk, lane, ra);
    }
#endif
    p = p2;
    u = u2;
    if (reload) slot ^= 1;
  }
    return;
  }
  const int total = hdr[n_phases - 1].before + hdr[n_phases - 1].n_units;
  volatile int* tick_s = reinterpret_cast<volatile int*>(smem + TICKET_OFF);
  gu32* const ticket = g32(sync);
  if (tid == 0) tick_s[0] = (int)__hip_atomic_fetch_add(ticket, 1u, RLX_AGENT);
  __syncthreads();
  int ta = rfl(tick_s[0]);
  int p = 0, u = 0, p_loaded = -1;
  Frag ra[PF];
  unsigned char* const D = smem + HDR_BYTES;
  while (ta < total) {
    ticket_decode(hdr, n_phases, ta, p, u);
    p = rfl(p);
    u = rfl(u);
    if (p != p_loaded) {                               // the phase's descriptor + per-wave tables: global -> LDS
      __syncthreads();                                 // (a ticket read of the previous round may still be under way)
      reinterpret_cast<u64*>(D)[tid] = reinterpret_cast<const u64*>(blobs + (size_t)p * BLOB)[tid];
      __syncthreads();
      p_loaded = p;
    }
    const int kind = hdr[p].kind;
    // the unit's weight ring: requested as soon as the unit is known, long before its dependency wait ends (a workgroup that is
    // free takes the smallest open ticket, typically a phase or two ahead of the ones being computed)
#ifndef JEN1_DEEP_EXP_NOPRE
    if constexpr ((KM >> JEN1_DEEP_GEMM) & 1) {
      if (kind == JEN1_DEEP_GEMM) gemm_prefill<T>(D, u, wk, lane, ra);
    }
#endif
    // the next ticket: requested late in the unit (a workgroup must not sit on a ticket while it computes: the unit it would hold
    // is on the critical path a phase later), picked up behind the unit
    unsigned tc = 0;
    auto publish_next = [&]() {
      if (tid == 0 && tc == 0) tc = 1u + __hip_atomic_fetch_add(ticket, 1u, RLX_AGENT);
    };
    sy.p = p;
    if (((KM >> JEN1_DEEP_GEMM) & 1) && kind == JEN1_DEEP_GEMM) {
      if constexpr ((KM >> JEN1_DEEP_GEMM) & 1) gemm_unit<T>(D, u, sy, ra, publish_next, tid);
    } else if (((KM >> JEN1_DEEP_ATTN) & 1) && kind == JEN1_DEEP_ATTN) {
      if constexpr ((KM >> JEN1_DEEP_ATTN) & 1) attn_unit<T>(D, u, sy, publish_next, tid);
    } else if (((KM >> JEN1_DEEP_TILE) & 1) && kind == JEN1_DEEP_TILE) {
      if constexpr (KM == KM_TILE) tile_unit<T>(D, u, sy, ra, publish_next, tid);
    } else {
      stats_unit<T>(D, u, sy, publish_next, tid);
    }
    DK_FLUSH(sy);
    if (tid == 0) tick_s[0] = (int)(tc - 1u);
    __syncthreads();
    ta = rfl(tick_s[0]);
  }
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int align16i(int x) { return (x + 15) & ~15; }
constexpr int LDS_TOTAL = 160 * 1024;
constexpr int MAX_TRIPS = 4;                        // staging trips of the normalised part of a unit (long rows)
constexpr int LDS_BUDGET = LDS_TOTAL - WS_OFF;      // what a unit may use behind the headers and the two descriptor slots

}  // namespace

extern "C" int jen1_deep_phase_size(void) { return (int)sizeof(jen1_deep_phase); }
extern "C" int jen1_deep_has_chunks(void) {
#ifdef JEN1_DEEP_CHUNKS
  return 1;
#else
  return 0;
#endif
}
extern "C" int jen1_deep_blob_bytes(void) { return BLOB; }

#ifdef JEN1_DEEP_PROFILE
// tuning builds: [n_phases][nwg][8] uint64 stamps (see DK_STAMP); not part of the product ABI
extern "C" int jen1_deep_debug_buffer(void* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_deep_dbg), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int jen1_deep_phase_conv(const jen1_conv_args* a, int nb_max, jen1_deep_phase* out) {
  JEN1_CHECK(a && out, "deep conv: null pointer");
  JEN1_CHECK(a->dtype == JEN1_F32 || a->dtype == JEN1_BF16 || a->dtype == JEN1_FP8, "deep conv: bad dtype");
  JEN1_CHECK(a->dtype != JEN1_FP8 || (a->w_scale && !a->y_f32), "deep conv: JEN1_FP8 needs the per-row weight scales (w_scale) and a bf16 output");
  JEN1_CHECK(a->x0 && a->w && a->y, "deep conv: null tensor");
  JEN1_CHECK(a->pro_mode == JEN1_PRO_NONE || a->pro_mode == JEN1_PRO_GN || a->pro_mode == JEN1_PRO_GN_SILU,
             "deep conv: prologue %d is not supported by the persistent kernel", a->pro_mode);
  JEN1_CHECK(!a->ln_fold && !a->row_scale, "deep conv: LayerNorm-fold epilogue / row scale are not supported by the persistent kernel");
  JEN1_CHECK(a->c0 > 0 && a->c0 % 32 == 0 && a->c1 % 32 == 0 && a->M % 16 == 0, "deep conv: channels must be multiples of 32, M of 16");
  JEN1_CHECK(a->taps >= 1 && a->stride >= 1 && a->B >= 1 && a->L_in >= 1 && a->L_out >= 1, "deep conv: bad geometry");
  JEN1_CHECK(a->nseg >= 0 && a->nseg <= JEN1_DEEP_MAX_SRC - 2, "deep conv: at most %d extra K segments", JEN1_DEEP_MAX_SRC - 2);
  JEN1_CHECK(a->taps + a->nseg <= JEN1_DEEP_MAX_SEG, "deep conv: too many K segments");
  const int es = a->dtype == JEN1_F32 ? 4 : (a->dtype == JEN1_FP8 ? 1 : 2);      // staged tile / packed weights
  const int maxv = a->dtype == JEN1_F32 ? JEN1_DEEP_MAXV_F : JEN1_DEEP_MAXV_B;
  jen1_deep_phase& p = *out;
  memset(&p, 0, sizeof(p));
  p.h.wscale = a->dtype == JEN1_FP8 ? a->w_scale : nullptr;
  p.h.live_mask = a->live_mask;
  p.h.kind = JEN1_DEEP_GEMM;
  p.h.dtype = a->dtype;
  p.h.dep = -1;
  // ---- sources and segments ------------------------------------------------------------------------------------------------
  int ns = 0, coff = 0;
  p.h.src[ns++] = jen1_deep_src{a->x0, a->ld0, a->c0, 0, 1.0f};
  coff = a->c0;
  if (a->c1) {
    JEN1_CHECK(a->x1, "deep conv: c1 without x1");
    p.h.src[ns++] = jen1_deep_src{a->x1, a->ld1, a->c1, coff, a->src1_scale};
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
    p.h.src[ns++] = jen1_deep_src{e.x, e.ld, 32 * e.kch, coff, 1.0f};
    G += e.kch;
    p.seg[nseg++] = jen1_deep_seg{coff, e.shift, G, 0};
    coff += 32 * e.kch;
  }
  p.h.nsrc = ns; p.h.nseg = nseg; p.h.G = G; p.h.Ctot = coff;
  p.h.pitch = coff + 8;
  p.h.MT = a->M / 16;
  const int64_t wb = (int64_t)G * p.h.MT * 512 * es;
  JEN1_CHECK(wb < ((int64_t)1 << 31), "deep conv: packed weight too large for 31-bit offsets");
  p.h.w = a->w; p.h.w_bytes = (uint32_t)wb;
  p.h.mt_split = a->m_split / 16;
  p.h.g_split = a->m_split ? a->k_split : 0;
  JEN1_CHECK(a->m_split % 16 == 0 && p.h.g_split <= G, "deep conv: bad dual-range split");
  p.h.B = a->B; p.h.L_in = a->L_in; p.h.L_out = a->L_out; p.h.stride = a->stride;
  // ---- prologue ---------------------------------------------------------------------------------------------------------------
  p.h.pro_mode = a->pro_mode;
  if (a->pro_mode != JEN1_PRO_NONE) {
    JEN1_CHECK(a->gn_gamma && a->gn_beta && a->gn_groups >= 1 && a->gn_cpg >= 1 && a->gn_count >= 1, "deep conv: incomplete GroupNorm");
    p.h.norm_C = cmain;
    p.h.gn_groups = a->gn_groups;
    p.h.gn_cpg = a->gn_groups == 1 ? cmain : a->gn_cpg;
    p.h.gran = p.h.gn_cpg >= 8 ? 8 : p.h.gn_cpg;
    JEN1_CHECK((p.h.gn_cpg >= 8 && p.h.gn_cpg % 8 == 0) || p.h.gn_cpg == 4 || p.h.gn_cpg == 2 || p.h.gn_cpg == 1, "deep conv: group size %d", p.h.gn_cpg);
    JEN1_CHECK(a->gn_groups * p.h.gn_cpg == cmain, "deep conv: %d groups of %d channels do not tile the %d channels", a->gn_groups, p.h.gn_cpg, cmain);
    p.h.inv_count = 1.0f / (float)a->gn_count;
    p.h.gn_eps = a->gn_eps;
    p.h.inv_groups = 1.0f / (float)a->gn_groups;
    if (a->film) {
      // `film` is the FUSED table here: gamma * (scale + 1) at film_off + c, beta * (scale + 1) + shift at film_off + film_C + c
      JEN1_CHECK(a->film_C == cmain && a->film_ld >= a->film_off + 2 * a->film_C, "deep conv: bad FiLM table geometry");
      p.h.p1 = a->film + a->film_off;
      p.h.p2 = a->film + a->film_off + a->film_C;
      p.h.p_ld = a->film_ld;
      p.h.film_row = a->film_row; p.h.film_step = a->film_step;
    } else {
      p.h.p1 = a->gn_gamma; p.h.p2 = a->gn_beta; p.h.p_ld = 0;
    }
  }
  // ---- epilogue ----------------------------------------------------------------------------------------------------------------
  p.h.bias = a->bias; p.h.residual = a->residual; p.h.y = a->y;
  p.h.out_C = a->out_C; p.h.ps_f = a->ps_f < 1 ? 1 : a->ps_f; p.h.ps_off = a->ps_off; p.h.L_y = a->L_y; p.h.y_brows = a->y_brows;
  p.h.y_row0 = a->y_row0; p.h.ld_y = a->ld_y; p.h.ld_res = a->ld_res; p.h.act = a->act; p.h.y_f32 = a->y_f32;
  JEN1_CHECK(a->out_C % 4 == 0 && a->ld_y % 4 == 0 && (!a->residual || a->ld_res % 4 == 0), "deep conv: output channels / pitches must be multiples of 4");
  JEN1_CHECK((int64_t)a->B * a->L_in * (a->ld0 > a->ld1 ? a->ld0 : a->ld1) < ((int64_t)1 << 31) && (int64_t)a->B * a->y_brows * a->ld_y < ((int64_t)1 << 31),
             "deep conv: tensor too large");
  // ---- staging geometry -------------------------------------------------------------------------------------------------------
  // raw part: a thread owns one 8-channel column (vectors per row: a power of two that divides the workgroup)
  {
    const int vr = (coff - p.h.norm_C) / 8;
    JEN1_CHECK((vr & (vr - 1)) == 0 && vr <= JEN1_DEEP_THREADS, "deep conv: %d raw channels: vectors per row must be a power of two <= %d",
               coff - p.h.norm_C, JEN1_DEEP_THREADS);
    int l = 0;
    while ((1 << l) < vr) ++l;
    p.h.lvr = l;
  }
  // halo rows: a tap is a plain row offset in the staged tile
  int min_sh = 0, max_sh = 0;
  for (int s = 0; s < nseg; ++s) {
    min_sh = p.seg[s].shift < min_sh ? p.seg[s].shift : min_sh;
    max_sh = p.seg[s].shift > max_sh ? p.seg[s].shift : max_sh;
  }
  p.h.Hb = -min_sh;
  int Ha = (a->L_out - 1) * a->stride + max_sh - (a->L_in - 1);
  Ha = Ha < 0 ? 0 : Ha;
  p.h.Lp = p.h.Hb + a->L_in + Ha;
  // normalised part: (batch element, group) pairs own 2^lS consecutive lanes, a lane owns one column of the group
  int lvpg = 0, lgroups = 0;
  if (p.h.norm_C) {
    const int vpg = p.h.gn_cpg / 8;
    JEN1_CHECK(p.h.gn_cpg % 8 == 0 && (vpg & (vpg - 1)) == 0 && (a->gn_groups & (a->gn_groups - 1)) == 0 && vpg <= 128,
               "deep conv: %d groups of %d channels: the persistent kernel needs power-of-two groups of at least 8 channels", a->gn_groups, p.h.gn_cpg);
    while ((1 << lvpg) < vpg) ++lvpg;
    while ((1 << lgroups) < a->gn_groups) ++lgroups;
  }
  p.h.lvpg = lvpg; p.h.lgroups = lgroups;
  // ---- unit geometry: as many batch elements per unit as fit (a power of two; fewer, fatter units re-read the weights less) -----------
  int nb0 = 1;
  while (nb0 * 2 <= a->B) nb0 *= 2;
  if (nb_max > 0) while (nb0 > nb_max) nb0 /= 2;
  if (p.h.p_ld && !p.h.film_step) nb0 = 1;     // a per-element FiLM table: one table row per unit
  // staging in ONE trip is preferred (a further trip is a further memory round trip); several trips only when not even one batch
  // element fits otherwise (long rows: e.g. 36 positions x 512 channels)
  int max_trips = 1, nb = nb0;
  for (;; nb /= 2) {
    if (nb < 1 && max_trips == 1) { max_trips = MAX_TRIPS; nb = nb0; }
    JEN1_CHECK(nb >= 1, "deep conv: one batch element (%d rows x %d channels) does not fit a unit", a->L_in, coff);
    int n_chunks = 1, Lc = a->L_out;
#ifdef JEN1_DEEP_CHUNKS
    const bool chunks_built = true;
#else
    const bool chunks_built = false;
#endif
    if (chunks_built && nb == 1 && a->L_out > 64) {                 // one batch element is wider than a unit: column chunks (balanced)
      n_chunks = ceil_div(a->L_out, 64);
      Lc = ceil_div(a->L_out, n_chunks);
    }
    const int cols = nb * Lc;
    const int NF = ceil_div(cols, 16);
    if (NF > 4) continue;
    int lS = 6, ntrips = 1;
    if (p.h.norm_C) {
      const int pairs = nb * a->gn_groups;
      if (pairs > JEN1_DEEP_THREADS / 2) continue;                       // at least 2 lanes per pair
      int S = JEN1_DEEP_THREADS / pairs;
      const int S_max = (1 << lvpg) > 64 ? 128 : 64;                      // a lane set is one wave, or two for groups of 128 columns
      S = S > S_max ? S_max : S;
      lS = 0;
      while ((1 << lS) < S) ++lS;
      if ((1 << lvpg) > S) continue;                                      // a lane owns a column
      if (ceil_div(a->L_in << lvpg, S) > maxv * max_trips) continue;      // vectors per lane (in trips of maxv)
      ntrips = ceil_div(ceil_div(a->L_in << lvpg, S), maxv);
    }
    const int Rtot = nb * p.h.Lp + (p.h.Hb + Ha + 1);
    const int tile_b = align16i(Rtot * p.h.pitch * es);
    int red_b = (JEN1_DEEP_THREADS / 64) * NF * 1024;                    // K-reduction scratch, also the per-thread dummy vector slots
    red_b = red_b > JEN1_DEEP_THREADS * 8 * es ? red_b : JEN1_DEEP_THREADS * 8 * es;
    const int tot = tile_b + red_b;
    if (tot > LDS_BUDGET) continue;
    p.h.ntrips = ntrips;
    p.h.n_chunks = n_chunks; p.h.Lc = Lc; p.h.inv_nchunks = 1.0f / (float)n_chunks;
    p.h.nb = nb; p.h.NF = NF; p.h.R = nb * a->L_in; p.h.Rtot = Rtot; p.h.zrow = nb * p.h.Lp + p.h.Hb; p.h.lS = lS;
    p.h.red_off = tile_b;
    p.h.red_bytes = red_b;
    p.h.lds_bytes = tot;
    break;
  }
  p.h.groups_n = ceil_div(a->B, p.h.nb);
  p.h.mrep = 1;
  p.h.n_units = p.h.MT * p.h.groups_n * p.h.n_chunks;
  JEN1_CHECK(p.h.n_units < (1 << 20), "deep conv: too many units");
  p.h.inv_vpr = 1.0f / (float)(coff / 8);
  p.h.inv_Lin = 1.0f / (float)a->L_in;
  p.h.inv_Lout = 1.0f / (float)p.h.Lc;
  return 0;
}

extern "C" int jen1_deep_phase_attention(const void* q, const void* k, const void* v, void* out_t, const int32_t* kv_row, const void* kv_extra,
                                         const int32_t* extra_row, const int32_t* extra_step, int ld_extra, int kx_off, int vx_off, int B,
                                         int H, int d, int Nq, int Nk, int ldq, int q_off, int ldkv, int k_off, int v_off, int ldo,
                                         int causal, float scale, const float* ln_u, const float* ln_b, int ln_C, float ln_eps,
                                         int finish_q, int finish_kv, int kv_live, int dtype, jen1_deep_phase* out) {
  JEN1_CHECK(q && k && v && out_t && out, "deep attention: null pointer");
  JEN1_CHECK(dtype == JEN1_F32 || dtype == JEN1_BF16 || dtype == JEN1_FP8, "deep attention: bad dtype");
  JEN1_CHECK(B >= 1 && H >= 1 && Nq >= 1 && Nk >= 1, "deep attention: bad sizes");
  JEN1_CHECK(d == 8 || d == 16 || d == 32 || d == 64 || d == 128, "deep attention: head dim %d must be 8, 16, 32, 64 or 128", d);
  JEN1_CHECK(Nk <= 192 && Nk * (d / 8) <= ((141 * 16 + JEN1_DEEP_THREADS - 1) / JEN1_DEEP_THREADS) * JEN1_DEEP_THREADS, "deep attention: Nk=%d d=%d outside the kernel's range", Nk, d);
  JEN1_CHECK(ldq % 8 == 0 && ldkv % 8 == 0 && ldo % 8 == 0 && q_off % 8 == 0 && k_off % 8 == 0 && v_off % 8 == 0 &&
             (!kv_extra || (ld_extra % 8 == 0 && kx_off % 8 == 0 && vx_off % 8 == 0)), "deep attention: offsets / strides must be multiples of 8 elements");
  const bool fin = finish_q || finish_kv;
  JEN1_CHECK(!fin || (ln_u && ln_b && ln_C >= 8 && ln_C % 8 == 0), "deep attention: a LayerNorm finish needs u, bias and ln_C");
  JEN1_CHECK(!(kv_live & 1) || Nk * (d / 8) <= JEN1_DEEP_ATTN_LIVE_VEC * JEN1_DEEP_THREADS, "deep attention: self-attention over Nk=%d d=%d is outside the unit's range", Nk, d);
  JEN1_CHECK(!finish_kv || (!kv_row && !kv_extra && Nk * (d / 8) <= JEN1_DEEP_THREADS && Nq == Nk && (kv_live & 1)),
             "deep attention: the K/V finish is for self-attention over at most %d vectors", JEN1_DEEP_THREADS);
  const int es = dtype == JEN1_F32 ? 4 : 2;      // (JEN1_FP8: sized like bf16 -- the output tile is bf16, the fp8 operands need less)
  const int DP = d < 32 ? 32 : d, DC = d < 16 ? 16 : d;
  const int NKP = (Nk + 31) & ~31, dq = DP + 8, vt = NKP + 8, sp = NKP + 4;
  const int st_rows = Nk > QCHUNK ? Nk : QCHUNK;
  auto lds_of = [&](bool vsep) {
    const int64_t kv_elems = vsep ? (((int64_t)NKP * dq + 15) & ~(int64_t)15) + (int64_t)DC * vt : ((NKP * dq > DC * vt) ? NKP * dq : DC * vt);
    return (int64_t)es * QCHUNK * dq + es * ((kv_elems + 15) & ~(int64_t)15) + 4 * (((int64_t)QCHUNK * sp + 3) & ~(int64_t)3) +
           es * (((int64_t)QCHUNK * vt + 7) & ~(int64_t)7) + 8 * st_rows + 64;
  };
  // V^T gets LDS of its own whenever it fits: one 512-thread workgroup of 256-register waves owns the CU either way, so LDS is free,
  // and a cross-attention then has K AND V^T staged before its dependency wait ends (restaging V^T in K's place behind the scores
  // was 1.2 - 2.0 us of every cross-attention phase with d >= 64)
  const bool vsep = lds_of(true) <= LDS_BUDGET;
  const int64_t lds = lds_of(vsep);
  JEN1_CHECK(lds <= LDS_BUDGET, "deep attention: Nk=%d d=%d needs %lld B of LDS", Nk, d, (long long)lds);
  jen1_deep_phase& p = *out;
  memset(&p, 0, sizeof(p));
  p.h.kind = JEN1_DEEP_ATTN;
  p.h.dtype = dtype;
  p.h.dep = -1;
  p.h.lds_bytes = align16i((int)lds);
  p.q = q; p.k = k; p.v = v; p.out = out_t; p.kv_row = kv_row; p.kv_extra = kv_extra; p.extra_row = extra_row; p.extra_step = extra_step;
  p.ln_u = ln_u; p.ln_b = ln_b;
  p.ld_extra = ld_extra; p.kx_off = kx_off; p.vx_off = vx_off; p.h.B = B; p.H = H; p.d = d; p.Nq = Nq; p.Nk = Nk; p.ldq = ldq; p.q_off = q_off;
  p.ldkv = ldkv; p.k_off = k_off; p.v_off = v_off; p.ldo = ldo; p.causal = causal;
  p.ln_C = ln_C; p.fin_q = finish_q; p.fin_kv = finish_kv; p.kv_live = kv_live;
  p.vsep = vsep ? 1 : 0;
  p.nqc = ceil_div(Nq, QCHUNK);
  int l2 = 0;
  while ((8 << l2) < d) ++l2;
  p.log2_vpr = l2;
  p.scale = scale; p.ln_eps = ln_eps; p.inv_H = 1.0f / (float)H; p.inv_nqc = 1.0f / (float)p.nqc;
  p.h.n_units = B * H * p.nqc;
  JEN1_CHECK(p.h.n_units < (1 << 20), "deep attention: too many units");
  return 0;
}

extern "C" int jen1_deep_phase_stats(const void* x, float* stats, int B, int L, int ld, int dtype, jen1_deep_phase* out) {
  JEN1_CHECK(x && stats && out && B >= 1 && L >= 1, "deep stats: bad arguments");
  JEN1_CHECK(dtype == JEN1_F32 || dtype == JEN1_BF16 || dtype == JEN1_FP8, "deep stats: bad dtype");
  JEN1_CHECK(ld >= 32 && (ld & (ld - 1)) == 0 && ld / 8 <= JEN1_DEEP_THREADS, "deep stats: row pitch %d must be a power of two in [32, 8192]", ld);
  jen1_deep_phase& p = *out;
  memset(&p, 0, sizeof(p));
  p.h.kind = JEN1_DEEP_STATS;
  p.h.dtype = dtype;
  p.h.dep = -1;
  p.sx = x; p.sstats = stats; p.sL = L; p.sld = ld;
  p.scpf = ld / JEN1_FINE_GROUPS;
  p.sgran = p.scpf >= 8 ? 8 : p.scpf;
  p.h.B = B;
  p.h.n_units = B;
  const int vpr = ld / 8, sub = 8 / p.sgran;
  p.h.lds_bytes = align16i((JEN1_DEEP_THREADS / vpr) * vpr * sub * 8);
  return 0;
}

extern "C" int jen1_deep_tile_count(int L_out, int tb, int M, int bm) {
  if (tb < 1 || bm < 1) return 0;
  return ceil_div(L_out, tb) * ceil_div(M, bm);
}

extern "C" int jen1_deep_phase_tile(const jen1_conv_args* a, int tb, int bm, const float* st0, int st0_tiles, int st0_nfg, const float* st1,
                                    int st1_tiles, int st1_nfg, int st_live, float* out_part, int out_nfg, jen1_deep_phase* out) {
  JEN1_CHECK(a && out, "deep tile: null pointer");
  JEN1_CHECK(a->dtype == JEN1_F32 || a->dtype == JEN1_BF16, "deep tile: dtype must be float32 or bf16 (the JEN1_FP8 mode runs the long levels in bf16)");
  JEN1_CHECK(a->x0 && a->w && a->y, "deep tile: null tensor");
  JEN1_CHECK(a->pro_mode == JEN1_PRO_NONE || a->pro_mode == JEN1_PRO_GN || a->pro_mode == JEN1_PRO_GN_SILU, "deep tile: prologue %d is not supported", a->pro_mode);
  JEN1_CHECK(!a->ln_fold && !a->row_scale && !a->out_rowstats && a->act == JEN1_ACT_NONE && a->m_split == 0, "deep tile: LayerNorm / row scale / activation / dual range are not tile options");
  JEN1_CHECK(a->c0 > 0 && a->c0 % 32 == 0 && a->c1 % 32 == 0 && a->M % 16 == 0, "deep tile: channels must be multiples of 32, M of 16");
  JEN1_CHECK(a->taps >= 1 && a->stride >= 1 && a->B >= 1 && a->L_in >= 1 && a->L_out >= 1, "deep tile: bad geometry");
  JEN1_CHECK(a->nseg >= 0 && a->nseg <= 2, "deep tile: at most two raw extra K segments");
  JEN1_CHECK(tb == 16 || tb == 32 || tb == 48 || tb == 64, "deep tile: %d positions per tile (16, 32, 48 or 64)", tb);
  JEN1_CHECK((bm == 128 || bm == 256) && a->M % bm == 0, "deep tile: %d output rows per unit (128 or 256, dividing M = %d)", bm, a->M);
  JEN1_CHECK(bm == 128 || tb <= 32, "deep tile: at 256 output rows per unit a tile has at most 32 positions (two fragments per wave)");
  const int es = a->dtype == JEN1_F32 ? 4 : 2;
  const bool gn = a->pro_mode != JEN1_PRO_NONE;
  jen1_deep_phase& p = *out;
  memset(&p, 0, sizeof(p));
  p.h.kind = JEN1_DEEP_TILE;
  p.h.dtype = a->dtype;
  p.h.dep = -1;
  p.h.mrep = 1;
  p.h.live_mask = a->live_mask;
  int ns = 0, coff = 0;
  p.h.src[ns++] = jen1_deep_src{a->x0, a->ld0, a->c0, 0, 1.0f};
  coff = a->c0;
  if (a->c1) {
    JEN1_CHECK(a->x1, "deep tile: c1 without x1");
    p.h.src[ns++] = jen1_deep_src{a->x1, a->ld1, a->c1, coff, a->src1_scale};
    coff += a->c1;
  }
  const int cmain = coff;
  JEN1_CHECK(cmain <= JEN1_DEEP_THREADS, "deep tile: %d main channels (at most %d: one affine pair per thread)", cmain, JEN1_DEEP_THREADS);
  int kx = 0;
  for (int s = 0; s < a->nseg; ++s) {
    const jen1_conv_seg& e = a->seg[s];
    JEN1_CHECK(e.x && e.kch > 0 && e.shift == 0 && e.ld >= 32 * e.kch, "deep tile: bad extra segment %d", s);
    p.h.src[ns++] = jen1_deep_src{e.x, e.ld, 32 * e.kch, coff, 1.0f};
    coff += 32 * e.kch;
    kx += e.kch;
  }
  for (int k = ns; k < JEN1_DEEP_MAX_SRC; ++k) p.h.src[k] = jen1_deep_src{a->x0, a->ld0, 0, 1 << 30, 1.0f};
  p.h.nsrc = ns; p.h.Ctot = coff; p.h.pitch = coff + 8;
  p.h.MT = a->M / 16;
  p.tl.cmain = cmain; p.tl.taps = a->taps; p.tl.pad_left = a->pad_left;
  p.tl.kch = cmain / 32;
  p.tl.KS = a->taps * p.tl.kch + kx;
  p.h.G = p.tl.KS;
  const int64_t wb = (int64_t)p.tl.KS * p.h.MT * 512 * es;
  JEN1_CHECK(wb < ((int64_t)1 << 31), "deep tile: packed weight too large for 31-bit offsets");
  p.h.w = a->w; p.h.w_bytes = (uint32_t)wb;
  p.h.B = a->B; p.h.L_in = a->L_in; p.h.L_out = a->L_out; p.h.stride = a->stride;
  p.h.pro_mode = a->pro_mode;
  if (gn) {
    JEN1_CHECK(a->gn_gamma && a->gn_beta && a->gn_groups >= 1 && a->gn_cpg >= 1 && a->gn_count >= 1 && st0, "deep tile: incomplete GroupNorm");
    p.h.norm_C = cmain;
    p.h.gn_groups = a->gn_groups;
    p.h.gn_cpg = a->gn_groups == 1 ? cmain : a->gn_cpg;
    JEN1_CHECK(a->gn_groups > 1 || a->c1 == 0, "deep tile: a single GroupNorm group over two sources is not supported");
    p.h.inv_count = 1.0f / (float)a->gn_count;
    p.h.gn_eps = a->gn_eps;
    if (a->film) {
      JEN1_CHECK(a->film_C == cmain && a->film_ld >= a->film_off + 2 * a->film_C, "deep tile: bad FiLM table geometry");
      p.h.p1 = a->film + a->film_off;
      p.h.p2 = a->film + a->film_off + a->film_C;
      p.h.p_ld = a->film_ld;
      p.h.film_row = a->film_row; p.h.film_step = a->film_step;
    } else {
      p.h.p1 = a->gn_gamma; p.h.p2 = a->gn_beta; p.h.p_ld = 0;
    }
    p.tl.st[0] = st0; p.tl.st_tiles[0] = st0_tiles; p.tl.st_nfg[0] = st0_tiles ? st0_nfg : 32;
    p.tl.st[1] = a->c1 ? st1 : st0; p.tl.st_tiles[1] = a->c1 ? st1_tiles : 0; p.tl.st_nfg[1] = (a->c1 && st1_tiles) ? st1_nfg : 32;
    JEN1_CHECK(!a->c1 || st1, "deep tile: the second normalised source needs statistics");
    p.tl.st_live = st_live & (a->c1 ? 3 : 1);
    for (int k = 0; k < (a->c1 ? 2 : 1); ++k) {
      const int nfg = p.tl.st_nfg[k], ck = k ? a->c1 : a->c0;
      JEN1_CHECK(nfg == 8 || nfg == 16 || nfg == 32, "deep tile: %d statistics groups per tile (8, 16 or 32)", nfg);
      JEN1_CHECK(ck % nfg == 0, "deep tile: %d channels do not split into %d statistics groups", ck, nfg);
      const int cps = ck / nfg;
      JEN1_CHECK(p.h.gn_cpg % cps == 0 || a->gn_groups == 1, "deep tile: GroupNorm groups of %d channels are not made of whole statistics groups of %d", p.h.gn_cpg, cps);
      p.tl.inv_cps[k] = 1.0f / (float)cps;
    }
    if (!a->c1) p.tl.inv_cps[1] = p.tl.inv_cps[0];
    p.tl.inv_cpg = 1.0f / (float)p.h.gn_cpg;
  }
  p.h.bias = a->bias; p.h.residual = a->residual; p.h.y = a->y;
  p.h.out_C = a->out_C; p.h.ps_f = a->ps_f < 1 ? 1 : a->ps_f; p.h.ps_off = a->ps_off; p.h.L_y = a->L_y; p.h.y_brows = a->y_brows;
  p.h.y_row0 = a->y_row0; p.h.ld_y = a->ld_y; p.h.ld_res = a->ld_res; p.h.act = a->act; p.h.y_f32 = a->y_f32;
  JEN1_CHECK(a->out_C % 16 == 0 && a->ld_y % 4 == 0 && (!a->residual || a->ld_res % 4 == 0) && a->out_C * p.h.ps_f == a->M,
             "deep tile: output channels must be a multiple of 16 (an M tile never straddles a sub-pixel phase), pitches of 4");
  JEN1_CHECK((int64_t)a->B * a->L_in * (a->ld0 > a->ld1 ? a->ld0 : a->ld1) < ((int64_t)1 << 31) && (int64_t)a->B * a->y_brows * a->ld_y < ((int64_t)1 << 31),
             "deep tile: tensor too large");
  p.tl.tb = tb;
  p.tl.tiles_t = ceil_div(a->L_out, tb);
  p.tl.BM = bm; p.tl.mblocks = a->M / bm; p.tl.MF = 2;
  p.tl.NF = tb / 16;
  p.tl.rows_in = (tb - 1) * a->stride + a->taps;
  p.tl.out_part = out_part;
  if (out_part) {
    JEN1_CHECK((out_nfg == 8 || out_nfg == 16 || out_nfg == 32) && a->out_C % out_nfg == 0 && (a->out_C / out_nfg) % 16 == 0 && !a->y_f32,
               "deep tile: %d output channels in %d statistics groups (groups of a multiple of 16 channels)", a->out_C, out_nfg);
    p.tl.out_nfg = out_nfg;
    p.tl.out_cps = a->out_C / out_nfg;
    p.tl.inv_out_cps = 1.0f / (float)p.tl.out_cps;
  }
  const int tile_b = align16i(p.tl.rows_in * p.h.pitch * es);
  p.tl.tab_off = tile_b;
  p.tl.red_off = tile_b + align16i(8 * cmain);
  const int tot = p.tl.red_off + (2 * (JEN1_DEEP_THREADS / 64) * 64 + 128) * 4;
  JEN1_CHECK(tot <= LDS_BUDGET, "deep tile: %d B of LDS", tot);
  p.h.lds_bytes = tot;
  p.h.n_units = p.tl.mblocks * a->B * p.tl.tiles_t;
  JEN1_CHECK(p.h.n_units < (1 << 20), "deep tile: too many units");
  p.tl.inv_tiles_t = 1.0f / (float)p.tl.tiles_t;
  p.tl.inv_bt = 1.0f / (float)(a->B * p.tl.tiles_t);
  p.h.inv_vpr = 1.0f / (float)(coff / 8);
  p.h.inv_Lin = 1.0f / (float)a->L_in;
  p.h.inv_Lout = 1.0f / (float)a->L_out;
  return 0;
}

extern "C" int jen1_deep_link(jen1_deep_phase* phases, int n_phases, int nwg, void* blobs, void* headers) {
  JEN1_CHECK(phases && blobs && headers && n_phases >= 1 && nwg >= 1, "deep link: bad arguments");
  JEN1_CHECK(n_phases <= JEN1_DEEP_MAX_PHASES, "deep link: %d phases (at most %d)", n_phases, JEN1_DEEP_MAX_PHASES);
  int lds = 0, rot = 0, units_before = 0;
  unsigned char* bl = reinterpret_cast<unsigned char*>(blobs);
  int32_t* hd = reinterpret_cast<int32_t*>(headers);
  memset(bl, 0, (size_t)n_phases * BLOB);
  for (int p = 0; p < n_phases; ++p) {
    jen1_deep_phase& P = phases[p];
    if (P.h.kind == JEN1_DEEP_GEMM && P.h.mrep == 1 && !getenv("JEN1_DEEP_NO_MREP")) {
      // more units than workgroups: a unit finishes several M tiles from one staged tile instead of the workgroup staging the
      // same rows once per tile (second K-reduction scratch behind the first)
      int mrep = 1;
      const char* cap_s = getenv("JEN1_DEEP_UNIT_CAP");          // tuning: at most this many units per phase (default: one per workgroup)
      int cap = cap_s ? atoi(cap_s) : nwg;
      // a phase in front of a cross-attention keeps half of the workgroups free: an attention unit stages its cached text K / V^T
      // (up to 129 x 128 x 2 elements) BEFORE its dependency wait, ~4.7 us that are hidden only on a workgroup that sat out the phase before
      // (tuning knob, off: measured 857 -> 875 us per launch at a cap of 128 or 192 -- the second M tile of a unit costs more than the
      // exposed staging)
      static const int cap_attn = getenv("JEN1_DEEP_CAP_BEFORE_ATTN") ? atoi(getenv("JEN1_DEEP_CAP_BEFORE_ATTN")) : 0;
      if (cap_attn > 0 && p + 1 < n_phases && phases[p + 1].h.kind == JEN1_DEEP_ATTN && cap_attn < cap) cap = cap_attn;
      const int nch = P.h.n_chunks < 1 ? 1 : P.h.n_chunks;
      while (mrep < 8 && (P.h.MT / mrep) * P.h.groups_n * nch > cap && P.h.MT % (2 * mrep) == 0 && P.h.mt_split % (2 * mrep) == 0) mrep *= 2;
      if (mrep > 1 && P.h.lds_bytes + P.h.red_bytes <= LDS_BUDGET) {
        P.h.mrep = mrep;
        P.h.n_units = (P.h.MT / mrep) * P.h.groups_n * nch;
        P.h.lds_bytes += P.h.red_bytes;
      }
    }
    P.h.dep = p - 1;
    P.h.dep_units = p ? phases[p - 1].h.n_units : 0;
    // successive phases start their units on successive workgroups (multiples of 8 keep a unit's XCD = its M tile mod 8):
    // a workgroup that just finished a unit is rarely the one the next phase waits for, so it has the time of a few
    // phases to pull the weight slice of its next unit, and the weight streams spread over all CUs
    P.h.rot = rot % nwg;
    rot += ((P.h.n_units + 7) / 8) * 8;
    lds = P.h.lds_bytes > lds ? P.h.lds_bytes : lds;
    unsigned char* b = bl + (size_t)p * BLOB;
    memcpy(b, &P, sizeof(P));
    hd[4 * p + 0] = P.h.n_units; hd[4 * p + 1] = P.h.rot; hd[4 * p + 2] = P.h.kind; hd[4 * p + 3] = units_before;
    units_before += P.h.n_units;
    if (P.h.kind != JEN1_DEEP_GEMM) continue;
    // K chunks that can touch a real input row (a segment whose every row is conv padding for every position is skipped
    // together with its weights: exact), dealt round-robin to the waves; per wave and segment that is one run
    int16_t* cnt = reinterpret_cast<int16_t*>(b + TAB_OFF);
    int32_t* runs = reinterpret_cast<int32_t*>(b + TAB_OFF + 64);
    const int tmax = (P.h.L_out - 1) * P.h.stride;
    int k0 = 0;
    for (int s = 0; s < P.h.nseg; ++s) {
      const int sb = s ? P.seg[s - 1].gend : 0, se = P.seg[s].gend, sh = P.seg[s].shift;
      if (!(tmax + sh >= 0 && sh < P.h.L_in)) continue;
      const bool low = P.h.mt_split && sb < P.h.g_split;
      JEN1_CHECK(!low || se <= P.h.g_split, "deep link: phase %d: the dual-range split must fall on a segment boundary", p);
      const int Ls = se - sb;
      for (int w = 0; w < NW; ++w) {
        const int i0 = ((w - k0) % NW + NW) % NW;
        if (i0 >= Ls) continue;
        const int n = (Ls - i0 + NW - 1) / NW;
        const int r = cnt[2 * NW + w];
        JEN1_CHECK(r < MAXRUN, "deep link: phase %d has more than %d runs per wave", p, MAXRUN);
        int32_t* e = runs + ((size_t)w * MAXRUN + r) * 4;
        e[0] = sb + i0; e[1] = n; e[2] = P.seg[s].coff + i0 * 32; e[3] = sh;
        cnt[w] = (int16_t)(cnt[w] + n);
        cnt[2 * NW + w] = (int16_t)(r + 1);
        if (low) { cnt[NW + w] = cnt[w]; cnt[3 * NW + w] = cnt[2 * NW + w]; }
      }
      k0 += Ls;
    }
    if (!P.h.mt_split) for (int w = 0; w < NW; ++w) { cnt[NW + w] = cnt[w]; cnt[3 * NW + w] = cnt[2 * NW + w]; }
    // the first ring round of every wave, tabulated (chunk index, staged offset), and the cursor behind it
    const int pf = P.h.dtype == JEN1_F32 ? JEN1_DEEP_PF_F : JEN1_DEEP_PF_B;      // (JEN1_FP8: DeepCfg<fp8_t>::PF = PF_B)
    int32_t* slots = reinterpret_cast<int32_t*>(b + SLOT_OFF);
    int32_t* curs = reinterpret_cast<int32_t*>(b + CUR_OFF);
    for (int w = 0; w < NW; ++w) {
      const int32_t* rw = runs + (size_t)w * MAXRUN * 4;
      const int nr = cnt[2 * NW + w], tot = cnt[w];
      int r = 0, left = nr ? rw[1] : 0, g = nr ? rw[0] : 0, col = nr ? rw[2] : 0, sh = nr ? rw[3] : 0;
      auto next = [&]() {
        if (--left > 0) { g += NW; col += NW * 32; }
        else if (r + 1 < nr) { ++r; left = rw[r * 4 + 1]; g = rw[r * 4]; col = rw[r * 4 + 2]; sh = rw[r * 4 + 3]; }
      };
      for (int i = 0; i < SLOTS; ++i) {
        slots[w * 2 * SLOTS + i] = g;
        slots[w * 2 * SLOTS + SLOTS + i] = sh * P.h.pitch + col;
        if (i + 1 == pf) {                 // the device cursor state after the ring's first round (kc_next semantics)
          int r2 = r, l2 = left, g2 = g, c2 = col, s2 = sh;
          if (i + 1 < tot) {
            if (--l2 > 0) { g2 += NW; c2 += NW * 32; }
            else if (r2 + 1 < nr) { ++r2; l2 = rw[r2 * 4 + 1]; g2 = rw[r2 * 4]; c2 = rw[r2 * 4 + 2]; s2 = rw[r2 * 4 + 3]; }
          }
          curs[w * 8 + 0] = r2; curs[w * 8 + 1] = l2; curs[w * 8 + 2] = g2; curs[w * 8 + 3] = c2; curs[w * 8 + 4] = s2;
        }
        if (i + 1 < tot) next();           // slots beyond the last chunk repeat its offset (zero weights)
      }
    }
  }
  return lds + WS_OFF;
}

// ---- the sentinel every in-launch tensor starts from (see "synchronisation") ------------------------------------------------------
namespace {
struct PoisonEntry {
  unsigned long long ptr, bytes;          // bytes: a multiple of 16
};
__global__ __launch_bounds__(256) void poison_kernel(const PoisonEntry* __restrict__ tab, unsigned* __restrict__ sync, int n_tab, void* zero_ptr,
                                                      unsigned long long zero_bytes) {
  if ((int)blockIdx.y >= n_tab) {
    // rows behind the table: the caller's per-step scratch (statistics arena) is zeroed by the same launch, the rows share it
    uint4* z = reinterpret_cast<uint4*>(zero_ptr);
    const size_t n = zero_bytes >> 4;
    const size_t nblk = (size_t)gridDim.x * (gridDim.y - n_tab), blk = (size_t)(blockIdx.y - n_tab) * gridDim.x + blockIdx.x;
    for (size_t i = blk * 256 + threadIdx.x; i < n; i += nblk * 256) z[i] = make_uint4(0u, 0u, 0u, 0u);
    return;
  }
  const PoisonEntry e = tab[blockIdx.y];
  uint4* p = reinterpret_cast<uint4*>(e.ptr);
  const size_t n = e.bytes >> 4;
  const uint4 ones = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = ones;
  // the ticket counter of the launch (it may lie inside the zeroed scratch: zero either way)
  if (sync && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) sync[0] = 0u;
}
}  // namespace

extern "C" int jen1_deep_poison(const void* table_dev, int n, uint32_t* sync, void* stream) {
  JEN1_CHECK(table_dev && n >= 1 && n <= 65535, "deep poison: bad table");
  hipLaunchKernelGGL(poison_kernel, dim3(8, n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const PoisonEntry*>(table_dev), sync,
                     n, (void*)nullptr, 0ull);
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_deep_poison_zero(const void* table_dev, int n, uint32_t* sync, void* zero_ptr, int64_t zero_bytes, void* stream) {
  JEN1_CHECK(table_dev && n >= 1 && n <= 60000, "deep poison: bad table");
  JEN1_CHECK(zero_ptr && zero_bytes > 0 && (zero_bytes & 15) == 0 && ((uintptr_t)zero_ptr & 15) == 0, "deep poison: the zeroed area must be 16-byte aligned and sized");
  // enough rows for the zeroing to keep up with the poisoning of the biggest tensors (8 blocks per row)
  int zrows = (int)((zero_bytes + (1 << 18) - 1) >> 18);
  zrows = zrows < 1 ? 1 : (zrows > 64 ? 64 : zrows);
  hipLaunchKernelGGL(poison_kernel, dim3(8, n + zrows), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const PoisonEntry*>(table_dev), sync,
                     n, zero_ptr, (unsigned long long)zero_bytes);
  JEN1_HIP(hipGetLastError());
  return 0;
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

extern "C" int jen1_deep_run_kinds(const void* blobs_dev, const void* headers_dev, int n_phases, uint32_t* sync, uint32_t* err, int nwg,
                                   int lds_bytes, int dtype, int tickets, int kind_mask, void* stream);
extern "C" int jen1_deep_run_mode(const void* blobs_dev, const void* headers_dev, int n_phases, uint32_t* sync, uint32_t* err, int nwg,
                                  int lds_bytes, int dtype, int tickets, void* stream) {
  return jen1_deep_run_kinds(blobs_dev, headers_dev, n_phases, sync, err, nwg, lds_bytes, dtype, tickets, KM_DEEP, stream);
}
extern "C" int jen1_deep_run(const void* blobs_dev, const void* headers_dev, int n_phases, uint32_t* sync, int nwg, int lds_bytes, int dtype,
                             void* stream) {
  JEN1_CHECK(sync, "deep run: bad arguments");
  return jen1_deep_run_mode(blobs_dev, headers_dev, n_phases, sync, sync + jen1_deep_error_word(n_phases), nwg, lds_bytes, dtype, 1, stream);
}
extern "C" int jen1_deep_run_err(const void* blobs_dev, const void* headers_dev, int n_phases, uint32_t* sync, uint32_t* err, int nwg,
                                 int lds_bytes, int dtype, void* stream) {
  return jen1_deep_run_mode(blobs_dev, headers_dev, n_phases, sync, err, nwg, lds_bytes, dtype, 1, stream);
}

namespace {
template <typename T, bool TK, int KM>
int launch_deep(const unsigned char* bl, const int4* hd, int n_phases, uint32_t* sync, uint32_t* err, int nwg, int lds_bytes, hipStream_t s) {
  auto kern = deep_kernel<T, TK, KM>;
  JEN1_MAX_LDS_ONCE(kern, LDS_TOTAL);
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(NT), (size_t)lds_bytes, s, bl, hd, n_phases, sync, err);
  JEN1_HIP(hipGetLastError());
  return 0;
}
}  // namespace

extern "C" int jen1_deep_run_kinds(const void* blobs_dev, const void* headers_dev, int n_phases, uint32_t* sync, uint32_t* err, int nwg,
                                   int lds_bytes, int dtype, int tickets, int kind_mask, void* stream) {
  JEN1_CHECK(blobs_dev && headers_dev && sync && err && n_phases >= 1 && n_phases <= JEN1_DEEP_MAX_PHASES && nwg >= 1, "deep run: bad arguments");
  JEN1_CHECK(lds_bytes >= WS_OFF && lds_bytes <= LDS_TOTAL, "deep run: %d B of LDS", lds_bytes);
  JEN1_CHECK(dtype == JEN1_F32 || dtype == JEN1_BF16 || dtype == JEN1_FP8, "deep run: bad dtype");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const unsigned char* bl = reinterpret_cast<const unsigned char*>(blobs_dev);
  const int4* hd = reinterpret_cast<const int4*>(headers_dev);
  if ((kind_mask & ~KM_TILE) == 0) {
    // tile phases (+ statistics) only; the JEN1_FP8 mode runs them in bf16
    const bool f32 = dtype == JEN1_F32;
    if (tickets) return f32 ? launch_deep<float, true, KM_TILE>(bl, hd, n_phases, sync, err, nwg, lds_bytes, s)
                            : launch_deep<bf16_t, true, KM_TILE>(bl, hd, n_phases, sync, err, nwg, lds_bytes, s);
    return f32 ? launch_deep<float, false, KM_TILE>(bl, hd, n_phases, sync, err, nwg, lds_bytes, s)
               : launch_deep<bf16_t, false, KM_TILE>(bl, hd, n_phases, sync, err, nwg, lds_bytes, s);
  }
  JEN1_CHECK((kind_mask & ~KM_DEEP) == 0, "deep run: a program mixes tile phases with GEMM / attention phases (kinds 0x%x): record them as separate programs", kind_mask);
  if (tickets) {
    if (dtype == JEN1_F32) return launch_deep<float, true, KM_DEEP>(bl, hd, n_phases, sync, err, nwg, lds_bytes, s);
    if (dtype == JEN1_FP8) return launch_deep<fp8_t, true, KM_DEEP>(bl, hd, n_phases, sync, err, nwg, lds_bytes, s);
    return launch_deep<bf16_t, true, KM_DEEP>(bl, hd, n_phases, sync, err, nwg, lds_bytes, s);
  }
  if (dtype == JEN1_F32) return launch_deep<float, false, KM_DEEP>(bl, hd, n_phases, sync, err, nwg, lds_bytes, s);
  if (dtype == JEN1_FP8) return launch_deep<fp8_t, false, KM_DEEP>(bl, hd, n_phases, sync, err, nwg, lds_bytes, s);
  return launch_deep<bf16_t, false, KM_DEEP>(bl, hd, n_phases, sync, err, nwg, lds_bytes, s);
}
