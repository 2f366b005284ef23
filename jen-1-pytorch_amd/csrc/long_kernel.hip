// Sample-resident long-level kernel of the JEN-1 denoiser (gfx950 / MI355X).  C ABI: include/jen1_long.h.
//
// to_in, levels 0..2 down and levels 2..0 + to_out up (reference jen1/model/model.py:243-262, blocks jen1/model/blocks.py:98-145
// ConvBlock1d, :168-231 ResnetBlock1d, :540-650, :653-764) as TWO launches of 256 resident workgroups x 512 threads that walk a
// list of phases (one convolution each).  Sample b belongs to workgroups b, b + B, ... (with B = 8: the 32 CUs of XCD b); workgroup
// k of the group computes M block k % mblocks x position tile k / mblocks of EVERY phase.  Per unit:
//   (0) [behind the previous unit's stores] the unit's weight slice (8 waves x one 16-row M tile x up to RING k-steps) is requested
//       into registers, then gamma / beta / FiLM, bias;
//   (1) ONE polled round: the statistics partials of the sample (wave w: slot w of every producing unit) and the unit's input
//       window (tile + conv halo), 8-byte words that start the step poisoned -- reading them complete IS the dependency wait;
//   (2) partials -> group sums (fixed cross-lane tree) -> affine pair y = silu(A x + S) per channel in LDS;
//   (3) the window is normalised from registers into the LDS tile (zero rows for the conv padding: a tap is a row offset);
//   (4) MFMA loop: wave w owns GEMM rows 16 w .. + 16 of the M block for all NF position fragments; weights from the register
//       ring (refilled for slices longer than the ring), activations as row-shifted views of the tile;
//   (5) epilogue: bias, residual, sub-pixel row map of ConvTranspose1d, write-through stores, ONE (sum, sumsq) word per wave.
// Inter-workgroup visibility as in deep_kernel.hip (cdna_hip_programming.md Guideline 16 R1): relaxed agent-scope 8-byte stores /
// loads (write-through, L1-bypassing); every spin is bounded, a time-out raises the error word and releases every waiter.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "jen1_long.h"

namespace {

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int NT = JEN1_LONG_THREADS;
constexpr int NW = NT / 64;
constexpr unsigned OOB = 0x80000000u;
constexpr int RSRC_FLAGS = 0x00020000;
constexpr u64 POISON = ~0ull;
constexpr int LDS_TOTAL = 160 * 1024;
constexpr int LDS_MIN = 84 * 1024;          // more than half of a CU's LDS: never two workgroups of this kernel on one CU
static_assert(sizeof(jen1_long_phase) == JEN1_LONG_DESC_BYTES, "the descriptor is read as two dwords per lane");

__device__ __forceinline__ gu64* g64(const void* p) { return (gu64*)(u64)p; }
__device__ __forceinline__ gu32* g32(const void* p) { return (gu32*)(u64)p; }
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---- the descriptor without LDS round trips: lane i of two registers holds dwords i and 64 + i; a field is a v_readlane ------------
struct DescRegs {
  unsigned d0, d1;
};
__device__ __forceinline__ DescRegs load_desc(const unsigned char* descs, int p, int lane) {
  const unsigned* q = reinterpret_cast<const unsigned*>(descs + (size_t)p * JEN1_LONG_DESC_BYTES);
  DescRegs r;
  r.d0 = q[lane];
  r.d1 = q[64 + lane];
  return r;
}
template <int OFF>
__device__ __forceinline__ int d_i32(const DescRegs& r) {
  constexpr int dw = OFF / 4;
  return __builtin_amdgcn_readlane((int)(dw < 64 ? r.d0 : r.d1), dw & 63);
}
template <int OFF>
__device__ __forceinline__ float d_f32(const DescRegs& r) { return __builtin_bit_cast(float, d_i32<OFF>(r)); }
template <int OFF, typename PT>
__device__ __forceinline__ PT d_ptr(const DescRegs& r) {
  const unsigned lo = (unsigned)d_i32<OFF>(r), hi = (unsigned)d_i32<OFF + 4>(r);
  return reinterpret_cast<PT>(((u64)hi << 32) | lo);
}
#define LI(f) d_i32<offsetof(jen1_long_phase, f)>(dr)
#define LF(f) d_f32<offsetof(jen1_long_phase, f)>(dr)
#define LP(f, type) d_ptr<offsetof(jen1_long_phase, f), type>(dr)

// ---- 8-element vectors through agent-scope accesses --------------------------------------------------------------------------
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
__device__ __forceinline__ bool raw_bad(const Raw8<bf16_t>& r) { return (r.d[0] == POISON) | (r.d[1] == POISON); }
__device__ __forceinline__ bool raw_bad(const Raw8<float>& r) {
  return (r.d[0] == POISON) | (r.d[1] == POISON) | (r.d[2] == POISON) | (r.d[3] == POISON);
}
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
// The all-ones 8-byte word is reserved ("not stored yet", include/jen1_deep.h): every live store breaks exactly that pattern
// LOC: every reader of the word runs on the XCD of the writer (a sample's group of workgroups on one XCD): a PLAIN store -- the line
// stays in that XCD's L2, where the readers' L1-bypassing polls find it (0.30 us hand-off against 0.47 - 0.60 written through,
// tools/xcd_handoff_probe.hip; a plain store never arrives on ANOTHER XCD before the kernel ends)
template <bool LOC, typename G>
__device__ __forceinline__ void st_word(G* p, int i, unsigned lo, unsigned hi) {
  lo -= ((lo & hi) == 0xffffffffu) ? 1u : 0u;
  if constexpr (LOC) *(g64(p) + i) = ((u64)hi << 32) | lo;
  else __hip_atomic_store(g64(p) + i, ((u64)hi << 32) | lo, RLX_AGENT);
}
template <bool LOC>
__device__ __forceinline__ void st_live4(bf16_t* p, const float (&v)[4]) {
  bf16x4 a;
#pragma unroll
  for (int i = 0; i < 4; ++i) a[i] = (bf16_t)v[i];
  const u32x2 w = __builtin_bit_cast(u32x2, a);
  st_word<LOC>(p, 0, w[0], w[1]);
}
template <bool LOC>
__device__ __forceinline__ void st_live4(float* p, const float (&v)[4]) {
  st_word<LOC>(p, 0, __float_as_uint(v[0]), __float_as_uint(v[1]));
  st_word<LOC>(p, 1, __float_as_uint(v[2]), __float_as_uint(v[3]));
}
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

// ---- MFMA fragments ---------------------------------------------------------------------------------------------------------------
template <typename T> struct LFrag;
template <> struct LFrag<bf16_t> { typedef bf16x8 type; };
template <> struct LFrag<float> { typedef f32x8 type; };
__device__ __forceinline__ void lmma(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void lmma(f32x4& acc, const f32x8& a, const f32x8& b) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], acc, 0, 0, 0);
}
__device__ __forceinline__ void llds(bf16x8& f, const bf16_t* p) { f = *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ void llds(f32x8& f, const float* p) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
  f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
}
// weight fragment through a buffer descriptor (out-of-range offsets return 0 and move no bytes)
__device__ __forceinline__ void wload(bf16x8& f, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
#ifdef JEN1_LONG_EXP_NOW         // timing experiment only: every weight request is out of range (no bytes move)
  voff = OOB;
#endif
  f = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void wload(f32x8& f, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(r, voff + 16u, soff, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f.v[j] = __uint_as_float(lo[j]);
    f.v[4 + j] = __uint_as_float(hi[j]);
  }
}
template <int CTRL>
__device__ __forceinline__ float ldpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum_l(float v) {
  v += ldpp<0xB1>(v);
  v += ldpp<0x4E>(v);
  v += ldpp<0x141>(v);
  v += ldpp<0x140>(v);
  return v;
}
__device__ __forceinline__ float rlane(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

template <typename T> struct LongCfg;
#ifndef JEN1_LONG_RING_B
#define JEN1_LONG_RING_B 24
#endif
#ifndef JEN1_LONG_VB_B
#define JEN1_LONG_VB_B 5
#endif
#ifndef JEN1_LONG_RING_F
#define JEN1_LONG_RING_F 12
#endif
#ifndef JEN1_LONG_VB_F
#define JEN1_LONG_VB_F 3
#endif
template <> struct LongCfg<bf16_t> { static constexpr int RING = JEN1_LONG_RING_B, MAXVB = JEN1_LONG_VB_B; };     // 96 + 36 registers
template <> struct LongCfg<float> { static constexpr int RING = JEN1_LONG_RING_F, MAXVB = JEN1_LONG_VB_F; };      // 96 + 30 registers
constexpr int SPL = 2;                         // statistics words per lane and source: up to 128 producing units per sample

#ifndef JEN1_LONG_POLL_LIMIT
#define JEN1_LONG_POLL_LIMIT (1u << 17)
#endif
#ifndef JEN1_LONG_POLL_SLEEP
#define JEN1_LONG_POLL_SLEEP 1
#endif
struct LSync {
  unsigned* err;
  bool dead;
  int p;
#ifdef JEN1_LONG_PROFILE
  unsigned long long tt[8];
#endif
};
#ifdef JEN1_LONG_PROFILE
__device__ unsigned long long* g_long_dbg = nullptr;
#define LK_STAMP(sy, i) do { (sy).tt[(i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define LK_STAMP(sy, i) do { } while (0)
#endif
// behind a round of loads of one wave: `bad` = this lane saw a sentinel word.  True when the wave has to load again.
__device__ __forceinline__ bool poll_again(LSync& sy, bool bad, unsigned& spins) {
#ifdef JEN1_LONG_EXP_NOWAIT      // timing experiment only (results are garbage): no dependency waits at all
  return false;
#endif
  if (!__builtin_amdgcn_ballot_w64(bad) || sy.dead) return false;
  ++spins;
  if ((spins & 63u) == 0u) {
    const unsigned ev = __hip_atomic_load(g32(sy.err), RLX_AGENT);
    if (rfl((int)ev) != 0) { sy.dead = true; return false; }
  }
  if (spins > JEN1_LONG_POLL_LIMIT) {
    if ((threadIdx.x & 63) == 0) __hip_atomic_store(g32(sy.err), (unsigned)(sy.p + 1), RLX_AGENT);
    sy.dead = true;
    return false;
  }
#if JEN1_LONG_POLL_SLEEP > 0
  __builtin_amdgcn_s_sleep(JEN1_LONG_POLL_SLEEP);
#endif
  return true;
}

// the staging vector -> its global address; which of the four sources it comes from by explicit selects
struct SrcTab {
  u64 x0, x1, x2, x3;
  int ld0, ld1, ld2, ld3, e0, e1, e2, live, vpr, nvec, tin0, L_in, b;
  float inv_vpr;
};
template <typename T>
__device__ __forceinline__ const T* src_vec_ptr(const SrcTab t, int v, bool& lv) {
  const int vv = v < t.nvec ? v : 0;
  const int row = (int)(((float)vv + 0.5f) * t.inv_vpr);
  const int c = (vv - row * t.vpr) * 8;
  const int tin = t.tin0 + row;
  const bool ok = tin >= 0 && tin < t.L_in;
  const bool k1 = c >= t.e0, k2 = c >= t.e1, k3 = c >= t.e2;
  const u64 xp = k3 ? t.x3 : (k2 ? t.x2 : (k1 ? t.x1 : t.x0));
  const int ld = k3 ? t.ld3 : (k2 ? t.ld2 : (k1 ? t.ld1 : t.ld0));
  const int coff = k3 ? t.e2 : (k2 ? t.e1 : (k1 ? t.e0 : 0));
  const int kk = k3 ? 3 : (k2 ? 2 : (k1 ? 1 : 0));
  lv = ok && v < t.nvec && ((t.live >> kk) & 1);
  return reinterpret_cast<const T*>(xp) + ((unsigned)(t.b * t.L_in + (ok ? tin : 0)) * (unsigned)ld + (unsigned)(c - coff));
}

// ---- (0) the unit's weight slice: wave wv owns M tile mblk * 8 + wv; the first RING k-steps -------------------------------------------
template <typename T>
__device__ __forceinline__ void ring_fill(const DescRegs& dr, int slot, int lane, int wv, typename LFrag<T>::type (&ringA)[LongCfg<T>::RING / 2],
                                          typename LFrag<T>::type (&ringB)[LongCfg<T>::RING / 2]) {
  constexpr int HR = LongCfg<T>::RING / 2;
  constexpr unsigned ES = sizeof(T), BLK = 512 * ES;
  const int MT = LI(MT), KS = LI(KS);
  const int mblk = slot & (LI(mblocks) - 1);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(LP(w, const void*)), 0, LI(w_bytes), RSRC_FLAGS);
  const unsigned voff = (unsigned)(mblk * 8 + wv) * BLK + (unsigned)lane * (8u * ES);
  const unsigned step = (unsigned)MT * BLK;
#pragma unroll
  for (int s = 0; s < HR; ++s) wload(ringA[s], rw, s < KS ? voff : OOB, s < KS ? (unsigned)s * step : 0u);
  if (KS > HR) {                 // (a 128-channel k = 3 slice is 12 k-steps: the second half of the ring stays untouched)
#pragma unroll
    for (int s = 0; s < HR; ++s) wload(ringB[s], rw, HR + s < KS ? voff : OOB, HR + s < KS ? (unsigned)(HR + s) * step : 0u);
  }
}

// ======================================================================================================================================
// one unit: sample b, slot (= entry index of its partials) of phase dr
// ======================================================================================================================================
template <typename T, bool LOC>
__device__ __forceinline__ void long_unit(const DescRegs& dr, int b, int slot, LSync& sy, typename LFrag<T>::type (&ringA)[LongCfg<T>::RING / 2],
                                          typename LFrag<T>::type (&ringB)[LongCfg<T>::RING / 2], int tid) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typedef typename LFrag<T>::type Frag;
  constexpr bool PRECISE = is_f32<T>::value;
  constexpr int RING = LongCfg<T>::RING, MAXVB = LongCfg<T>::MAXVB;
  constexpr unsigned ES = sizeof(T), BLK = 512 * ES;
  const int lane = tid & 63;
  const int wv = rfl(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  LK_STAMP(sy, 0);
  // ---- geometry (all scalar) -------------------------------------------------------------------------------------------------------
  const int tb = LI(tb);
  const int mblk = slot & (LI(mblocks) - 1), tt = slot >> LI(mb_shift);
  const int t0 = tt * tb;
  const int L_in = LI(L_in), L_out = LI(L_out), stride = LI(stride), taps = LI(taps), pad_left = LI(pad_left);
  const int cmain = LI(cmain), call = LI(call), pitch = LI(pitch), kch = LI(kch), KS = LI(KS), MT = LI(MT), NF = LI(NF);
  const int rows_in = LI(rows_in);
  const int tin0 = t0 * stride - pad_left;
  const int c0 = LI(src[0].C), c2 = LI(src[2].C);
  const int live = LI(live_mask);
  const int pro_mode = LI(pro_mode);
  const bool gn = pro_mode == JEN1_PRO_GN || pro_mode == JEN1_PRO_GN_SILU;
  const bool do_silu = pro_mode == JEN1_PRO_GN_SILU;
  T* tile = reinterpret_cast<T*>(smem);
  float* tabA = reinterpret_cast<float*>(smem + LI(tab_off));
  float* tabS = tabA + cmain;
  float2* stl = reinterpret_cast<float2*>(smem + LI(st_off));        // [2 sources][32 entries] (sum, sumsq)

  // ---- (1) addresses of the staging vectors + the statistics words; the polled round goes out FIRST, everything else of the set-up
  // (parameters of the prologue, bias, output rows, the residual's addresses) is computed while it is in flight ------------------------------
  const int vpr = call >> 3;
  const float inv_vpr = LF(inv_vpr);
  const int nvec = rows_in * vpr;
  const void* sx0 = LP(src[0].x, const void*); const int sld0 = LI(src[0].ld);
  const void* sx1 = LP(src[1].x, const void*); const int sld1 = LI(src[1].ld);
  const void* sx2 = LP(src[2].x, const void*); const int sld2 = LI(src[2].ld);
  const void* sx3 = LP(src[3].x, const void*); const int sld3 = LI(src[3].ld);
  // vector v of the tile: row v / vpr, 8 channels from (v % vpr) * 8  (a free function with by-value arguments: as a lambda that captures
  // the four sources by reference the selects below become selects of ADDRESSES of the captured scalars, which then live in scratch)
  const SrcTab stab = {(u64)sx0, (u64)sx1, (u64)sx2, (u64)sx3, sld0, sld1, sld2, sld3, c0, cmain, cmain + c2, live, vpr, nvec, tin0, L_in, b, inv_vpr};
  auto vec_ptr = [&](int v, bool& lv) __attribute__((always_inline)) -> const T* { return src_vec_ptr<T>(stab, v, lv); };
  struct Batch {
    Raw8<T> x[MAXVB];
    const T* gp[MAXVB];
    bool lv[MAXVB];
  };
  auto prep_batch = [&](Batch& bt, int v0) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < MAXVB; ++k) {
      if (v0 + k * NT < nvec) bt.gp[k] = vec_ptr(v0 + k * NT + tid, bt.lv[k]);
    }
  };
  auto issue_batch = [&](Batch& bt, int v0) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < MAXVB; ++k) {
      if (v0 + k * NT < nvec) {
        if (bt.lv[k]) ld_live(bt.x[k], bt.gp[k]);
        else ld_plain(bt.x[k], bt.gp[k]);
      }
    }
  };
  auto batch_bad = [&](const Batch& bt, int v0) __attribute__((always_inline)) -> bool {
    bool bad = false;
#pragma unroll
    for (int k = 0; k < MAXVB; ++k) {
      if (v0 + k * NT < nvec) bad |= bt.lv[k] && raw_bad(bt.x[k]);
    }
    return bad;
  };
  Batch cur;
  prep_batch(cur, 0);
  // statistics: wave wv reads slot wv of every entry of the sample (partial form), wave 0 the 32 fine-group totals (totals form)
  const int ne0 = gn ? LI(st_entries[0]) : 0, ne1 = (gn && cmain > c0) ? LI(st_entries[1]) : 0;
  const bool tot0 = gn && ne0 == 0, tot1 = gn && cmain > c0 && ne1 == 0;
  const gu64* q0 = g64(LP(st[0], const float*)) + (ne0 ? (size_t)(b * 8 + wv) * ne0 : (size_t)b * 32);
  const gu64* q1 = g64(LP(st[1], const float*)) + (ne1 ? (size_t)(b * 8 + wv) * ne1 : (size_t)b * 32);
  const bool stlive0 = (live >> 4) & 1, stlive1 = (live >> 5) & 1;
  u64 w0[SPL], w1[SPL];
  auto issue_stats = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
      const int e = lane + k * 64;
      if (k * 64 < ne0) w0[k] = e < ne0 ? __hip_atomic_load(q0 + e, RLX_AGENT) : 0ull;
      if (k * 64 < ne1) w1[k] = e < ne1 ? __hip_atomic_load(q1 + e, RLX_AGENT) : 0ull;
    }
  };
  auto stats_bad = [&]() __attribute__((always_inline)) -> bool {
    bool bad = false;
#pragma unroll
    for (int k = 0; k < SPL; ++k) {
      if (k * 64 < ne0) bad |= stlive0 && w0[k] == POISON;
      if (k * 64 < ne1) bad |= stlive1 && w1[k] == POISON;
    }
    return bad;
  };
  issue_stats();
  issue_batch(cur, 0);
  u64 t0w = 0, t1w = 0;
  if (tot0 && wv == 0 && lane < 32) t0w = q0[lane];
  if (tot1 && wv == 0 && lane < 32) t1w = q1[lane];
  LK_STAMP(sy, 1);
  // the residual of the epilogue: requested here, OLDER than the weight ring (the compiler counts the loads YOUNGER than a fragment to wait
  // for it: conditional loads in front of the ring cost nothing, behind it they make every fragment wait for everything in flight)
  constexpr int NFM = JEN1_LONG_MAX_NF;
  const int out_C = LI(out_C), ps_f = LI(ps_f), ps_off = LI(ps_off), L_y = LI(L_y), y_brows = LI(y_brows), y_row0 = LI(y_row0);
  const int m0 = mblk * JEN1_LONG_BM;                        // first GEMM row of the unit; one sub-pixel phase per M block (out_C % 128 == 0)
  const int ph = m0 >> LI(outc_shift);
  const int co = (m0 - ph * out_C) + wv * 16 + lg * 4;       // this lane's 4 consecutive output channels
  const T* resb = LP(residual, const T*);
  const bool reslive = resb && ((live >> 8) & 1);
  const int ld_res = LI(ld_res), ld_y = LI(ld_y);
  int yrow[NFM];
#pragma unroll
  for (int nf = 0; nf < NFM; ++nf) {
    yrow[nf] = -1;
    if (nf < NF) {
      const int n = nf * 16 + li;
      const int q = t0 + n;
      const int ty = q * ps_f + ph - ps_off;
      const bool ok = n < tb && q < L_out && ty >= 0 && ty < L_y;
      yrow[nf] = ok ? b * y_brows + y_row0 + ty : -1;
    }
  }
  Raw4<T> rr[NFM];
  auto issue_res = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int nf = 0; nf < NFM; ++nf) {
      if (nf < NF) ld_live4r(rr[nf], resb + ((unsigned)(yrow[nf] < 0 ? 0 : yrow[nf]) * (unsigned)ld_res + (unsigned)co));
    }
  };
  auto res_bad = [&]() __attribute__((always_inline)) -> bool {
    bool bad = false;
#pragma unroll
    for (int nf = 0; nf < NFM; ++nf) {
      if (nf < NF) bad |= yrow[nf] >= 0 && raw_bad(rr[nf]);
    }
    return bad;
  };
  if (resb) issue_res();
  // the unit's weight slice: requested right behind the polled round (the polls of a unit go out as early as possible: a poll is a
  // round trip of ~0.7 us that starts only here), long before the MFMA loop needs it
  ring_fill<T>(dr, slot, lane, wv, ringA, ringB);

  // ---- behind the first polled round: per-channel parameters of the prologue, bias, output rows, the residual -------------------------------
  float g1 = 0.f, g2 = 0.f;
  if (gn && tid < cmain) {
    const int p_ld = LI(p_ld);
    const int* fstep = LP(film_step, const int*);
    const int* frow = LP(film_row, const int*);
    const int fr = p_ld ? (fstep ? fstep[0] : (frow ? frow[b] : b)) : 0;
    const size_t po = (size_t)((unsigned)fr * (unsigned)p_ld) + (unsigned)tid;
    g1 = LP(p1, const float*)[po];
    g2 = LP(p2, const float*)[po];
  }
  f32x4 bias4 = (f32x4){0.f, 0.f, 0.f, 0.f};
  {
    const float* biasp = LP(bias, const float*);
    if (biasp) bias4 = *reinterpret_cast<const f32x4*>(biasp + co);
  }
  {
    unsigned spins = 0;
    bool bad = batch_bad(cur, 0) | stats_bad();
    while (poll_again(sy, bad, spins)) {
      issue_stats();
      issue_batch(cur, 0);
      bad = batch_bad(cur, 0) | stats_bad();
    }
  }
  LK_STAMP(sy, 2);

  // ---- (2) partials -> sub-sums in LDS -> the affine pair of every channel ------------------------------------------------------------------
  if (gn) {
    auto reduce_src = [&](const u64 (&w)[SPL], int ne, int nsub, float2* dst) __attribute__((always_inline)) {
      // entry e was written by M block e % mb of its phase: slot wv of it holds channels 16 (wv + 8 cls), cls = (e % mb) * 8 % nsub / 8
      // = e & 1 for 256 channels (mb is even there: 2 M blocks per sub-pixel phase), 0 for 128
      float s0 = 0.f, qq0 = 0.f, s1 = 0.f, qq1 = 0.f;
#pragma unroll
      for (int k = 0; k < SPL; ++k) {
        if (k * 64 < ne) {
          const int e = lane + k * 64;
          const int cls = nsub > 8 ? (e & 1) : 0;
          const float s = __uint_as_float((unsigned)w[k]), q = __uint_as_float((unsigned)(w[k] >> 32));
          const bool in = e < ne;
          s0 += (in && cls == 0) ? s : 0.f; qq0 += (in && cls == 0) ? q : 0.f;
          s1 += (in && cls == 1) ? s : 0.f; qq1 += (in && cls == 1) ? q : 0.f;
        }
      }
      s0 = wave_sum(s0); qq0 = wave_sum(qq0);
      if (nsub > 8) { s1 = wave_sum(s1); qq1 = wave_sum(qq1); }
      if (lane == 0) {
        dst[wv] = make_float2(s0, qq0);
        if (nsub > 8) dst[8 + wv] = make_float2(s1, qq1);
      }
    };
    if (ne0) reduce_src(w0, ne0, LI(st_nsub[0]), stl);
    else if (wv == 0 && lane < 32) stl[lane] = make_float2(__uint_as_float((unsigned)t0w), __uint_as_float((unsigned)(t0w >> 32)));
    if (ne1) reduce_src(w1, ne1, LI(st_nsub[1]), stl + 32);
    else if (tot1 && wv == 0 && lane < 32) stl[32 + lane] = make_float2(__uint_as_float((unsigned)t1w), __uint_as_float((unsigned)(t1w >> 32)));
  }
  __syncthreads();              // (also: every wave has left the previous unit's MFMA loop before the tile is written again)
  LK_STAMP(sy, 3);
  if (gn && tid < cmain) {
    const int c = tid;
    const int groups = LI(gn_groups);
    const bool k1 = c >= c0;
    int i0, cnt;
    if (groups == 1) {
      i0 = 0;
      cnt = k1 ? LI(st_nsub[1]) : LI(st_nsub[0]);        // one group over everything (Patcher / Unpatcher, blocks.py:251, :279)
    } else {
      const int cs = LI(cpg_shift), gs = k1 ? LI(gran_shift[1]) : LI(gran_shift[0]);
      const int lo = ((c >> cs) << cs) - (k1 ? c0 : 0);
      i0 = lo >> gs;
      cnt = 1 << (cs - gs);
    }
    const float2* rk = stl + (k1 ? 32 : 0);
    float s = 0.f, q = 0.f;
    // (entries in fixed order; four independent reads in flight)
    for (int i = 0; i < cnt; i += 4) {
      float2 e4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) e4[j] = rk[(i + j < cnt) ? i0 + i + j : i0];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s += (i + j < cnt) ? e4[j].x : 0.f;
        q += (i + j < cnt) ? e4[j].y : 0.f;
      }
    }
    const float sc = k1 ? LF(src1_scale) : 1.0f;
    s *= sc;
    q *= sc * sc;
    const float inv_count = LF(inv_count);
    const float mean = s * inv_count;
    float var = q * inv_count - mean * mean;
    var = var < 0.f ? 0.f : var;
    const float rstd = PRECISE ? 1.0f / sqrtf(var + LF(gn_eps)) : rsqrtf(var + LF(gn_eps));
    const float A = rstd * g1;
    tabA[c] = A * sc;
    tabS[c] = g2 - mean * A;
  }
  if (gn) __syncthreads();
  LK_STAMP(sy, 4);

  // ---- (3) stage the tile: prologue applied once, zero padding applied after it ------------------------------------------------------------
  const float sc1 = LF(src1_scale);
  for (int v0 = 0; v0 < nvec; v0 += NT * MAXVB) {
    if (v0 > 0) {
      prep_batch(cur, v0);
      unsigned spins = 0;
      bool bad;
      do {
        issue_batch(cur, v0);
        bad = batch_bad(cur, v0);
      } while (poll_again(sy, bad, spins));
    }
#pragma unroll
    for (int k = 0; k < MAXVB; ++k) {
      const int v = v0 + k * NT + tid;
      if (v0 + k * NT >= nvec) continue;
      if (v >= nvec) continue;
      const int row = (int)(((float)v + 0.5f) * inv_vpr);
      const int c = (v - row * vpr) * 8;
      const int tin = tin0 + row;
      float x[8];
      raw_to_float(cur.x[k], x);
      if (tin >= 0 && tin < L_in) {
        if (c < cmain) {
          if (gn) {
            const float4 a0 = *reinterpret_cast<const float4*>(tabA + c), a1 = *reinterpret_cast<const float4*>(tabA + c + 4);
            const float4 e0 = *reinterpret_cast<const float4*>(tabS + c), e1 = *reinterpret_cast<const float4*>(tabS + c + 4);
            x[0] = x[0] * a0.x + e0.x; x[1] = x[1] * a0.y + e0.y; x[2] = x[2] * a0.z + e0.z; x[3] = x[3] * a0.w + e0.w;
            x[4] = x[4] * a1.x + e1.x; x[5] = x[5] * a1.y + e1.y; x[6] = x[6] * a1.z + e1.z; x[7] = x[7] * a1.w + e1.w;
#ifndef JEN1_LONG_EXP_NOSILU     // (timing experiment only: staging without the activation)
            if (do_silu) {
#pragma unroll
              for (int j = 0; j < 8; ++j) x[j] = PRECISE ? silu_precise(x[j]) : silu_f(x[j]);
            }
#endif
          } else if (c >= c0 && sc1 != 1.0f) {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] *= sc1;
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = 0.f;
      }
      store8(tile + row * pitch + c, x);
    }
  }
  __syncthreads();
  LK_STAMP(sy, 5);

  // ---- (4), (5): MFMA loop and epilogue, specialised by the number of position fragments ----------------------------------------------------
  float* const out_part = LP(out_part, float*);
  {
    constexpr int HR = RING / 2;
    int ldsrow[NFM];
#pragma unroll
    for (int nf = 0; nf < NFM; ++nf) {
      const int n = nf * 16 + li;
      ldsrow[nf] = ((n < tb ? n : 0) * stride) * pitch + lg * 8;
    }
    f32x4 acc[NFM];
#pragma unroll
    for (int nf = 0; nf < NFM; ++nf) acc[nf] = bias4;                 // the accumulators start from the bias
    {
      const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(LP(w, const void*)), 0, LI(w_bytes), RSRC_FLAGS);
      const unsigned voff = (unsigned)(mblk * 8 + wv) * BLK + (unsigned)lane * (8u * ES);
      const unsigned step = (unsigned)MT * BLK;
      // k-step ks reads the staged tile at element offset koff[ks] = tap * pitch + 32 chunk ((tap, chunk) order; behind the last tap the
      // extra chunks at the centre row): tabulated by the host, two per descriptor dword, a field read is one v_readlane.
      constexpr int KOFF_DW = offsetof(jen1_long_phase, koff) / 4 - 64;
      static_assert(offsetof(jen1_long_phase, koff) % 4 == 0 && KOFF_DW >= 0 && KOFF_DW + JEN1_LONG_MAX_KS / 2 <= 64, "the k-step table lies in the second descriptor register");
      auto koff = [&](int ks, bool odd) __attribute__((always_inline)) -> int {
        const int w2 = __builtin_amdgcn_readlane((int)dr.d1, KOFF_DW + (ks >> 1));
        return odd ? (int)((unsigned)w2 >> 16) : (w2 & 0xffff);
      };
      // HALF a ring round (HR k-steps from one half of the weight ring), one straight-line copy per fragment count: a wave retires an
      // instruction every ~2 ns here, guards per fragment and k-step cost more than the matrix work.  The copies only READ the ring:
      // refills happen outside the dispatch (a copy that redefines ring registers leaves the allocator with six-way merges of all 96 of
      // them -- 150 spilled registers).  Activation fragments of k-step k + 1 are requested before the MFMAs of k-step k (two register
      // sets by the parity of the slot; HR is even).
      auto half = [&](auto nfc, const Frag (&rg)[HR], int ksb) __attribute__((always_inline)) {
        constexpr int NFW = decltype(nfc)::value;
        // fragment sets in flight: the LDS round trip (~130 clocks) against the matrix work of a k-step (NFW x 32 clocks)
        constexpr int D = is_f32<T>::value ? 2 : (NFW == 1 ? 4 : NFW == 2 ? 3 : 2);
        Frag bb[D][NFW];
        auto bload = [&](Frag (&dst)[NFW], int off) __attribute__((always_inline)) {
          const T* bp = tile + off;
#pragma unroll
          for (int nf = 0; nf < NFW; ++nf) llds(dst[nf], bp + ldsrow[nf]);
        };
        // straight-line, no guard per k-step (a conditional LDS read makes the compiler wait for ALL reads in flight at every join): a
        // k-step beyond KS multiplies a ZERO weight fragment (ring slots beyond the slice are out-of-range loads) with the tile's first
        // rows (the table's entries beyond KS are 0)
#pragma unroll
        for (int j = 0; j < D - 1; ++j) bload(bb[j], koff(ksb + j, (j & 1) != 0));
#pragma unroll
        for (int s = 0; s < HR; ++s) {
          if (s + D - 1 < HR) bload(bb[(s + D - 1) % D], koff(ksb + s + D - 1, ((s + D - 1) & 1) != 0));
#pragma unroll
          for (int nf = 0; nf < NFW; ++nf) lmma(acc[nf], rg[s], bb[s % D][nf]);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      auto dispatch = [&](const Frag (&rg)[HR], int ksb) __attribute__((always_inline)) {
        switch (NF) {
          case 1: half(std::integral_constant<int, 1>(), rg, ksb); break;
          case 2: half(std::integral_constant<int, 2>(), rg, ksb); break;
          case 3: half(std::integral_constant<int, 3>(), rg, ksb); break;
          case 4: half(std::integral_constant<int, 4>(), rg, ksb); break;
          case 5: half(std::integral_constant<int, 5>(), rg, ksb); break;
          default: half(std::integral_constant<int, 6>(), rg, ksb); break;
        }
      };
      static_assert(HR % 2 == 0, "the parity of a k-step is the parity of its slot");
#ifdef JEN1_LONG_EXP_NOMMA       // timing experiment only: no MFMA loop
      for (int ks0 = KS; ks0 < KS; ks0 += RING) {
#else
      for (int ks0 = 0; ks0 < KS; ks0 += RING) {
#endif
        dispatch(ringA, ks0);
        if (ks0 + RING < KS) {                            // the next round's first half: in flight during this round's second half
#pragma unroll
          for (int s = 0; s < HR; ++s) {
            const bool in = ks0 + RING + s < KS;
            wload(ringA[s], rw, in ? voff : OOB, in ? (unsigned)(ks0 + RING + s) * step : 0u);
          }
        }
        if (ks0 + HR < KS) {
          dispatch(ringB, ks0 + HR);
          if (ks0 + RING + HR < KS) {
#pragma unroll
            for (int s = 0; s < HR; ++s) {
              const bool in = ks0 + RING + HR + s < KS;
              wload(ringB[s], rw, in ? voff : OOB, in ? (unsigned)(ks0 + RING + HR + s) * step : 0u);
            }
          }
        }
      }
    }
    LK_STAMP(sy, 6);
    if (reslive) {
      unsigned spins = 0;
      bool rbad = res_bad();
      while (poll_again(sy, rbad, spins)) {
        issue_res();
        rbad = res_bad();
      }
    }
    float gs = 0.f, gq = 0.f;
#pragma unroll
    for (int nf = 0; nf < NFM; ++nf) {
      if (nf < NF && yrow[nf] >= 0) {
        float v[4] = {acc[nf][0], acc[nf][1], acc[nf][2], acc[nf][3]};
        if (resb) {
          float r4[4];
          raw4_to_float(rr[nf], r4);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += r4[r];
        }
        st_live4<LOC>(LP(y, T*) + ((unsigned)yrow[nf] * (unsigned)ld_y + (unsigned)co), v);
        gs += (v[0] + v[1]) + (v[2] + v[3]);
        gq += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
      }
    }
    if (out_part) {
      // the wave's 16 channels x its valid positions: all 64 lanes, fixed tree, one word
      gs = row16_sum_l(gs); gq = row16_sum_l(gq);
      const float ts = (rlane(gs, 0) + rlane(gs, 16)) + (rlane(gs, 32) + rlane(gs, 48));
      const float tq = (rlane(gq, 0) + rlane(gq, 16)) + (rlane(gq, 32) + rlane(gq, 48));
      if (lane == 0) {
        const int ne = LI(out_entries);
        st_word<LOC>(out_part + 2 * ((size_t)(b * 8 + wv) * ne + slot), 0, __float_as_uint(ts), __float_as_uint(tq));
      }
    }
  }
  LK_STAMP(sy, 7);
}

// TK = false: the static map -- sample b = workgroup % B, slot = workgroup / B in EVERY phase; the next unit's descriptor and weight
// slice are requested during / right behind the current unit.  Correct only while all workgroups of the launch are resident together
// (the caller holds the device's static schedule: engine.DeepProgram.claim_static).
// TK = true: units by ticket from one device counter (zero when the launch starts): ticket t is unit t % U of phase t / U with
// U = B * G unit slots per phase (slots beyond a phase's units are empty).  Whoever holds the smallest unfinished ticket depends only on
// smaller tickets, which are finished or held by running workgroups: the launch makes progress with ANY number of resident workgroups, so
// it can share the GPU with other persistent launches.  No cross-phase prefetch: slower, safe.
template <typename T, bool TK, bool LOC>
__global__ __launch_bounds__(NT) void long_kernel(const unsigned char* __restrict__ descs, int n_phases, int Bs, unsigned* err, unsigned* ticket) {
  typedef typename LFrag<T>::type Frag;
  constexpr int RING = LongCfg<T>::RING;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = rfl(tid >> 6);
  const int wg = blockIdx.x;
  const int G = (int)gridDim.x / Bs;
  LSync sy;
  sy.err = err;
  sy.dead = false;
  Frag ringA[RING / 2], ringB[RING / 2];
  if constexpr (TK) {
    __shared__ int tick_s;
    const int U = Bs * G, total = n_phases * U;
    for (;;) {
      __syncthreads();                                   // (the previous round's ticket has been read by everybody; LDS is free)
      if (tid == 0) tick_s = (int)__hip_atomic_fetch_add(g32(ticket), 1u, RLX_AGENT);
      __syncthreads();
      const int t = rfl(tick_s);
      if (t >= total) break;
      const int p = t / U, u = t - p * U;
      const int b = u % Bs, slot = u / Bs;
      const DescRegs dr = load_desc(descs, p, lane);
      if (slot >= LI(out_entries)) continue;
      sy.p = p;
      long_unit<T, false>(dr, b, slot, sy, ringA, ringB, tid);
    }
    return;
  }
  int b = wg % Bs, slot = wg / Bs;
  if constexpr (LOC) {
    // plain stores reach their readers only inside one XCD, and HIP promises nothing about which XCD a workgroup runs on (observed: workgroup
    // i on XCD i % 8 on an idle device, a rotation of that behind other kernels).  So the groups are formed from where the workgroups
    // ACTUALLY are: every workgroup reads its XCD and takes a number t on it (one counter per XCD, zero when the launch starts); XCD x
    // hosts samples x, x + 8, ... -- sample x + 8 (t / G), slot t % G.  With one workgroup per CU and every workgroup resident each XCD
    // holds exactly nwg / 8 of them; one that finds its XCD full raises the error word.
    __shared__ int xt_s;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xfu;
    if (tid == 0) xt_s = (int)__hip_atomic_fetch_add(g32(ticket) + 8 + xcc, 1u, RLX_AGENT);
    __syncthreads();
    const int t = rfl(xt_s);
    const int cap = (int)gridDim.x / 8;
    if (xcc >= 8u || t >= cap) {
      if (tid == 0) __hip_atomic_store(g32(err), 0x40000000u | (unsigned)wg, RLX_AGENT);
      return;
    }
    b = (int)xcc + 8 * (t / G);
    slot = t % G;
  }
  if (slot >= G || b >= Bs) return;
  DescRegs dr = load_desc(descs, 0, lane);
  for (int p = 0; p < n_phases; ++p) {
    const bool more = p + 1 < n_phases;
    DescRegs nx = dr;
    if (more) nx = load_desc(descs, p + 1, lane);       // in flight during the unit
    sy.p = p;
    const bool mine = slot < LI(out_entries);
#ifdef JEN1_LONG_PROFILE
#pragma unroll
    for (int i_ = 0; i_ < 8; ++i_) sy.tt[i_] = 0;
#endif
    if (mine) long_unit<T, LOC>(dr, b, slot, sy, ringA, ringB, tid);
#ifdef JEN1_LONG_PROFILE
    if (tid == 0 && g_long_dbg) {
#pragma unroll
      for (int i_ = 0; i_ < 8; ++i_) g_long_dbg[((size_t)p * gridDim.x + wg) * 8 + i_] = sy.tt[i_];
    }
#endif
    if (!more) break;
    dr = nx;
  }
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline int align16i(int x) { return (x + 15) & ~15; }

}  // namespace

extern "C" int jen1_long_debug_buffer(void* p) {
#ifdef JEN1_LONG_PROFILE
  return hipMemcpyToSymbol(HIP_SYMBOL(g_long_dbg), &p, sizeof(p)) == hipSuccess ? 0 : 1;
#else
  (void)p;
  return jen1_set_error("jen1_long_debug_buffer: the library was built without -DJEN1_LONG_PROFILE");
#endif
}

extern "C" int jen1_long_geometry(int M, int L_out, int G, int* mblocks, int* tiles_t, int* tb) {
  JEN1_CHECK(M >= JEN1_LONG_BM && M % JEN1_LONG_BM == 0, "long phase: %d GEMM rows are not a multiple of %d", M, JEN1_LONG_BM);
  JEN1_CHECK(G >= 1 && L_out >= 1, "long phase: bad geometry");
  const int mb = M / JEN1_LONG_BM;
  JEN1_CHECK(mb <= G, "long phase: %d M blocks on %d workgroups per sample", mb, G);
  const int tiles = G / mb;
  const int t = ceil_div(L_out, tiles);
  JEN1_CHECK(t <= 16 * JEN1_LONG_MAX_NF, "long phase: %d positions on %d tiles: %d per tile (at most %d)", L_out, tiles, t, 16 * JEN1_LONG_MAX_NF);
  if (mblocks) *mblocks = mb;
  if (tiles_t) *tiles_t = tiles;
  if (tb) *tb = t;
  return 0;
}

extern "C" int jen1_long_phase_conv(const jen1_conv_args* a, int G, const float* st0, int st0_entries, int st0_mblocks, int st0_nsub,
                                    const float* st1, int st1_entries, int st1_mblocks, int st1_nsub, int st_live, float* out_part,
                                    jen1_long_phase* out) {
  JEN1_CHECK(a && out, "long phase: null pointer");
  JEN1_CHECK(a->dtype == JEN1_F32 || a->dtype == JEN1_BF16, "long phase: dtype must be float32 or bf16 (the JEN1_FP8 mode runs the long levels in bf16)");
  JEN1_CHECK(a->x0 && a->w && a->y, "long phase: null tensor");
  JEN1_CHECK(a->pro_mode == JEN1_PRO_NONE || a->pro_mode == JEN1_PRO_GN || a->pro_mode == JEN1_PRO_GN_SILU, "long phase: prologue %d is not supported", a->pro_mode);
  JEN1_CHECK(!a->ln_fold && !a->row_scale && !a->out_rowstats && a->act == JEN1_ACT_NONE && a->m_split == 0 && !a->y_f32,
             "long phase: LayerNorm / row scale / activation / dual range / float32 output are not options of the long levels");
  JEN1_CHECK(a->c0 > 0 && a->c0 % 32 == 0 && a->c1 % 32 == 0, "long phase: channels must be multiples of 32");
  JEN1_CHECK(a->taps >= 1 && a->stride >= 1 && a->B >= 1 && a->L_in >= 1 && a->L_out >= 1, "long phase: bad geometry");
  JEN1_CHECK(a->nseg >= 0 && a->nseg <= 2, "long phase: at most two raw extra K segments");
  const int es = a->dtype == JEN1_F32 ? 4 : 2;
  const bool gn = a->pro_mode != JEN1_PRO_NONE;
  jen1_long_phase& p = *out;
  memset(&p, 0, sizeof(p));
  int mb = 0, tiles = 0, tb = 0;
  if (jen1_long_geometry(a->M, a->L_out, G, &mb, &tiles, &tb)) return 1;
  const int ps_f = a->ps_f < 1 ? 1 : a->ps_f;
  JEN1_CHECK(a->out_C % JEN1_LONG_BM == 0 && a->out_C * ps_f == a->M, "long phase: %d output channels (a multiple of %d: an M block never straddles a sub-pixel phase)",
             a->out_C, JEN1_LONG_BM);
  JEN1_CHECK(a->ld_y % 4 == 0 && (!a->residual || a->ld_res % 4 == 0), "long phase: output pitches must be multiples of 4");
  p.src[0] = jen1_long_src{a->x0, a->ld0, a->c0};
  int cmain = a->c0;
  if (a->c1) {
    JEN1_CHECK(a->x1, "long phase: c1 without x1");
    p.src[1] = jen1_long_src{a->x1, a->ld1, a->c1};
    cmain += a->c1;
  } else {
    p.src[1] = jen1_long_src{a->x0, a->ld0, 0};
  }
  JEN1_CHECK(cmain <= NT, "long phase: %d main channels (at most %d: one affine pair per thread)", cmain, NT);
  int call = cmain, kx = 0;
  int live = a->live_mask & 1;
  int bit = 1;
  if (a->c1) { live |= ((a->live_mask >> bit) & 1) << 1; ++bit; }
  for (int s = 0; s < 2; ++s) {
    if (s < a->nseg) {
      const jen1_conv_seg& e = a->seg[s];
      JEN1_CHECK(e.x && e.kch > 0 && e.shift == 0 && e.ld >= 32 * e.kch, "long phase: bad extra segment %d", s);
      p.src[2 + s] = jen1_long_src{e.x, e.ld, 32 * e.kch};
      call += 32 * e.kch;
      kx += e.kch;
      live |= ((a->live_mask >> bit) & 1) << (2 + s);
      ++bit;
    } else {
      p.src[2 + s] = jen1_long_src{a->x0, a->ld0, 0};
    }
  }
  JEN1_CHECK(a->nseg == 0 || a->stride == 1, "long phase: extra K segments need stride 1");
  live |= a->live_mask & 256;
  p.cmain = cmain; p.call = call; p.pitch = call + 8;
  p.kch = cmain / 32;
  p.KS = a->taps * p.kch + kx;
  p.MT = a->M / 16;
  const int64_t wb = (int64_t)p.KS * p.MT * 512 * es;
  JEN1_CHECK(wb < ((int64_t)1 << 31), "long phase: packed weight too large for 31-bit offsets");
  p.w = a->w; p.w_bytes = (uint32_t)wb;
  p.B = a->B; p.G = G; p.L_in = a->L_in; p.L_out = a->L_out; p.stride = a->stride; p.taps = a->taps; p.pad_left = a->pad_left;
  p.tb = tb; p.tiles_t = tiles; p.mblocks = mb; p.NF = ceil_div(tb, 16);
  p.rows_in = (tb - 1) * a->stride + a->taps;
  p.out_entries = tiles * mb;
  p.pro_mode = a->pro_mode;
  p.src1_scale = a->src1_scale;
  if (gn) {
    JEN1_CHECK(a->gn_gamma && a->gn_beta && a->gn_groups >= 1 && a->gn_cpg >= 1 && a->gn_count >= 1 && st0, "long phase: incomplete GroupNorm");
    JEN1_CHECK(a->gn_groups > 1 || a->c1 == 0, "long phase: a single GroupNorm group over two sources is not supported");
    p.gn_groups = a->gn_groups;
    p.gn_cpg = a->gn_groups == 1 ? cmain : a->gn_cpg;
    p.inv_count = 1.0f / (float)a->gn_count;
    p.gn_eps = a->gn_eps;
    if (a->film) {
      JEN1_CHECK(a->film_C == cmain && a->film_ld >= a->film_off + 2 * a->film_C, "long phase: bad FiLM table geometry");
      p.p1 = a->film + a->film_off;
      p.p2 = a->film + a->film_off + a->film_C;
      p.p_ld = a->film_ld;
      p.film_row = a->film_row; p.film_step = a->film_step;
    } else {
      p.p1 = a->gn_gamma; p.p2 = a->gn_beta; p.p_ld = 0;
    }
    const float* stp[2] = {st0, a->c1 ? st1 : st0};
    const int ent[2] = {st0_entries, a->c1 ? st1_entries : 0}, mbs[2] = {st0_mblocks, st1_mblocks}, nsb[2] = {st0_nsub, st1_nsub};
    JEN1_CHECK(!a->c1 || st1, "long phase: the second normalised source needs statistics");
    for (int k = 0; k < (a->c1 ? 2 : 1); ++k) {
      const int ck = k ? a->c1 : a->c0, ldk = k ? a->ld1 : a->ld0;
      p.st[k] = stp[k];
      p.st_entries[k] = ent[k];
      if (ent[k] > 0) {
        JEN1_CHECK(ent[k] <= 64 * SPL && mbs[k] >= 1 && (nsb[k] == 8 || nsb[k] == 16) && nsb[k] * 16 == ck,
                   "long phase: statistics partials of source %d: %d entries, %d M blocks, %d sub-groups for %d channels", k, ent[k], mbs[k], nsb[k], ck);
        p.st_mblocks[k] = mbs[k]; p.st_nsub[k] = nsb[k]; p.st_gran[k] = 16;
      } else {
        JEN1_CHECK(ldk % 32 == 0, "long phase: a source with totals must have a pitch that is a multiple of 32");
        p.st_mblocks[k] = 1; p.st_nsub[k] = 32; p.st_gran[k] = ldk / 32;
      }
      JEN1_CHECK(a->gn_groups == 1 || (p.gn_cpg % p.st_gran[k] == 0 && (k == 0 || a->c0 % p.gn_cpg == 0)),
                 "long phase: GroupNorm groups of %d channels are not made of whole statistics entries of %d", p.gn_cpg, p.st_gran[k]);
    }
    if (!a->c1) { p.st[1] = st0; p.st_mblocks[1] = 1; p.st_nsub[1] = 8; p.st_gran[1] = 16; }
    live |= (st_live & (a->c1 ? 3 : 1)) << 4;
  } else {
    p.st[0] = p.st[1] = nullptr;
    p.st_mblocks[0] = p.st_mblocks[1] = 1; p.st_nsub[0] = p.st_nsub[1] = 8; p.st_gran[0] = p.st_gran[1] = 16;
  }
  p.live_mask = live;
  p.bias = a->bias; p.residual = a->residual; p.y = a->y;
  p.out_C = a->out_C; p.ps_f = ps_f; p.ps_off = a->ps_off; p.L_y = a->L_y; p.y_brows = a->y_brows; p.y_row0 = a->y_row0;
  p.ld_y = a->ld_y; p.ld_res = a->ld_res;
  JEN1_CHECK((int64_t)a->B * a->L_in * (a->ld0 > a->ld1 ? a->ld0 : a->ld1) < ((int64_t)1 << 31) && (int64_t)a->B * a->y_brows * a->ld_y < ((int64_t)1 << 31),
             "long phase: tensor too large for 32-bit element offsets");
  p.out_part = out_part;
  JEN1_CHECK(!out_part || a->out_C == 128 || a->out_C == 256, "long phase: partial statistics of %d output channels (128 or 256)", a->out_C);
  const int tile_b = align16i(p.rows_in * p.pitch * es);
  p.tab_off = tile_b;
  p.st_off = tile_b + align16i(8 * cmain);
  const int tot = p.st_off + 2 * 32 * 8;
  JEN1_CHECK(tot <= LDS_TOTAL - 1024, "long phase: %d B of LDS", tot);
  p.lds_bytes = tot;
  p.inv_vpr = 1.0f / (float)(call / 8);
  // (the unit reads table entries up to 14 k-steps beyond KS: zeros, inside the 64-entry table)
  JEN1_CHECK(p.KS <= JEN1_LONG_MAX_KS - 16, "long phase: %d k-steps (at most %d)", p.KS, JEN1_LONG_MAX_KS - 16);
  JEN1_CHECK((a->taps - 1) * p.pitch + call < 65536, "long phase: staged tile offsets do not fit 16 bits");
  {
    int ks = 0;
    for (int tap = 0; tap < a->taps; ++tap)
      for (int c = 0; c < p.kch; ++c) p.koff[ks++] = (uint16_t)(tap * p.pitch + c * 32);
    for (int j = 0; j < kx; ++j) p.koff[ks++] = (uint16_t)(a->pad_left * p.pitch + cmain + j * 32);
  }
  auto log2i = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
  p.mb_shift = log2i(mb);
  p.outc_shift = log2i(a->out_C);
  JEN1_CHECK(p.mb_shift >= 0 && p.outc_shift >= 0, "long phase: %d M blocks / %d output channels are not powers of two", mb, a->out_C);
  if (gn && a->gn_groups > 1) {
    p.cpg_shift = log2i(p.gn_cpg);
    p.gran_shift[0] = log2i(p.st_gran[0]);
    p.gran_shift[1] = log2i(p.st_gran[1]);
    JEN1_CHECK(p.cpg_shift >= 0 && p.gran_shift[0] >= 0 && p.gran_shift[1] >= 0 && p.cpg_shift >= p.gran_shift[0] && p.cpg_shift >= p.gran_shift[1],
               "long phase: GroupNorm groups of %d channels over statistics entries of %d / %d channels (powers of two)", p.gn_cpg, p.st_gran[0], p.st_gran[1]);
  }
  return 0;
}

namespace {
template <typename T, bool TK, bool LOC>
int launch_long(const unsigned char* descs, int n_phases, int B, uint32_t* err, uint32_t* ticket, int nwg, int lds, hipStream_t s) {
  auto kern = long_kernel<T, TK, LOC>;
  JEN1_MAX_LDS_ONCE(kern, LDS_TOTAL - 1024);      // (the ticket form keeps one static LDS word)
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(NT), (size_t)lds, s, descs, n_phases, B, err, ticket);
  JEN1_HIP(hipGetLastError());
  return 0;
}
__global__ __launch_bounds__(NT) void census_kernel(int* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (threadIdx.x == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[blockIdx.x] = (int)(xcc & 0xfu);
    smem[0] = 0;
  }
}
}  // namespace

extern "C" int jen1_long_phase_lds(const jen1_long_phase* p) { return p ? p->lds_bytes : 0; }
extern "C" int jen1_long_phase_units(const jen1_long_phase* p) { return p ? p->out_entries : 0; }

extern "C" int jen1_long_census(int* out_dev, int nwg, void* stream) {
  JEN1_CHECK(out_dev && nwg >= 1, "long census: bad arguments");
  JEN1_MAX_LDS_ONCE(census_kernel, LDS_TOTAL - 1024);
  hipLaunchKernelGGL(census_kernel, dim3(nwg), dim3(NT), (size_t)LDS_MIN, reinterpret_cast<hipStream_t>(stream), out_dev);
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_long_run(const void* descs_dev, int n_phases, int B, uint32_t* err, uint32_t* ticket, int nwg, int lds_bytes, int dtype,
                             int local, void* stream) {
  JEN1_CHECK(descs_dev && err && n_phases >= 1 && n_phases <= JEN1_LONG_MAX_PHASES && B >= 1 && nwg >= B, "long run: bad arguments");
  JEN1_CHECK(lds_bytes > 0 && lds_bytes <= LDS_TOTAL - 1024, "long run: %d B of LDS", lds_bytes);
  JEN1_CHECK(dtype == JEN1_F32 || dtype == JEN1_BF16 || dtype == JEN1_FP8, "long run: bad dtype");
  JEN1_CHECK(!local || (ticket && B % 8 == 0 && nwg % B == 0 && nwg % 8 == 0),
             "long run: XCD-local stores need the zeroed synchronisation words, a multiple of 8 samples and whole groups (a sample's workgroups on one XCD)");
  const int lds = lds_bytes < LDS_MIN ? LDS_MIN : lds_bytes;       // one workgroup per CU, always
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const unsigned char* d = reinterpret_cast<const unsigned char*>(descs_dev);
  const bool f32 = dtype == JEN1_F32;
  if (local) return f32 ? launch_long<float, false, true>(d, n_phases, B, err, ticket, nwg, lds, s) : launch_long<bf16_t, false, true>(d, n_phases, B, err, ticket, nwg, lds, s);
  if (ticket) return f32 ? launch_long<float, true, false>(d, n_phases, B, err, ticket, nwg, lds, s) : launch_long<bf16_t, true, false>(d, n_phases, B, err, ticket, nwg, lds, s);
  return f32 ? launch_long<float, false, false>(d, n_phases, B, err, ticket, nwg, lds, s) : launch_long<bf16_t, false, false>(d, n_phases, B, err, ticket, nwg, lds, s);
}
