// Weight-streaming GEMM for the deep levels of the JEN-1 denoiser (gfx950 / MI355X).
//
// The "direct" mode of jen1_conv_gemm (include/jen1_hip.h): the activation operand is tiny
// (B*T' <= 64 rows per tile) and needs no prologue, a launch is pure weight streaming
// (reference jen1/model/blocks.py: _Conv1d :34-53, Upsample1d :69-95, Attention projections
// :427-429, FeedForward :440-446 at T' <= 24).  With ~64-384 workgroups there is ONE wave per SIMD,
// so the kernel is bound by its dynamic INSTRUCTION COUNT and by serial memory round trips, not by
// bandwidth or MFMA rate.  Everything here is shaped by that:
//   * the K axis is a flat list of 32-channel chunks over up to JEN1_MAX_SEG *segments*
//     (source tensor, row shift): conv taps, concatenated sources and a fused 1x1 shortcut are all
//     just segments; weights are packed [chunk][m-tile][lane][8] so the weight address is one
//     scalar offset that advances by a constant;
//   * both operands come through buffer descriptors: 32-bit per-lane offsets computed once per
//     segment, scalar offsets per chunk, and rows that fall into the conv zero padding are
//     out-of-range offsets (the hardware returns 0) -- no selects, no zero row, no 64-bit math;
//   * the first thing a wave does is fill its prefetch ring; epilogue operands are requested after
//     that and consumed at the end;
//   * segments whose rows are all padding for this tile are skipped together with their weights
//     (at T' = 1 that is 2/3 of a k=3 conv) -- exact.
//   * 16-row M tile per workgroup, the 4 waves split K and reduce through LDS; optional
//     inter-workgroup split-K with write-through slabs + ticket (cdna_hip_programming.md G16 R1).
#include "common.h"

#ifdef JEN1_PROFILE
#define SG_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0 && a.dbg) \
    a.dbg[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SG_STAMP(i) do { } while (0)
#endif

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct SegDev {
  const void* x;      // [B][L_in][ld]
  uint32_t nbytes;    // extent of x in bytes (buffer descriptor range)
  int32_t ldb;        // row pitch in bytes
  int32_t shift;      // input row = q*stride + shift
  int32_t gend;       // cumulative chunk count up to and including this segment
};

// Kernel arguments, ordered by when a wave needs them.  Scalar (kernarg) loads are a memory round
// trip each; the kernel reads every field of a block in straight-line code so that they leave as ONE
// batch: `hot` before the first weight load, `epi` while the prefetch ring is in flight.
struct HotArgs {
  const void* w;
  uint32_t w_bytes;
  int32_t G, cps, MT;
  int32_t B, L_in, L_out, stride, tb, nb, tiles_t;
  float inv_tiles_t, inv_tb;
  int32_t nseg;
  SegDev first[4];    // copies of seg[0..3]: the first usable segment is picked without a dependent load
  int32_t mt_split;   // dual-range GEMM: M tiles below it sum only the first g_split chunks (0 = off)
  int32_t g_split;
  int32_t pad[6];
};
struct EpiArgs {
  const float* bias;          // every pointer has a byte extent: 0 = absent (descriptor loads return 0)
  const void* residual;
  const float* row_scale;
  const float* ln_u;
  const float* ln_rowstats;
  void* y;
  float* out_gn_stats;
  float* out_rowstats;
  uint32_t bias_bytes, res_bytes, rsc_bytes, lnu_bytes, lnrs_bytes;
  int32_t out_C, ps_f, ps_off, L_y, y_brows, y_row0, ld_y, ld_res, y_f32, act, ln_fold, ngrp;
  float inv_cpf, ln_eps, inv_lnC;
};
struct StreamArgs {
  HotArgs hot;
  EpiArgs epi;
  float* slab;
  unsigned* counters;
  unsigned long long* dbg;
  int32_t splitk;
  SegDev seg[JEN1_MAX_SEG];
};

static_assert(sizeof(HotArgs) == 192 && sizeof(EpiArgs) == 144 && offsetof(StreamArgs, epi) == 192, "kernarg blocks are read with fixed-size scalar loads");

typedef unsigned int u32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x8 __attribute__((ext_vector_type(8)));

// One batch of scalar loads + one wait.  (Left to the compiler, the kernarg reads are sunk next to their
// uses behind uniform branches: 6-8 dependent scalar-memory round trips before the first weight load.)
__device__ __forceinline__ HotArgs load_hot_args() {
  const auto kp = __builtin_amdgcn_kernarg_segment_ptr();
  u32x16 k0, k1, k2;
  unsigned t0, t1, t2, t3;
  // the four single-dword loads only pull the epilogue block and the head of the segment table into the
  // scalar cache, so that the later batches hit it instead of paying a memory round trip
  asm volatile("s_load_dwordx16 %0, %7, 0x0\n\ts_load_dwordx16 %1, %7, 0x40\n\ts_load_dwordx16 %2, %7, 0x80\n\t"
               "s_load_dword %3, %7, 0xc0\n\ts_load_dword %4, %7, 0x100\n\ts_load_dword %5, %7, 0x140\n\ts_load_dword %6, %7, 0x180\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&s"(k0), "=&s"(k1), "=&s"(k2), "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3) : "s"(kp) : "memory");
  struct Raw { unsigned d[48]; } raw;
#pragma unroll
  for (int i = 0; i < 16; ++i) { raw.d[i] = k0[i]; raw.d[16 + i] = k1[i]; raw.d[32 + i] = k2[i]; }
  return __builtin_bit_cast(HotArgs, raw);
}
__device__ __forceinline__ EpiArgs load_epi_args() {
  const auto kp = __builtin_amdgcn_kernarg_segment_ptr();
  u32x16 k0, k1;
  u32x4 k2;
  asm volatile("s_load_dwordx16 %0, %3, 0xc0\n\ts_load_dwordx16 %1, %3, 0x100\n\ts_load_dwordx4 %2, %3, 0x140\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(k0), "=&s"(k1), "=&s"(k2) : "s"(kp) : "memory");
  struct Raw { unsigned d[36]; } raw;
#pragma unroll
  for (int i = 0; i < 16; ++i) { raw.d[i] = k0[i]; raw.d[16 + i] = k1[i]; }
#pragma unroll
  for (int i = 0; i < 4; ++i) raw.d[32 + i] = k2[i];
  return __builtin_bit_cast(EpiArgs, raw);
}

constexpr unsigned OOB = 0x80000000u;     // per-lane offset beyond every descriptor range: loads return 0
constexpr int RSRC_FLAGS = 0x00020000;

template <typename T> struct Frag8;
template <> struct Frag8<bf16_t> { typedef bf16x8 type; };
template <> struct Frag8<float> { typedef f32x8 type; };

__device__ __forceinline__ void mma(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma(f32x4& acc, const f32x8& a, const f32x8& b) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], acc, 0, 0, 0);
}

// one 8-element fragment through a buffer descriptor: address = base + voff + soff
template <int AUX>
__device__ __forceinline__ void bload(bf16x8& f, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX);
  f = __builtin_bit_cast(bf16x8, v);
}
template <int AUX>
__device__ __forceinline__ void bload(f32x8& f, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX);
  const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(r, voff + 16u, soff, AUX);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f.v[j] = __uint_as_float(lo[j]);
    f.v[4 + j] = __uint_as_float(hi[j]);
  }
}

#ifndef JEN1_X_AUX
#define JEN1_X_AUX 0      // cache policy of the activation operand (re-read by every M-tile workgroup: keep it cached)
#endif
#ifndef JEN1_W_AUX
#define JEN1_W_AUX 2      // cache policy of the weight stream: nt (each weight byte is used once per launch; measured +7 % end to end)
#endif

#ifndef JEN1_STREAM_WAVES
#define JEN1_STREAM_WAVES 4   // waves per workgroup = in-workgroup split of the K chunks (tuning builds: -DJEN1_STREAM_WAVES=8)
#endif
constexpr int SG_NW = JEN1_STREAM_WAVES;

template <typename T, int NF, int PF>
__global__ __launch_bounds__(64 * SG_NW) void stream_gemm_kernel(const StreamArgs a) {
  typedef typename Frag8<T>::type Frag;
  constexpr bool PRECISE = is_f32<T>::value;
  constexpr unsigned ES = sizeof(T);
  constexpr unsigned BLK = 512 * ES;      // bytes of one (chunk, m-tile) weight block
  constexpr unsigned CHB = 32 * ES;       // bytes of one 32-channel chunk of an activation row
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* misc = reinterpret_cast<int*>(smem);
  float* red = smem + 4;                   // [SG_NW - 1][NF][256]
  float* st_lds = red + (SG_NW - 1) * NF * 256;      // [nb][ngrp fine groups][2]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wk = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  SG_STAMP(0);

  // ---- everything the first weight load depends on, from ONE batch of kernarg loads ----------
  const HotArgs h = load_hot_args();
  const int by = blockIdx.y;
  const int bt = (int)(((float)by + 0.5f) * h.inv_tiles_t), tt = by - bt * h.tiles_t;
  const int b0 = bt * h.nb, t0 = tt * h.tb;
  const int mt = blockIdx.x, z = blockIdx.z;
  const int g0 = z * h.cps;
  const bool low_m = mt < h.mt_split;         // dual-range GEMM (see jen1_conv_args.m_split)
  const int g_end = low_m ? h.g_split : h.G;
  const int g1 = (g0 + h.cps < g_end) ? g0 + h.cps : g_end;
  const int t_last = ((t0 + h.tb < h.L_out) ? t0 + h.tb : h.L_out) - 1;
  const int n_rows = h.nb * h.tb;
  const int tmin = t0 * h.stride, tmax = t_last * h.stride;

  // ---- this lane's activation rows (one per 16-column fragment) ------------------------------
  int rowi[NF], tpos[NF], n_b[NF], n_t[NF];
  bool n_ok[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int n = nf * 16 + li;
    const int bl = (int)(((float)n + 0.5f) * h.inv_tb), tl = n - bl * h.tb;     // exact for n < 64
    n_ok[nf] = (n < n_rows) && (b0 + bl < h.B) && (t0 + tl < h.L_out);
    n_b[nf] = bl;
    n_t[nf] = tl;
    rowi[nf] = (b0 + bl) * h.L_in;
    tpos[nf] = (t0 + tl) * h.stride;
  }

  f32x4 acc[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) acc[nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- K cursor of this wave: chunk cur_g of segment s_cur, chunks lo+wk, lo+wk+4, ... ----------
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(h.w), 0, (int)h.w_bytes, RSRC_FLAGS);
  __amdgpu_buffer_rsrc_t rx = rw;
  unsigned voffA = (unsigned)lane * (8u * ES);
  unsigned voff[NF];
  unsigned soffA = 0, soffB = 0;
  int s_cur = -1, cur_g = 0, cur_hi = 0;
  int issued = 0, total = 0x7fffffff;
  bool parked = false;
  // enter segment: per-lane row offsets (rows in the conv zero padding -> out-of-range -> 0)
  auto enter = [&](const void* x, unsigned nbytes, int ldb, int sh, int sb, int lo, int hi) {
    cur_g = lo + wk;
    cur_hi = hi;
    soffA = (unsigned)(cur_g * h.MT + mt) * BLK;
    soffB = (unsigned)(cur_g - sb) * CHB;
    rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(x), 0, (int)nbytes, RSRC_FLAGS);
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const int tin = tpos[nf] + sh;
      const bool ok = n_ok[nf] && tin >= 0 && tin < h.L_in;
      voff[nf] = ok ? (unsigned)((rowi[nf] + tin) * ldb) + (unsigned)lg * (8u * ES) : OOB;
    }
  };
  // a segment is usable by this wave if some position of the tile reads a real input row through it
  // (else it only multiplies zero padding: skipped together with its weights) and the wave owns a chunk
  auto advance = [&]() -> bool {
    for (int s = s_cur + 1; s < h.nseg; ++s) {
      const int sb = s ? a.seg[s - 1].gend : 0, se = a.seg[s].gend;
      const int lo = sb > g0 ? sb : g0, hi = se < g1 ? se : g1;
      const int sh = a.seg[s].shift;
      if ((tmax + sh >= 0) && (tmin + sh < h.L_in) && lo + wk < hi) {
        s_cur = s;
        enter(a.seg[s].x, a.seg[s].nbytes, a.seg[s].ldb, sh, sb, lo, hi);
        return true;
      }
    }
    return false;
  };
  // first usable segment among the four preloaded ones: branch-free, no dependent scalar load
  bool any;
  {
    int pick = -1, p_sb = 0, p_lo = 0, p_hi = 0, p_sh = 0, p_ldb = 0;
    unsigned p_nb = 0;
    const void* p_x = h.w;
#pragma unroll
    for (int s = 3; s >= 0; --s) {
      const int sb = s ? h.first[s - 1].gend : 0, se = h.first[s].gend;
      const int lo = sb > g0 ? sb : g0, hi = se < g1 ? se : g1;
      const int sh = h.first[s].shift;
      const bool use = (tmax + sh >= 0) && (tmin + sh < h.L_in) && (lo + wk < hi);
      pick = use ? s : pick;
      p_sb = use ? sb : p_sb;
      p_lo = use ? lo : p_lo;
      p_hi = use ? hi : p_hi;
      p_sh = use ? sh : p_sh;
      p_ldb = use ? h.first[s].ldb : p_ldb;
      p_nb = use ? h.first[s].nbytes : p_nb;
      p_x = use ? h.first[s].x : p_x;
    }
    if (pick >= 0) {
      s_cur = pick;
      enter(p_x, p_nb, p_ldb, p_sh, p_sb, p_lo, p_hi);
      any = true;
    } else {
      s_cur = 3;
      any = advance();
    }
  }
  auto issue = [&](Frag& fa, Frag(&fb)[NF]) {
    bload<JEN1_W_AUX>(fa, rw, voffA, soffA);
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) bload<JEN1_X_AUX>(fb[nf], rx, voff[nf], soffB);
    if (!parked) {
      ++issued;
      cur_g += SG_NW;
      if (cur_g < cur_hi) {
        soffA += (unsigned)SG_NW * BLK * (unsigned)h.MT;
        soffB += (unsigned)SG_NW * CHB;
      } else if (!advance()) {
        // past the end: the ring keeps issuing (the load count per slot must stay fixed for vmcnt),
        // but with out-of-range offsets -- no memory traffic
        parked = true;
        total = issued;
        voffA = OOB;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) voff[nf] = OOB;
      }
    }
  };

  // ---- epilogue operands: requested right after the ring is filled, used at the very end.
  // Absent operands have a zero-length descriptor (loads return 0): no branches, one kernarg batch.
  const bool owner = (wk == 0);
  bool okk[NF];
  int yrow[NF];
  int co = 0;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, lnu = {0.f, 0.f, 0.f, 0.f};
  float rr[NF][4];
  float rsc[NF];
  float2 lnrs[NF];
  EpiArgs e;
  auto request_epilogue_operands = [&]() {
    const __amdgpu_buffer_rsrc_t r_bias = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(e.bias), 0, (int)e.bias_bytes, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t r_lnu = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(e.ln_u), 0, (int)e.lnu_bytes, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(e.residual), 0,
                                                                           (h.mt_split == 0 || low_m) ? (int)e.res_bytes : 0, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t r_rsc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(e.row_scale), 0, (int)e.rsc_bytes, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t r_lnrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(e.ln_rowstats), 0, (int)e.lnrs_bytes, RSRC_FLAGS);
    const int m = mt * 16 + lg * 4;
    int ph = 0;
    for (int k = 1; k < e.ps_f; ++k) ph += (m >= k * e.out_C) ? 1 : 0;
    co = m - ph * e.out_C;
    bias4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_bias, (unsigned)co * 4u, 0, 0));
    lnu = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_lnu, (unsigned)m * 4u, 0, 0));
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const int ty = (t0 + n_t[nf]) * e.ps_f + ph - e.ps_off;
      okk[nf] = n_ok[nf] && ty >= 0 && ty < e.L_y;
      yrow[nf] = okk[nf] ? (b0 + n_b[nf]) * e.y_brows + e.y_row0 + ty : 0;
      const unsigned roff = okk[nf] ? ((unsigned)yrow[nf] * (unsigned)e.ld_res + (unsigned)co) * ES : OOB;
      if (PRECISE) {
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_res, roff, 0, 0));
#pragma unroll
        for (int r = 0; r < 4; ++r) rr[nf][r] = v[r];
      } else {
        const bf16x4 v = __builtin_bit_cast(bf16x4, __builtin_amdgcn_raw_buffer_load_b64(r_res, roff, 0, 0));
#pragma unroll
        for (int r = 0; r < 4; ++r) rr[nf][r] = (float)v[r];
      }
      const float sc = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_rsc, (unsigned)yrow[nf] * 4u, 0, 0));
      rsc[nf] = e.rsc_bytes ? sc : 1.0f;
      const int irow = n_ok[nf] ? rowi[nf] + t0 + n_t[nf] : 0;
      const u32x2 st = __builtin_amdgcn_raw_buffer_load_b64(r_lnrs, (unsigned)irow * 8u, 0, 0);
      lnrs[nf] = make_float2(__uint_as_float(st.x), __uint_as_float(st.y));
    }
  };

  // ---- main loop ------------------------------------------------------------------------------
  {
    Frag ra[PF], rb[PF][NF];
    if (any) {
#pragma unroll
      for (int u = 0; u < PF; ++u) issue(ra[u], rb[u]);
    } else {
      total = 0;
    }
    SG_STAMP(1);
    e = load_epi_args();
    if (owner) request_epilogue_operands();
    if (e.out_gn_stats) {
      for (int i = tid; i < h.nb * e.ngrp * 2; i += 64 * SG_NW) st_lds[i] = 0.f;
    }
    SG_STAMP(2);
    if (any) {
      for (int c = 0; c < total; c += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
          if (c + u < total) {
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) mma(acc[nf], ra[u], rb[u][nf]);
          }
#ifdef JEN1_PROFILE
          if (c == 0 && u == 0) { if (acc[0][0] == 12345.f) __builtin_amdgcn_s_sleep(1); SG_STAMP(3); }
#endif
          issue(ra[u], rb[u]);
        }
      }
    }
  }
  SG_STAMP(4);

  // ---- K reduction across the 4 waves ---------------------------------------------------------
  if (wk > 0) {
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
      *reinterpret_cast<float4*>(red + ((wk - 1) * NF + nf) * 256 + lane * 4) = make_float4(acc[nf][0], acc[nf][1], acc[nf][2], acc[nf][3]);
  }
  __syncthreads();
  if (owner) {
#pragma unroll
    for (int w2 = 0; w2 < SG_NW - 1; ++w2)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const float4 o = *reinterpret_cast<const float4*>(red + (w2 * NF + nf) * 256 + lane * 4);
        acc[nf][0] += o.x; acc[nf][1] += o.y; acc[nf][2] += o.z; acc[nf][3] += o.w;
      }
  }

  // ---- inter-workgroup split-K: write-through partial slab, ticket, last arriver reduces -------
  if (a.splitk > 1) {
    typedef unsigned long long u64;
    const unsigned tile_id = blockIdx.y * gridDim.x + blockIdx.x;
    constexpr int SLAB = NF * 256;
    float* slab = a.slab + ((size_t)tile_id * a.splitk) * (size_t)SLAB;
    float* mine = slab + (size_t)z * SLAB;
    if (owner) {
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        u64* p = reinterpret_cast<u64*>(mine + nf * 256 + lane * 4);
        const u64 lo = ((u64)__float_as_uint(acc[nf][1]) << 32) | __float_as_uint(acc[nf][0]);
        const u64 hi = ((u64)__float_as_uint(acc[nf][3]) << 32) | __float_as_uint(acc[nf][2]);
        __hip_atomic_store(p, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0)
      misc[0] = (int)__hip_atomic_fetch_add(a.counters + tile_id, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int ticket = misc[0];
    if (ticket != a.splitk - 1) return;
    if (tid == 0) __hip_atomic_store(a.counters + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (owner) {
      for (int zz = 0; zz < a.splitk; ++zz) {
        if (zz == z) continue;
        float* other = slab + (size_t)zz * SLAB;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          u64* p = reinterpret_cast<u64*>(other + nf * 256 + lane * 4);
          const u64 lo = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const u64 hi = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          acc[nf][0] += __uint_as_float((unsigned)lo);
          acc[nf][1] += __uint_as_float((unsigned)(lo >> 32));
          acc[nf][2] += __uint_as_float((unsigned)hi);
          acc[nf][3] += __uint_as_float((unsigned)(hi >> 32));
        }
      }
    }
  }
  SG_STAMP(5);

  // ---- epilogue (wave 0): bias / folded LayerNorm, GELU, residual, row mask, store, statistics ----
  if (owner) {
    T* yT = reinterpret_cast<T*>(e.y);
    float* yF = reinterpret_cast<float*>(e.y);
    // fine groups of the statistics this 16-row tile can touch: fg0 .. fg0 + ngrp - 1
    const int co_tile = co - lg * 4;                                   // first output channel of the tile
    const int fg0 = (int)(((float)co_tile + 0.5f) * e.inv_cpf);
    float gs[2] = {0.f, 0.f}, gq[2] = {0.f, 0.f};
    int rel[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) rel[p] = (int)(((float)(co + 2 * p) + 0.5f) * e.inv_cpf) - fg0;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      float v[4];
      if (e.ln_fold) {
        // Linear(LayerNorm(x)) = rstd * (W'x - mean * rowsum(W')) + W beta   (blocks.py:427-429)
        const float mean = lnrs[nf].x * e.inv_lnC;
        float var = lnrs[nf].y * e.inv_lnC - mean * mean;
        var = var < 0.f ? 0.f : var;
        const float rstd = PRECISE ? 1.0f / sqrtf(var + e.ln_eps) : rsqrtf(var + e.ln_eps);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (acc[nf][r] - mean * lnu[r]) * rstd + bias4[r];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[nf][r] + bias4[r];
      }
      if (e.act == JEN1_ACT_GELU && !low_m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
      }
      float s2 = 0.f, q2 = 0.f;
      if (okk[nf]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (v[r] + rr[nf][r]) * rsc[nf];
        const size_t off = (size_t)((unsigned)yrow[nf] * (unsigned)e.ld_y + (unsigned)co);
        if (e.y_f32) store4(yF + off, v);
        else store4(yT + off, v);
        const float s01 = v[0] + v[1], s23 = v[2] + v[3];
        const float q01 = v[0] * v[0] + v[1] * v[1], q23 = v[2] * v[2] + v[3] * v[3];
        s2 = s01 + s23;
        q2 = q01 + q23;
        if (e.out_gn_stats) {
          if (h.nb == 1) {
            gs[0] += s01; gq[0] += q01;
            gs[1] += s23; gq[1] += q23;
          } else {
            float* sl = st_lds + n_b[nf] * e.ngrp * 2;
            atomicAdd(sl + rel[0] * 2, s01);
            atomicAdd(sl + rel[0] * 2 + 1, q01);
            atomicAdd(sl + rel[1] * 2, s23);
            atomicAdd(sl + rel[1] * 2 + 1, q23);
          }
        }
      }
      if (e.out_rowstats && (h.mt_split == 0 || low_m)) {
        s2 += __shfl_xor(s2, 16); q2 += __shfl_xor(q2, 16);
        s2 += __shfl_xor(s2, 32); q2 += __shfl_xor(q2, 32);
        if (lg == 0 && okk[nf]) {
          unsafeAtomicAdd(e.out_rowstats + 2 * yrow[nf], s2);
          unsafeAtomicAdd(e.out_rowstats + 2 * yrow[nf] + 1, q2);
        }
      }
    }
    if (e.out_gn_stats && h.nb == 1) {
      // all columns belong to one batch element: reduce across the 16 columns of the fragment
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        float s = gs[p], q = gq[p];
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
          s += __shfl_xor(s, off);
          q += __shfl_xor(q, off);
        }
        if (li == 0) {
          atomicAdd(st_lds + rel[p] * 2, s);
          atomicAdd(st_lds + rel[p] * 2 + 1, q);
        }
      }
    }
    SG_STAMP(6);
    if (e.out_gn_stats) {
      // (the LDS atomics above are this wave's own: the LDS pipeline is in order, no barrier needed)
      const int per_b = e.ngrp * 2;
      const float inv_per_b = 1.0f / (float)per_b;
      for (int i = lane; i < h.nb * per_b; i += 64) {
        const int bl = (int)(((float)i + 0.5f) * inv_per_b), el = i - bl * per_b;
        const int b = b0 + bl;
        const float v = st_lds[i];
        const int fg = fg0 + (el >> 1);
        if (b < h.B && fg < JEN1_FINE_GROUPS && v != 0.f) unsafeAtomicAdd(e.out_gn_stats + (size_t)b * 64 + fg * 2 + (el & 1), v);
      }
    }
  }
  SG_STAMP(7);
}

template <typename T, int NF, int PF>
int launch_stream(const StreamArgs& sa, hipStream_t s) {
  const int tiles_b = (sa.hot.B + sa.hot.nb - 1) / sa.hot.nb;
  dim3 grid(sa.hot.MT, sa.hot.tiles_t * tiles_b, sa.splitk);
  const size_t lds = (size_t)(4 + (SG_NW - 1) * NF * 256 + sa.hot.nb * sa.epi.ngrp * 2) * sizeof(float);
  hipLaunchKernelGGL((stream_gemm_kernel<T, NF, PF>), grid, dim3(64 * SG_NW), lds, s, sa);
  JEN1_HIP(hipGetLastError());
  return 0;
}

}  // namespace

// prefetch ring depth (slots of one weight fragment + NF activation fragments) per tile shape
#ifndef JEN1_PF_B16
#define JEN1_PF_B16 8
#endif
#ifndef JEN1_PF_B32
#define JEN1_PF_B32 8
#endif
#ifndef JEN1_PF_B64
#define JEN1_PF_B64 6
#endif
#ifndef JEN1_PF_F16
#define JEN1_PF_F16 8
#endif
#ifndef JEN1_PF_F32
#define JEN1_PF_F32 5
#endif
#ifndef JEN1_PF_F64
#define JEN1_PF_F64 3
#endif

// called by jen1_conv_gemm for args->direct (validated there)
int jen1_stream_gemm_launch(const jen1_conv_args& a, void* stream) {
  StreamArgs sa;
  memset(&sa, 0, sizeof(sa));
  HotArgs& h = sa.hot;
  EpiArgs& e = sa.epi;
  const int es = a.dtype == JEN1_F32 ? 4 : 2;
  sa.slab = a.slab; sa.counters = a.counters; sa.splitk = a.splitk;
#ifdef JEN1_PROFILE
  sa.dbg = (a.splitk == 1) ? reinterpret_cast<unsigned long long*>(a.slab) : nullptr;
#endif
  // ---- K segments ------------------------------------------------------------------------------
  const int64_t rows_in = (int64_t)a.B * a.L_in;
  int G = 0, ns = 0;
  if (a.nseg > 0) {
    for (int s = 0; s < a.nseg; ++s) {
      const jen1_conv_seg& g = a.seg[s];
      G += g.kch;
      const int64_t nb = rows_in * g.ld * es;
      JEN1_CHECK(nb < (int64_t)OOB, "conv_gemm: segment %d too large for 31-bit offsets", s);
      sa.seg[ns++] = SegDev{g.x, (uint32_t)nb, g.ld * es, g.shift, G};
    }
  } else {
    for (int tap = 0; tap < a.taps; ++tap) {
      for (int src = 0; src < 2; ++src) {
        const int c = src ? a.c1 : a.c0;
        if (!c) continue;
        JEN1_CHECK(ns < JEN1_MAX_SEG, "conv_gemm: direct mode supports at most %d (tap, source) segments", JEN1_MAX_SEG);
        const int ld = src ? a.ld1 : a.ld0;
        const int64_t nb = rows_in * ld * es;
        JEN1_CHECK(nb < (int64_t)OOB, "conv_gemm: source too large for 31-bit offsets");
        G += c / 32;
        sa.seg[ns++] = SegDev{src ? a.x1 : a.x0, (uint32_t)nb, ld * es, tap - a.pad_left, G};
      }
    }
  }
  for (int s = 0; s < 4; ++s) {
    if (s < ns) h.first[s] = sa.seg[s];
    else h.first[s] = SegDev{a.w, 0u, 0, 0, G};      // empty: [G, G) holds no chunk
  }
  h.nseg = ns;
  h.G = G;
  h.cps = (G + a.splitk - 1) / a.splitk;
  JEN1_CHECK((a.splitk - 1) * h.cps < G, "conv_gemm: splitk %d leaves an empty K slice (%d chunks)", a.splitk, G);
  h.MT = a.M / 16;
  const int64_t wb = (int64_t)G * h.MT * 512 * es;
  JEN1_CHECK(wb < (int64_t)OOB, "conv_gemm: packed weight too large for 31-bit offsets");
  h.w = a.w;
  h.w_bytes = (uint32_t)wb;
  h.B = a.B; h.L_in = a.L_in; h.L_out = a.L_out; h.stride = a.stride;
  h.tb = a.tb; h.nb = a.nb;
  h.tiles_t = (a.L_out + a.tb - 1) / a.tb;
  h.inv_tiles_t = 1.0f / (float)h.tiles_t;
  h.inv_tb = 1.0f / (float)a.tb;
  h.mt_split = a.m_split / 16;
  h.g_split = a.m_split ? a.k_split : 0;
  // ---- epilogue ----------------------------------------------------------------------------------
  const int64_t y_rows = (int64_t)a.B * a.y_brows;
  JEN1_CHECK(y_rows * a.ld_y * 4 < (int64_t)OOB && (!a.residual || y_rows * a.ld_res * es < (int64_t)OOB),
             "conv_gemm: output too large for 31-bit offsets");
  e.bias = a.bias; e.bias_bytes = a.bias ? (uint32_t)a.out_C * 4u : 0u;
  e.residual = a.residual; e.res_bytes = a.residual ? (uint32_t)(y_rows * a.ld_res * es) : 0u;
  e.row_scale = a.row_scale; e.rsc_bytes = a.row_scale ? (uint32_t)(y_rows * 4) : 0u;
  e.ln_u = a.ln_fold ? a.ln_u : nullptr; e.lnu_bytes = a.ln_fold ? (uint32_t)a.M * 4u : 0u;
  e.ln_rowstats = a.ln_fold ? a.ln_rowstats : nullptr; e.lnrs_bytes = a.ln_fold ? (uint32_t)(rows_in * 8) : 0u;
  e.y = a.y; e.out_gn_stats = a.out_gn_stats; e.out_rowstats = a.out_rowstats;
  e.out_C = a.out_C; e.ps_f = a.ps_f; e.ps_off = a.ps_off; e.L_y = a.L_y; e.y_brows = a.y_brows;
  e.y_row0 = a.y_row0; e.ld_y = a.ld_y; e.ld_res = a.ld_res; e.y_f32 = a.y_f32; e.act = a.act;
  e.ln_fold = a.ln_fold;
  e.ngrp = (!a.out_gn_stats || a.out_cpf >= 16) ? 2 : 16 / a.out_cpf + 1;
  e.inv_cpf = a.out_gn_stats ? 1.0f / (float)a.out_cpf : 1.0f;
  e.ln_eps = a.ln_eps;
  e.inv_lnC = a.ln_fold ? 1.0f / (float)a.ln_C : 0.f;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (a.dtype == JEN1_F32) {
    switch (a.cfg) {
      case JEN1_CFG_S16x64: return launch_stream<float, 4, JEN1_PF_F64>(sa, s);
      case JEN1_CFG_S16x32: return launch_stream<float, 2, JEN1_PF_F32>(sa, s);
      case JEN1_CFG_S16x16: return launch_stream<float, 1, JEN1_PF_F16>(sa, s);
    }
  } else {
    switch (a.cfg) {
      case JEN1_CFG_S16x64: return launch_stream<bf16_t, 4, JEN1_PF_B64>(sa, s);
      case JEN1_CFG_S16x32: return launch_stream<bf16_t, 2, JEN1_PF_B32>(sa, s);
      case JEN1_CFG_S16x16: return launch_stream<bf16_t, 1, JEN1_PF_B16>(sa, s);
    }
  }
  return jen1_set_error("jen1_conv_gemm: direct mode needs a streaming (S16) cfg, got %d", a.cfg);
}
