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

struct StreamArgs {
  const void* w;
  const float* bias;
  const void* residual;
  void* y;
  const float* ln_rowstats;
  const float* ln_u;
  const float* row_scale;
  float* out_gn_stats;
  float* out_rowstats;
  float* slab;
  unsigned* counters;
  unsigned long long* dbg;
  SegDev seg[JEN1_MAX_SEG];
  uint32_t w_bytes;
  int32_t nseg, G, cps, splitk;
  int32_t B, L_in, L_out, stride;
  int32_t MT, out_C, ps_f, ps_off, L_y, y_brows, y_row0, ld_y, ld_res, y_f32, act, tb, nb, tiles_t;
  float inv_tiles_t, inv_tb, inv_cpf, ln_eps, inv_lnC;
  int32_t ln_fold;
  int32_t ngrp;       // statistics fine groups a 16-channel tile can touch (2 when out_cpf >= 16)
};

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

#ifndef JEN1_W_AUX
#define JEN1_W_AUX 0      // cache policy of the weight stream (2 = nt)
#endif

template <typename T, int NF, int PF>
__global__ __launch_bounds__(256) void stream_gemm_kernel(const StreamArgs a) {
  typedef typename Frag8<T>::type Frag;
  constexpr bool PRECISE = is_f32<T>::value;
  constexpr unsigned ES = sizeof(T);
  constexpr unsigned BLK = 512 * ES;      // bytes of one (chunk, m-tile) weight block
  constexpr unsigned CHB = 32 * ES;       // bytes of one 32-channel chunk of an activation row
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* misc = reinterpret_cast<int*>(smem);
  float* red = smem + 4;                   // [3][NF][256]
  float* st_lds = red + 3 * NF * 256;      // [nb][ngrp fine groups][2]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wk = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
  SG_STAMP(0);

  // ---- tile coordinates (all wave-uniform) ---------------------------------------------------
  const int by = blockIdx.y;
  const int bt = (int)(((float)by + 0.5f) * a.inv_tiles_t), tt = by - bt * a.tiles_t;
  const int b0 = bt * a.nb, t0 = tt * a.tb;
  const int mt = blockIdx.x, z = blockIdx.z;
  const int g0 = z * a.cps;
  const int g1 = (g0 + a.cps < a.G) ? g0 + a.cps : a.G;
  const int t_last = ((t0 + a.tb < a.L_out) ? t0 + a.tb : a.L_out) - 1;
  const int n_rows = a.nb * a.tb;

  // ---- this lane's activation rows (one per 16-column fragment) ------------------------------
  int rowi[NF], tpos[NF], n_b[NF], n_t[NF];
  bool n_ok[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int n = nf * 16 + li;
    const int bl = (int)(((float)n + 0.5f) * a.inv_tb), tl = n - bl * a.tb;     // exact for n < 64
    n_ok[nf] = (n < n_rows) && (b0 + bl < a.B) && (t0 + tl < a.L_out);
    n_b[nf] = bl;
    n_t[nf] = tl;
    rowi[nf] = (b0 + bl) * a.L_in;
    tpos[nf] = (t0 + tl) * a.stride;
  }

  f32x4 acc[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) acc[nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- K cursor of this wave: chunk cur_g of segment s_cur, chunks lo+wk, lo+wk+4, ... ----------
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, (int)a.w_bytes, RSRC_FLAGS);
  __amdgpu_buffer_rsrc_t rx = rw;
  unsigned voffA = (unsigned)lane * (8u * ES);
  unsigned voff[NF];
  unsigned soffA = 0, soffB = 0;
  int s_cur = -1, cur_g = 0, cur_hi = 0;
  int issued = 0, total = 0x7fffffff;
  bool parked = false;
  auto advance = [&]() -> bool {
    for (int s = s_cur + 1; s < a.nseg; ++s) {
      const int sb = s ? a.seg[s - 1].gend : 0, se = a.seg[s].gend;
      const int lo = sb > g0 ? sb : g0, hi = se < g1 ? se : g1;
      const int sh = a.seg[s].shift;
      // live: some position of this tile reads a real input row through this segment
      const bool live = (t_last * a.stride + sh >= 0) && (t0 * a.stride + sh < a.L_in);
      if (live && lo + wk < hi) {
        s_cur = s;
        cur_g = lo + wk;
        cur_hi = hi;
        soffA = (unsigned)(cur_g * a.MT + mt) * BLK;
        soffB = (unsigned)(cur_g - sb) * CHB;
        rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.seg[s].x), 0, (int)a.seg[s].nbytes, RSRC_FLAGS);
        const int ldb = a.seg[s].ldb;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          const int tin = tpos[nf] + sh;
          const bool ok = n_ok[nf] && tin >= 0 && tin < a.L_in;
          voff[nf] = ok ? (unsigned)((rowi[nf] + tin) * ldb) + (unsigned)lg * (8u * ES) : OOB;
        }
        return true;
      }
    }
    return false;
  };
  auto issue = [&](Frag& fa, Frag(&fb)[NF]) {
    bload<JEN1_W_AUX>(fa, rw, voffA, soffA);
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) bload<0>(fb[nf], rx, voff[nf], soffB);
    if (!parked) {
      ++issued;
      cur_g += 4;
      if (cur_g < cur_hi) {
        soffA += 4u * BLK * (unsigned)a.MT;
        soffB += 4u * CHB;
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

  // ---- epilogue operands: requested right after the ring is filled, used at the very end -----
  const bool owner = (wk == 0);
  const T* res = reinterpret_cast<const T*>(a.residual);
  bool okk[NF];
  int yrow[NF];
  int co = 0;
  float bias4[4] = {0.f, 0.f, 0.f, 0.f};
  float rr[NF][4];
  float rsc[NF];
  float lnu[4] = {0.f, 0.f, 0.f, 0.f};
  float2 lnrs[NF];
  auto request_epilogue_operands = [&]() {
    const int m = mt * 16 + lg * 4;
    int ph = 0;
    for (int k = 1; k < a.ps_f; ++k) ph += (m >= k * a.out_C) ? 1 : 0;
    co = m - ph * a.out_C;
    if (a.bias) {
      const float4 bb = *reinterpret_cast<const float4*>(a.bias + co);
      bias4[0] = bb.x; bias4[1] = bb.y; bias4[2] = bb.z; bias4[3] = bb.w;
    }
    if (a.ln_fold) {
      const float4 uu = *reinterpret_cast<const float4*>(a.ln_u + m);
      lnu[0] = uu.x; lnu[1] = uu.y; lnu[2] = uu.z; lnu[3] = uu.w;
    }
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const int ty = (t0 + n_t[nf]) * a.ps_f + ph - a.ps_off;
      okk[nf] = n_ok[nf] && ty >= 0 && ty < a.L_y;
      yrow[nf] = okk[nf] ? (b0 + n_b[nf]) * a.y_brows + a.y_row0 + ty : 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) rr[nf][r] = 0.f;
      if (res) load4(res + (size_t)((unsigned)yrow[nf] * (unsigned)a.ld_res + (unsigned)co), rr[nf]);
      rsc[nf] = a.row_scale ? a.row_scale[yrow[nf]] : 1.0f;
      if (a.ln_fold) {
        const int irow = n_ok[nf] ? rowi[nf] + t0 + n_t[nf] : 0;
        lnrs[nf] = *reinterpret_cast<const float2*>(a.ln_rowstats + 2 * irow);
      }
    }
  };

  // ---- main loop ------------------------------------------------------------------------------
  {
    Frag ra[PF], rb[PF][NF];
    const bool any = advance();
    if (any) {
#pragma unroll
      for (int u = 0; u < PF; ++u) issue(ra[u], rb[u]);
    } else {
      total = 0;
    }
    SG_STAMP(1);
    if (owner) request_epilogue_operands();
    if (a.out_gn_stats) {
      for (int i = tid; i < a.nb * a.ngrp * 2; i += 256) st_lds[i] = 0.f;
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
    for (int w2 = 0; w2 < 3; ++w2)
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
    T* yT = reinterpret_cast<T*>(a.y);
    float* yF = reinterpret_cast<float*>(a.y);
    // fine groups of the statistics this 16-row tile can touch: fg0 .. fg0 + ngrp - 1
    const int co_tile = co - lg * 4;                                   // first output channel of the tile
    const int fg0 = (int)(((float)co_tile + 0.5f) * a.inv_cpf);
    float gs[2] = {0.f, 0.f}, gq[2] = {0.f, 0.f};
    int rel[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) rel[p] = (int)(((float)(co + 2 * p) + 0.5f) * a.inv_cpf) - fg0;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      float v[4];
      if (a.ln_fold) {
        // Linear(LayerNorm(x)) = rstd * (W'x - mean * rowsum(W')) + W beta   (blocks.py:427-429)
        const float mean = lnrs[nf].x * a.inv_lnC;
        float var = lnrs[nf].y * a.inv_lnC - mean * mean;
        var = var < 0.f ? 0.f : var;
        const float rstd = PRECISE ? 1.0f / sqrtf(var + a.ln_eps) : rsqrtf(var + a.ln_eps);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (acc[nf][r] - mean * lnu[r]) * rstd + bias4[r];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[nf][r] + bias4[r];
      }
      if (a.act == JEN1_ACT_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
      }
      float s2 = 0.f, q2 = 0.f;
      if (okk[nf]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (v[r] + rr[nf][r]) * rsc[nf];
        const size_t off = (size_t)((unsigned)yrow[nf] * (unsigned)a.ld_y + (unsigned)co);
        if (a.y_f32) store4(yF + off, v);
        else store4(yT + off, v);
        const float s01 = v[0] + v[1], s23 = v[2] + v[3];
        const float q01 = v[0] * v[0] + v[1] * v[1], q23 = v[2] * v[2] + v[3] * v[3];
        s2 = s01 + s23;
        q2 = q01 + q23;
        if (a.out_gn_stats) {
          if (a.nb == 1) {
            gs[0] += s01; gq[0] += q01;
            gs[1] += s23; gq[1] += q23;
          } else {
            float* sl = st_lds + n_b[nf] * a.ngrp * 2;
            atomicAdd(sl + rel[0] * 2, s01);
            atomicAdd(sl + rel[0] * 2 + 1, q01);
            atomicAdd(sl + rel[1] * 2, s23);
            atomicAdd(sl + rel[1] * 2 + 1, q23);
          }
        }
      }
      if (a.out_rowstats) {
        s2 += __shfl_xor(s2, 16); q2 += __shfl_xor(q2, 16);
        s2 += __shfl_xor(s2, 32); q2 += __shfl_xor(q2, 32);
        if (lg == 0 && okk[nf]) {
          unsafeAtomicAdd(a.out_rowstats + 2 * yrow[nf], s2);
          unsafeAtomicAdd(a.out_rowstats + 2 * yrow[nf] + 1, q2);
        }
      }
    }
    if (a.out_gn_stats && a.nb == 1) {
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
    if (a.out_gn_stats) {
      // (the LDS atomics above are this wave's own: the LDS pipeline is in order, no barrier needed)
      const int per_b = a.ngrp * 2;
      const float inv_per_b = 1.0f / (float)per_b;
      for (int i = lane; i < a.nb * per_b; i += 64) {
        const int bl = (int)(((float)i + 0.5f) * inv_per_b), e = i - bl * per_b;
        const int b = b0 + bl;
        const float v = st_lds[i];
        const int fg = fg0 + (e >> 1);
        if (b < a.B && fg < JEN1_FINE_GROUPS && v != 0.f) unsafeAtomicAdd(a.out_gn_stats + (size_t)b * 64 + fg * 2 + (e & 1), v);
      }
    }
  }
  SG_STAMP(7);
}

template <typename T, int NF, int PF>
int launch_stream(const StreamArgs& sa, hipStream_t s) {
  const int tiles_b = (sa.B + sa.nb - 1) / sa.nb;
  dim3 grid(sa.MT, sa.tiles_t * tiles_b, sa.splitk);
  const size_t lds = (size_t)(4 + 3 * NF * 256 + sa.nb * sa.ngrp * 2) * sizeof(float);
  hipLaunchKernelGGL((stream_gemm_kernel<T, NF, PF>), grid, dim3(256), lds, s, sa);
  JEN1_HIP(hipGetLastError());
  return 0;
}

}  // namespace

// prefetch ring depth (slots of one weight fragment + NF activation fragments) per tile shape
#ifndef JEN1_PF_B16
#define JEN1_PF_B16 12
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
  const int es = a.dtype == JEN1_F32 ? 4 : 2;
  sa.w = a.w; sa.bias = a.bias; sa.residual = a.residual; sa.y = a.y;
  sa.ln_rowstats = a.ln_rowstats; sa.ln_u = a.ln_u; sa.row_scale = a.row_scale;
  sa.out_gn_stats = a.out_gn_stats; sa.out_rowstats = a.out_rowstats;
  sa.slab = a.slab; sa.counters = a.counters;
#ifdef JEN1_PROFILE
  sa.dbg = (a.splitk == 1) ? reinterpret_cast<unsigned long long*>(a.slab) : nullptr;
#endif
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
  sa.nseg = ns;
  sa.G = G;
  sa.splitk = a.splitk;
  sa.cps = (G + a.splitk - 1) / a.splitk;
  JEN1_CHECK((a.splitk - 1) * sa.cps < G, "conv_gemm: splitk %d leaves an empty K slice (%d chunks)", a.splitk, G);
  sa.MT = a.M / 16;
  const int64_t wb = (int64_t)G * sa.MT * 512 * es;
  JEN1_CHECK(wb < (int64_t)OOB, "conv_gemm: packed weight too large for 31-bit offsets");
  sa.w_bytes = (uint32_t)wb;
  sa.B = a.B; sa.L_in = a.L_in; sa.L_out = a.L_out; sa.stride = a.stride;
  sa.out_C = a.out_C; sa.ps_f = a.ps_f; sa.ps_off = a.ps_off; sa.L_y = a.L_y; sa.y_brows = a.y_brows;
  sa.y_row0 = a.y_row0; sa.ld_y = a.ld_y; sa.ld_res = a.ld_res; sa.y_f32 = a.y_f32; sa.act = a.act;
  sa.tb = a.tb; sa.nb = a.nb;
  sa.tiles_t = (a.L_out + a.tb - 1) / a.tb;
  sa.inv_tiles_t = 1.0f / (float)sa.tiles_t;
  sa.inv_tb = 1.0f / (float)a.tb;
  sa.inv_cpf = a.out_gn_stats ? 1.0f / (float)a.out_cpf : 1.0f;
  sa.ngrp = (!a.out_gn_stats || a.out_cpf >= 16) ? 2 : 16 / a.out_cpf + 1;
  sa.ln_eps = a.ln_eps;
  sa.inv_lnC = a.ln_fold ? 1.0f / (float)a.ln_C : 0.f;
  sa.ln_fold = a.ln_fold;
  JEN1_CHECK((int64_t)a.B * a.y_brows * a.ld_y < (int64_t)1 << 31, "conv_gemm: output too large for 32-bit element offsets");
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
