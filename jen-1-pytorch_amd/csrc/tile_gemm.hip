// Tiled implicit-GEMM convolution for the long levels of the JEN-1 denoiser (T' >= 64: levels 0-2,
// 1500 / 375 / 94 positions) on gfx950 (MI355X).
//
// Same operation as conv_gemm.hip (include/jen1_hip.h: jen1_conv_gemm, reference jen1/model/blocks.py
// _Conv1d :34-53, Upsample1d :69-95, ConvBlock1d :137-145, ResnetBlock1d :219-231, skip concat :732-734),
// restricted to what these levels need -- GroupNorm(+FiLM)(+SiLU) or no prologue, k <= 9 taps with stride,
// the sub-pixel form of ConvTranspose1d, bias / residual / statistics epilogue -- and written for the regime
// measured on the chip: a launch is a few microseconds, every workgroup is resident at once, so a workgroup's
// time is its dynamic instruction count plus its serial memory round trips.
//   * one workgroup = ALL output channels of a tile when M <= 256 (BM = 64 * MF): each activation row is
//     normalised / activated exactly once per launch;
//   * kernel arguments arrive in two explicit scalar-load batches; the weight ring, the first staging batch
//     and everything the GroupNorm tables need (gamma, beta, FiLM rows, the producers' fine-group sums) are
//     requested before the first wait;
//   * every thread derives the affine pair (A, S) of its own channel -- y = silu?(A x + S) folds GroupNorm,
//     FiLM and the skip scale -- straight from the fine-group sums (no group loop, no extra barrier);
//   * the activation tile (+ conv halo) is staged once in LDS, taps are row-shifted views of it; weights
//     stream from L2 through a register ring with scalar offsets;
//   * 4 waves x (16 MF) output rows x (16 NF) positions, v_mfma_f32_16x16x32_bf16 / 16x16x4_f32.
#include "common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x16 __attribute__((ext_vector_type(16)));

struct TileHot {                 // 48 dwords: read before the first load is issued
  const void* x0;
  const void* x1;
  const void* w;
  const float* st0;
  const float* st1;
  const float* gamma;
  const float* beta;
  const float* film;             // already offset by film_off
  const int32_t* film_step;
  const int32_t* film_row;       // 20 dwords of pointers
  uint32_t w_bytes;
  int32_t B, L_in, L_out, c0, c1, ld0, ld1, taps, stride, pad_left, MT, tb, tiles_t, pro_mode, groups, cpg, film_C, film_ld,
      nfg0, nfg1;                // 21 dwords
  float inv_tiles_t, eps, src1_scale, inv_count, inv_cpg, inv_cpf0, inv_cpf1;     // 7 dwords
};
struct TileEpi {                 // 32 dwords: read while the first loads are in flight
  void* y;
  const void* residual;
  const float* bias;
  float* out_gn_stats;           // 8 dwords
  uint32_t res_bytes, bias_bytes;
  int32_t M, out_C, ps_f, ps_off, L_y, y_brows, y_row0, ld_y, ld_res, y_f32, out_cpf;
  float inv_out_cpf;
  int32_t pad[2];
  // raw extra K segments behind the taps (a 1x1 shortcut over the block's input riding on its second conv, blocks.py:229-231):
  // channels [c0 + c1, +c2) of the staged tile come from x2, the next c3 from x3, both at row shift 0.  These 8 dwords sit at
  // kernarg offset 0x120 and are read with the hot block (TileExtra), the rest of this block only before the epilogue.
  const void* x2;
  const void* x3;
  int32_t ld2, c2, ld3, c3;
};
struct TileExtra {
  const void* x2;
  const void* x3;
  int32_t ld2, c2, ld3, c3;
};
struct TileArgs {
  TileHot hot;
  TileEpi epi;
};
static_assert(sizeof(TileHot) == 192 && sizeof(TileEpi) == 128 && offsetof(TileArgs, epi) == 192, "kernarg blocks are read with fixed-size scalar loads");
static_assert(offsetof(TileArgs, epi) + offsetof(TileEpi, x2) == 0x120 && sizeof(TileExtra) == 32, "the extra-segment block is read at a fixed kernarg offset");

// the hot block and the extra-segment block in ONE batch of scalar loads (one wait)
__device__ __forceinline__ TileHot load_tile_hot(TileExtra& ex) {
  const auto kp = __builtin_amdgcn_kernarg_segment_ptr();
  typedef unsigned int u32x8 __attribute__((ext_vector_type(8)));
  u32x16 k0, k1, k2;
  u32x8 k3;
  unsigned t0, t1;
  asm volatile("s_load_dwordx16 %0, %6, 0x0\n\ts_load_dwordx16 %1, %6, 0x40\n\ts_load_dwordx16 %2, %6, 0x80\n\t"
               "s_load_dwordx8 %5, %6, 0x120\n\t"
               "s_load_dword %3, %6, 0xc0\n\ts_load_dword %4, %6, 0x100\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(k0), "=&s"(k1), "=&s"(k2), "=&s"(t0), "=&s"(t1), "=&s"(k3) : "s"(kp) : "memory");
  struct Raw { unsigned d[48]; } raw;
#pragma unroll
  for (int i = 0; i < 16; ++i) { raw.d[i] = k0[i]; raw.d[16 + i] = k1[i]; raw.d[32 + i] = k2[i]; }
  struct RawX { unsigned d[8]; } rx;
#pragma unroll
  for (int i = 0; i < 8; ++i) rx.d[i] = k3[i];
  ex = __builtin_bit_cast(TileExtra, rx);
  return __builtin_bit_cast(TileHot, raw);
}
__device__ __forceinline__ TileEpi load_tile_epi() {
  const auto kp = __builtin_amdgcn_kernarg_segment_ptr();
  u32x16 k0, k1;
  asm volatile("s_load_dwordx16 %0, %2, 0xc0\n\ts_load_dwordx16 %1, %2, 0x100\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(k0), "=&s"(k1) : "s"(kp) : "memory");
  struct Raw { unsigned d[32]; } raw;
#pragma unroll
  for (int i = 0; i < 16; ++i) { raw.d[i] = k0[i]; raw.d[16 + i] = k1[i]; }
  return __builtin_bit_cast(TileEpi, raw);
}

constexpr unsigned OOB = 0x80000000u;
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
#ifndef JEN1_TILE_W_AUX
#define JEN1_TILE_W_AUX 0      // cache policy of the weight ring (2 = nt)
#endif
__device__ __forceinline__ void bload(bf16x8& f, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  f = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, JEN1_TILE_W_AUX));
}
__device__ __forceinline__ void bload(f32x8& f, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, JEN1_TILE_W_AUX);
  const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(r, voff + 16u, soff, JEN1_TILE_W_AUX);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f.v[j] = __uint_as_float(lo[j]);
    f.v[4 + j] = __uint_as_float(hi[j]);
  }
}
__device__ __forceinline__ void lds_read(bf16x8& f, const bf16_t* p) { f = *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ void lds_read(f32x8& f, const float* p) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  const float4 b = *reinterpret_cast<const float4*>(p + 4);
  f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w;
  f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
}

// sum over the 16 lanes of a DPP row (the 16 positions of an MFMA tile) without touching LDS
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f<0xB1>(v);      // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);      // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);     // row_half_mirror
  v += dpp_f<0x140>(v);     // row_mirror
  return v;
}

constexpr int VB = 4;          // staging vectors (8 channels) per thread per batch

// Tuning builds only (-DJEN1_TILE_PROFILE): thread 0 of every workgroup records the 100 MHz counter at the stages of the kernel:
// dbg[workgroup * 8 + stage]  (jen1_tile_debug_buffer sets the pointer)
#ifdef JEN1_TILE_PROFILE
__device__ unsigned long long* g_tile_dbg = nullptr;
#define TK_STAMP(i) do { tk_t[(i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TK_STAMP(i) do { } while (0)
#endif

template <typename T, int MF, int NF, int PF, bool XS>      // XS: raw extra K segments behind the taps
__global__ __launch_bounds__(256) void tile_gemm_kernel(const TileArgs a_unused) {
  typedef typename Frag8<T>::type Frag;
  typedef typename VecOf<T>::type Vec;
  constexpr bool PRECISE = is_f32<T>::value;
  constexpr unsigned ES = sizeof(T);
  constexpr unsigned BLK = 512 * ES;
  constexpr int BN = 16 * NF;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lg = lane >> 4;
#ifdef JEN1_TILE_PROFILE
  unsigned long long tk_t[8];
#pragma unroll
  for (int i_ = 0; i_ < 8; ++i_) tk_t[i_] = 0;
#endif
  TK_STAMP(0);
  TileExtra ex;
  const TileHot h = load_tile_hot(ex);
  TK_STAMP(1);

  // ---- tile coordinates ---------------------------------------------------------------------------
  const int by = blockIdx.y;
  const int b = (int)(((float)by + 0.5f) * h.inv_tiles_t), tt = by - b * h.tiles_t;
  const int t0 = tt * h.tb;
  const int ctot = h.c0 + h.c1;                                // normalised / main channels
  const int call = XS ? ctot + ex.c2 + ex.c3 : ctot;           // + the raw extra segments
  const int kch = ctot >> 5;
  const int vpr = call >> 3;
  const int ldsld = call + 8;
  const int rows_in = (h.tb - 1) * h.stride + h.taps;
  const int tin0 = t0 * h.stride - h.pad_left;
  const int KS = h.taps * kch + (XS ? (ex.c2 + ex.c3) >> 5 : 0);   // k-steps: (tap, 32-channel chunk), then the extra chunks
  const int mt0 = (blockIdx.x * 4 + wv) * MF;                   // this wave's first 16-row tile of M

  T* tile = reinterpret_cast<T*>(smem);
  float* tabA = reinterpret_cast<float*>(smem + (((size_t)rows_in * ldsld * ES + 15) & ~(size_t)15));
  float* tabS = tabA + ctot;
  float* st_lds = tabS + ctot;                                 // [BM / out_cpf + 1][2] output statistics

  // ---- (1) weight ring: the first PF k-steps ---------------------------------------------------------
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(h.w), 0, (int)h.w_bytes, RSRC_FLAGS);
  unsigned voffA[MF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) voffA[mf] = (mt0 + mf < h.MT) ? (unsigned)(mt0 + mf) * BLK + (unsigned)lane * (8u * ES) : OOB;
  const unsigned stepA = (unsigned)h.MT * BLK;
  unsigned soffA = 0;
  int issuedA = 0;
  Frag ring[PF][MF];
  auto issueA = [&](Frag(&dst)[MF]) {
    const unsigned so = issuedA < KS ? soffA : 0u;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) bload(dst[mf], rw, issuedA < KS ? voffA[mf] : OOB, so);
    soffA += stepA;
    ++issuedA;
  };
#pragma unroll
  for (int u = 0; u < PF; ++u) issueA(ring[u]);

  // ---- (2) first staging batch -------------------------------------------------------------------------
  const float inv_vpr = 1.0f / (float)vpr;
  const int nvec = rows_in * vpr;
  const T* x0p = reinterpret_cast<const T*>(h.x0);
  const T* x1p = reinterpret_cast<const T*>(h.x1);
  struct Batch { Vec x[VB]; };
  auto load_batch = [&](Batch& bt, int v0) {
#pragma unroll
    for (int u = 0; u < VB; ++u) {
      const int v = v0 + u * 256 + tid;
      const int vv = v < nvec ? v : 0;
      const int row = (int)(((float)vv + 0.5f) * inv_vpr), c = (vv - row * vpr) * 8;
      const int tin = tin0 + row;
      const bool ok = v < nvec && tin >= 0 && tin < h.L_in;
      const unsigned grow = (unsigned)(b * h.L_in + (ok ? tin : 0));
      const T* p = (c < h.c0) ? x0p + (size_t)(grow * (unsigned)h.ld0 + (unsigned)c) : x1p + (size_t)(grow * (unsigned)h.ld1 + (unsigned)(c - h.c0));
      if (XS && c >= ctot) {
        const int cx = c - ctot;
        p = (cx < ex.c2) ? reinterpret_cast<const T*>(ex.x2) + (size_t)(grow * (unsigned)ex.ld2 + (unsigned)cx)
                         : reinterpret_cast<const T*>(ex.x3) + (size_t)(grow * (unsigned)ex.ld3 + (unsigned)(cx - ex.c2));
      }
      bt.x[u] = *reinterpret_cast<const Vec*>(p);
    }
  };
  Batch cur;
  load_batch(cur, 0);
  TK_STAMP(2);

  // ---- (3) affine tables of the prologue: y = silu?(A[c] x + S[c]) ---------------------------------------
  const bool gn = h.pro_mode == JEN1_PRO_GN || h.pro_mode == JEN1_PRO_GN_SILU;
  const bool do_silu = h.pro_mode == JEN1_PRO_GN_SILU || h.pro_mode == JEN1_PRO_SILU;
  if (gn) {
    int fr = b;
    if (h.film) fr = h.film_step ? h.film_step[0] : (h.film_row ? h.film_row[b] : b);
    // one group over all channels (Patcher / Unpatcher, blocks.py:251, :279): every channel needs the sum of ALL 32 fine groups -- each
    // wave adds them once with shuffles (32 lanes, one fine group each) instead of every thread walking them (2.2 -> 0.7 us of table time)
    float one_s = 0.f, one_q = 0.f;
    const bool one_group = h.groups == 1 && h.c1 == 0;
    if (one_group) {
      const float2* fine = reinterpret_cast<const float2*>(h.st0 + b * 64);
      const float2 v = (lane < JEN1_FINE_GROUPS) ? fine[lane] : make_float2(0.f, 0.f);
      one_s = v.x;
      one_q = v.y;
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) {
        one_s += __shfl_xor(one_s, off);
        one_q += __shfl_xor(one_q, off);
      }
      one_s = __shfl(one_s, 0);
      one_q = __shfl(one_q, 0);
    }
    for (int c = tid; c < ctot; c += 256) {
      const bool s1 = c >= h.c0;
      const int gch = (int)(((float)c + 0.5f) * h.inv_cpg);
      const int g = gch < h.groups ? gch : h.groups - 1;
      const int lo = g * h.cpg - (s1 ? h.c0 : 0);
      const int nfg = s1 ? h.nfg1 : h.nfg0;
      const int f0 = (int)(((float)lo + 0.5f) * (s1 ? h.inv_cpf1 : h.inv_cpf0));
      const float2* fine = reinterpret_cast<const float2*>((s1 ? h.st1 : h.st0) + b * 64) + f0;
      const float gam = h.gamma[c], bet = h.beta[c];
      float fs = 0.f, fh = 0.f;
      if (h.film) {
        const float* fp = h.film + (size_t)((unsigned)fr * (unsigned)h.film_ld + (unsigned)c);
        fs = fp[0];
        fh = fp[h.film_C];
      }
      float s = one_s, q = one_q;
      for (int k = 0; !one_group && k < nfg && f0 + k < JEN1_FINE_GROUPS; k += 4) {
        float2 t4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) t4[j] = fine[(k + j < nfg && f0 + k + j < JEN1_FINE_GROUPS) ? k + j : 0];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool in = k + j < nfg && f0 + k + j < JEN1_FINE_GROUPS;
          s += in ? t4[j].x : 0.f;
          q += in ? t4[j].y : 0.f;
        }
      }
      const float sc = s1 ? h.src1_scale : 1.0f;
      s *= sc;
      q *= sc * sc;
      const float mean = s * h.inv_count;
      float var = q * h.inv_count - mean * mean;
      var = var < 0.f ? 0.f : var;
      const float rstd = PRECISE ? 1.0f / sqrtf(var + h.eps) : rsqrtf(var + h.eps);
      float A = rstd * gam;
      float S = bet - mean * A;
      A *= sc;
      if (h.film) {
        A *= fs + 1.0f;
        S = S * (fs + 1.0f) + fh;
      }
      tabA[c] = A;
      tabS[c] = S;
    }
  }
  const TileEpi e = load_tile_epi();
  if (e.out_gn_stats) {
    for (int i = tid; i < 2 * (64 * MF / 2 + 2); i += 256) st_lds[i] = 0.f;
  }
  TK_STAMP(3);
  __syncthreads();

  // ---- (4) stage the tile: prologue applied once, zero padding applied after it ------------------------------
  for (int v0 = 0; v0 < nvec; v0 += 256 * VB) {
    Batch nxt;
    const bool more = v0 + 256 * VB < nvec;
    if (more) load_batch(nxt, v0 + 256 * VB);
#pragma unroll
    for (int u = 0; u < VB; ++u) {
      const int v = v0 + u * 256 + tid;
      if (v >= nvec) continue;
      const int row = (int)(((float)v + 0.5f) * inv_vpr), c = (v - row * vpr) * 8;
      const int tin = tin0 + row;
      float x[8];
      vec_to_float(cur.x[u], x);
      if (tin >= 0 && tin < h.L_in) {
        if (XS && c >= ctot) {
          // raw extra segment: staged as it is
        } else if (gn) {
          const float4 a0 = *reinterpret_cast<const float4*>(tabA + c), a1 = *reinterpret_cast<const float4*>(tabA + c + 4);
          const float4 s0 = *reinterpret_cast<const float4*>(tabS + c), s1 = *reinterpret_cast<const float4*>(tabS + c + 4);
          x[0] = x[0] * a0.x + s0.x; x[1] = x[1] * a0.y + s0.y; x[2] = x[2] * a0.z + s0.z; x[3] = x[3] * a0.w + s0.w;
          x[4] = x[4] * a1.x + s1.x; x[5] = x[5] * a1.y + s1.y; x[6] = x[6] * a1.z + s1.z; x[7] = x[7] * a1.w + s1.w;
        } else if (c >= h.c0 && h.src1_scale != 1.0f) {
#pragma unroll
          for (int j = 0; j < 8; ++j) x[j] *= h.src1_scale;
        }
        if (do_silu && (!XS || c < ctot)) {
#pragma unroll
          for (int j = 0; j < 8; ++j) x[j] = PRECISE ? silu_precise(x[j]) : silu_f(x[j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = 0.f;
      }
      store8(tile + (size_t)row * ldsld + c, x);
    }
    if (more) cur = nxt;
  }
  __syncthreads();
  TK_STAMP(4);

  // ---- (5) MFMA loop: weights from the ring, activations from row-shifted views of the LDS tile -------------
  // the residual of the epilogue is requested first: its round trip hides behind the loop
  float rres[MF][NF][4];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int r = 0; r < 4; ++r) rres[mf][nf][r] = 0.f;
  if (e.residual) {
    const T* res0 = reinterpret_cast<const T*>(e.residual);
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const int m = (mt0 + mf) * 16 + lg * 4;
      if (mt0 + mf >= h.MT) continue;
      int ph = 0;
      for (int k = 1; k < e.ps_f; ++k) ph += (m >= k * e.out_C) ? 1 : 0;
      const int co = m - ph * e.out_C;
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const int n = nf * 16 + li;
        const int q = t0 + n;
        const int ty = q * e.ps_f + ph - e.ps_off;
        const bool ok = n < h.tb && q < h.L_out && ty >= 0 && ty < e.L_y;
        if (ok) load4(res0 + (size_t)((unsigned)(b * e.y_brows + e.y_row0 + ty) * (unsigned)e.ld_res + (unsigned)co), rres[mf][nf]);
      }
    }
  }
  f32x4 acc[MF][NF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int ldsrow[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int n = nf * 16 + li;
    ldsrow[nf] = ((n < h.tb ? n : 0) * h.stride) * ldsld + lg * 8;
  }
  int c_tap = 0, c_kc = 0;
  bool in_extra = false;
  for (int ks = 0; ks < KS; ks += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      if (ks + u < KS) {
        const T* bp = tile + c_tap * ldsld + c_kc * 32;
        Frag bfr[NF];
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) lds_read(bfr[nf], bp + ldsrow[nf]);
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) mma(acc[mf][nf], ring[u][mf], bfr[nf]);
        // (tap, chunk) order; behind the last tap the extra chunks follow at the centre row, columns ctot + 32 j
        if (++c_kc == kch && !in_extra) {
          c_kc = 0;
          if (++c_tap == h.taps && XS) { in_extra = true; c_tap = h.pad_left; c_kc = kch; }
        }
      }
      issueA(ring[u]);
    }
  }

  TK_STAMP(5);
  // ---- (6) epilogue: bias, residual, sub-pixel row mapping, store, statistics of the next GroupNorm -----------
  T* yT = reinterpret_cast<T*>(e.y);
  float* yF = reinterpret_cast<float*>(e.y);
  const int m_wg0 = blockIdx.x * 64 * MF;                       // first GEMM row of the workgroup
  int phw = 0;
  for (int k = 1; k < e.ps_f; ++k) phw += (m_wg0 >= k * e.out_C) ? 1 : 0;
  const int fgw0 = (int)(((float)(m_wg0 - phw * e.out_C) + 0.5f) * e.inv_out_cpf);      // first fine group the workgroup can touch
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    const int m = (mt0 + mf) * 16 + lg * 4;
    if (mt0 + mf >= h.MT) continue;
    int ph = 0;
    for (int k = 1; k < e.ps_f; ++k) ph += (m >= k * e.out_C) ? 1 : 0;
    const int co = m - ph * e.out_C;
    float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e.bias) bb = *reinterpret_cast<const float4*>(e.bias + co);
    float gs[2] = {0.f, 0.f}, gq[2] = {0.f, 0.f};
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const int n = nf * 16 + li;
      const int q = t0 + n;
      const int ty = q * e.ps_f + ph - e.ps_off;
      const bool ok = n < h.tb && q < h.L_out && ty >= 0 && ty < e.L_y;
      if (ok) {
        const unsigned yrow = (unsigned)(b * e.y_brows + e.y_row0 + ty);
        float v[4] = {acc[mf][nf][0] + bb.x, acc[mf][nf][1] + bb.y, acc[mf][nf][2] + bb.z, acc[mf][nf][3] + bb.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += rres[mf][nf][r];
        const size_t off = (size_t)(yrow * (unsigned)e.ld_y + (unsigned)co);
        if (e.y_f32) store4(yF + off, v);
        else store4(yT + off, v);
        gs[0] += v[0] + v[1]; gq[0] += v[0] * v[0] + v[1] * v[1];
        gs[1] += v[2] + v[3]; gq[1] += v[2] * v[2] + v[3] * v[3];
      }
    }
    if (e.out_gn_stats) {
      // all columns of the tile belong to batch element b: reduce over the 16 columns, then LDS, then global
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const float s = row16_sum(gs[p]), q2 = row16_sum(gq[p]);
        if (li == 0) {
          const int rel = (int)(((float)(co + 2 * p) + 0.5f) * e.inv_out_cpf) - fgw0;
          atomicAdd(st_lds + 2 * rel, s);
          atomicAdd(st_lds + 2 * rel + 1, q2);
        }
      }
    }
  }
  TK_STAMP(6);
  if (e.out_gn_stats) {
    __syncthreads();
    const int nrel = 64 * MF / 2 + 2;
    for (int i = tid; i < 2 * nrel; i += 256) {
      const float v = st_lds[i];
      const int fg = fgw0 + (i >> 1);
      if (v != 0.f && fg < JEN1_FINE_GROUPS) unsafeAtomicAdd(e.out_gn_stats + (size_t)b * 64 + fg * 2 + (i & 1), v);
    }
  }
#ifdef JEN1_TILE_PROFILE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TK_STAMP(7);
  if (tid == 0 && g_tile_dbg) {
    unsigned long long* d = g_tile_dbg + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
#pragma unroll
    for (int i_ = 0; i_ < 8; ++i_) d[i_] = tk_t[i_];
  }
#endif
}

template <typename T, int MF, int NF, int PF, bool XS>
int launch_tile_x(const TileArgs& ta, int rows_in, hipStream_t s) {
  const int ctot = ta.hot.c0 + ta.hot.c1;
  const size_t tile_bytes = ((size_t)rows_in * (ctot + ta.epi.c2 + ta.epi.c3 + 8) * sizeof(T) + 15) & ~(size_t)15;
  const size_t lds = tile_bytes + (size_t)(2 * ctot + 2 * (64 * MF / 2 + 2)) * sizeof(float);
  JEN1_CHECK(lds <= 160 * 1024, "conv_gemm: tile kernel LDS request %zu B exceeds 160 KiB", lds);
  auto kern = tile_gemm_kernel<T, MF, NF, PF, XS>;
  JEN1_MAX_LDS_ONCE(kern, 160 * 1024);
  dim3 grid((ta.hot.MT * 16 + 64 * MF - 1) / (64 * MF), ta.hot.tiles_t * ta.hot.B);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, ta);
  JEN1_HIP(hipGetLastError());
  return 0;
}

template <typename T, int MF, int NF, int PF>
int launch_tile(const TileArgs& ta, int rows_in, hipStream_t s) {
  return (ta.epi.c2 + ta.epi.c3) ? launch_tile_x<T, MF, NF, PF, true>(ta, rows_in, s) : launch_tile_x<T, MF, NF, PF, false>(ta, rows_in, s);
}

}  // namespace

#ifdef JEN1_TILE_PROFILE
extern "C" int jen1_tile_debug_buffer(void* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_tile_dbg), &p, sizeof(p)) == hipSuccess ? 0 : 1;
}
#endif

#ifndef JEN1_TILE_PF
#define JEN1_TILE_PF 4
#endif

// called by jen1_conv_gemm for the T* tile configurations (validated there)
int jen1_tile_gemm_launch(const jen1_conv_args& a, void* stream) {
  TileArgs ta;
  memset(&ta, 0, sizeof(ta));
  TileHot& h = ta.hot;
  TileEpi& e = ta.epi;
  const int es = a.dtype == JEN1_F32 ? 4 : 2;
  const bool gn = a.pro_mode == JEN1_PRO_GN || a.pro_mode == JEN1_PRO_GN_SILU;
  JEN1_CHECK(a.nb == 1 && a.splitk == 1 && !a.ln_fold && !a.row_scale && !a.out_rowstats && a.act == JEN1_ACT_NONE && a.nseg >= 0 && a.nseg <= 2 && a.m_split == 0 &&
             (gn || a.pro_mode == JEN1_PRO_NONE || a.pro_mode == JEN1_PRO_SILU),
             "conv_gemm: the T* tile configurations take GroupNorm / SiLU / no prologue, one batch element per tile, no split-K, no LayerNorm");
  JEN1_CHECK(!a.out_gn_stats || a.out_cpf >= 2, "conv_gemm: bad out_cpf");
  const int ctot = a.c0 + a.c1;
  h.x0 = a.x0; h.x1 = a.x1; h.w = a.w;
  // for the T* tiles ``seg`` lists only the raw EXTRA K segments behind the taps (row shift 0)
  int kx = 0;
  for (int i = 0; i < a.nseg; ++i) {
    JEN1_CHECK(a.seg[i].x && a.seg[i].shift == 0 && a.seg[i].kch >= 1 && a.seg[i].ld >= 32 * a.seg[i].kch, "conv_gemm: bad extra K segment %d for a T* tile", i);
    kx += a.seg[i].kch;
  }
  if (a.nseg >= 1) { e.x2 = a.seg[0].x; e.ld2 = a.seg[0].ld; e.c2 = 32 * a.seg[0].kch; }
  if (a.nseg >= 2) { e.x3 = a.seg[1].x; e.ld3 = a.seg[1].ld; e.c3 = 32 * a.seg[1].kch; }
  const int64_t wb = ((int64_t)a.taps * (ctot / 32) + kx) * (a.M / 16) * 512 * es;
  JEN1_CHECK(wb < (int64_t)OOB, "conv_gemm: packed weight too large for 31-bit offsets");
  h.w_bytes = (uint32_t)wb;
  h.B = a.B; h.L_in = a.L_in; h.L_out = a.L_out; h.c0 = a.c0; h.c1 = a.c1; h.ld0 = a.ld0; h.ld1 = a.ld1;
  h.taps = a.taps; h.stride = a.stride; h.pad_left = a.pad_left; h.MT = a.M / 16; h.tb = a.tb;
  h.tiles_t = (a.L_out + a.tb - 1) / a.tb;
  h.inv_tiles_t = 1.0f / (float)h.tiles_t;
  h.pro_mode = a.pro_mode;
  h.src1_scale = a.src1_scale;
  JEN1_CHECK((int64_t)a.B * a.L_in * (a.ld0 > a.ld1 ? a.ld0 : a.ld1) < ((int64_t)1 << 31) && (int64_t)a.B * a.y_brows * a.ld_y < ((int64_t)1 << 31),
             "conv_gemm: tensor too large for 32-bit element offsets");
  if (gn) {
    h.st0 = a.gn_stats0; h.st1 = a.gn_stats1; h.gamma = a.gn_gamma; h.beta = a.gn_beta;
    h.film = a.film ? a.film + a.film_off : nullptr;
    h.film_step = a.film_step; h.film_row = a.film_row;
    h.groups = a.gn_groups; h.cpg = a.gn_cpg; h.film_C = a.film_C; h.film_ld = a.film_ld;
    const int cpf0 = a.c0 / JEN1_FINE_GROUPS, cpf1 = a.c1 ? a.c1 / JEN1_FINE_GROUPS : 1;
    JEN1_CHECK(a.c0 % JEN1_FINE_GROUPS == 0 && a.c1 % JEN1_FINE_GROUPS == 0, "conv_gemm: GroupNorm sources must be multiples of 32 channels");
    h.nfg0 = a.gn_groups == 1 ? JEN1_FINE_GROUPS : (a.gn_cpg + cpf0 - 1) / cpf0;
    h.nfg1 = a.c1 ? (a.gn_cpg + cpf1 - 1) / cpf1 : 0;
    JEN1_CHECK(a.gn_groups == 1 || (a.gn_cpg % cpf0 == 0 && (a.c1 == 0 || a.gn_cpg % cpf1 == 0)),
               "conv_gemm: T* tiles need GroupNorm groups made of whole statistics fine groups (cpg=%d)", a.gn_cpg);
    JEN1_CHECK(a.gn_groups > 1 || a.c1 == 0, "conv_gemm: a single GroupNorm group over two sources is not supported by the T* tiles");
    h.eps = a.gn_eps;
    h.inv_count = 1.0f / (float)a.gn_count;
    h.inv_cpg = 1.0f / (float)a.gn_cpg;
    h.inv_cpf0 = 1.0f / (float)cpf0;
    h.inv_cpf1 = 1.0f / (float)cpf1;
  }
  e.y = a.y; e.residual = a.residual; e.bias = a.bias; e.out_gn_stats = a.out_gn_stats;
  e.M = a.M; e.out_C = a.out_C; e.ps_f = a.ps_f; e.ps_off = a.ps_off; e.L_y = a.L_y; e.y_brows = a.y_brows; e.y_row0 = a.y_row0;
  e.ld_y = a.ld_y; e.ld_res = a.ld_res; e.y_f32 = a.y_f32; e.out_cpf = a.out_cpf;
  e.inv_out_cpf = a.out_gn_stats ? 1.0f / (float)a.out_cpf : 1.0f;
  const int rows_in = (a.tb - 1) * a.stride + a.taps;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // ring depth 4: measured on the bench shapes, a ring that covers every k-step (12) costs ~0.7 us more issue time per workgroup and the
  // MFMA loop does not get shorter -- every workgroup re-reads the whole packed weight from L2 (98 KB at C = 128, k = 3), which is a
  // throughput cost, not a latency one (tools/tile_profile.py)
  constexpr int PF = JEN1_TILE_PF;
  if (a.dtype == JEN1_F32) {
    switch (a.cfg) {
      case JEN1_CFG_T128x64: return launch_tile<float, 2, 4, PF>(ta, rows_in, s);
      case JEN1_CFG_T128x32: return launch_tile<float, 2, 2, PF>(ta, rows_in, s);
      case JEN1_CFG_T128x16: return launch_tile<float, 2, 1, PF>(ta, rows_in, s);
      case JEN1_CFG_T256x32: return launch_tile<float, 4, 2, PF>(ta, rows_in, s);
      case JEN1_CFG_T256x16: return launch_tile<float, 4, 1, PF>(ta, rows_in, s);
      case JEN1_CFG_T64x64: return launch_tile<float, 1, 4, PF>(ta, rows_in, s);
    }
  } else {
    switch (a.cfg) {
      case JEN1_CFG_T128x64: return launch_tile<bf16_t, 2, 4, PF>(ta, rows_in, s);
      case JEN1_CFG_T128x32: return launch_tile<bf16_t, 2, 2, PF>(ta, rows_in, s);
      case JEN1_CFG_T128x16: return launch_tile<bf16_t, 2, 1, PF>(ta, rows_in, s);
      case JEN1_CFG_T256x32: return launch_tile<bf16_t, 4, 2, PF>(ta, rows_in, s);
      case JEN1_CFG_T256x16: return launch_tile<bf16_t, 4, 1, PF>(ta, rows_in, s);
      case JEN1_CFG_T64x64: return launch_tile<bf16_t, 1, 4, PF>(ta, rows_in, s);
    }
  }
  return jen1_set_error("jen1_conv_gemm: unknown tile cfg %d", a.cfg);
}
