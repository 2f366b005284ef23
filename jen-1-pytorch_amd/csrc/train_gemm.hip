// jen1_train_gemm: the one matrix-core kernel of the training path (include/jen1_train.h).
//
// Forward conv / linear / attention products, their data gradients and their weight gradients are all
//   C[m][n] = alpha * sum_tap sum_k A(m, tap, k) * B(n, tap, k)
// with operands addressed through (row stride, inner stride, tap stride) and an optional index map that
// restates the zero padding / striding of _Conv1d and ConvTranspose1d (reference blocks.py:34-53, 69-95).
// 64 x 64 output tile per workgroup, 4 waves as 2 x 2, each 2 x 2 MFMA 16x16 tiles; K step 32 staged through LDS
// with the next step's global loads in flight during the MFMAs.  Operands that are contiguous along the output
// row instead of along K (data gradient's weights, every weight-gradient operand) are transposed on the way into
// LDS, so the MFMA fragments are always 8 consecutive K values of one row.
#include "common.h"
#include "jen1_train.h"
#include <stdlib.h>

namespace {

constexpr int BM = 64, BN = 64, BK = 32, NT = 256;
#ifndef JEN1_SKINNY_PF8
#define JEN1_SKINNY_PF8 8
#endif
#ifndef JEN1_SKINNY_NW96
#define JEN1_SKINNY_NW96 16     // waves per workgroup for K walks of 64 steps or more
#endif
#ifndef JEN1_LEAN_PF
#define JEN1_LEAN_PF 4          // K steps of loads in flight per wave in the lean register-direct kernel (bf16); measured at 24 000 rows: 4 -> 22.4 us (4 waves per SIMD), 6 -> 23.5 (3), 8 -> 27.9 (2)
#endif

struct Operand {
  const void* p;
  long long ld_r, ld_k, tap_stride;
  int map_axis, map_L, map_Lsrc, map_mul, map_tapmul, map_shift, map_div, map_reflect;
  int rows;   // number of valid rows (M or N)
  const int* shift_b;   // per batch element b of the mapped axis: its map_shift (a pass that mixes causal and non-causal clips), or NULL
};

struct GemmDev {
  Operand a, b;
  void* c;
  const float* bias;
  float* rowsum;     // += sum_k A(m, tap 0, k) for every m (bias gradient riding on the weight gradient), or NULL
  const void* res;   // residual added in the epilogue (C's dtype and indexing), or NULL
  long long ldc_m, ldc_n, c_tap_stride;
  long long a_zs0, a_zs1, b_zs0, b_zs1, c_zs0, c_zs1;
  int a_zdiv, b_zdiv, c_zdiv;
  int M, N, K, taps, taps_in_z, splitk, atomic, accumulate, c_f32;
  int direct;        // both operands K-contiguous, fragments straight from global memory (no LDS staging)
  int tall;          // direct path only: the 4 waves stack along M (tile 128 x 32) because N <= 32
  int fastw;         // weight-gradient layout (both operands [k][row], rows contiguous, bf16): wgrad_loop instead of the generic staging
  float alpha;
};

template <typename T> struct Vec;
template <> struct Vec<bf16_t> { static constexpr int N = 8; typedef bf16x8 type; };
template <> struct Vec<float> { static constexpr int N = 4; typedef f32x4 type; };

// index map of jen1_gemm_operand: returns the mapped index or -1
__device__ __forceinline__ int shift_of(const Operand& o, int b) { return o.shift_b != nullptr ? o.shift_b[b] : o.map_shift; }

__device__ __forceinline__ long long map_index(const Operand& o, int i, int tap) {
  const int b = i / o.map_L, t = i - b * o.map_L;
  int s = t * o.map_mul + tap * o.map_tapmul + shift_of(o, b);
  if (o.map_reflect) {                       // F.pad(mode="reflect"): ... 2 1 | 0 1 2 ... L-1 | L-2 L-3 ...
    if (s < 0) s = -s;
    if (s >= o.map_Lsrc) s = 2 * (o.map_Lsrc - 1) - s;
  }
  if (s < 0) return -1;
  if (o.map_div > 1) {
    const int q = s / o.map_div;
    if (q * o.map_div != s) return -1;
    s = q;
  }
  if (s >= o.map_Lsrc) return -1;
  return (long long)b * o.map_Lsrc + s;
}

// element offset of (row, tap, k) or -1 when it reads as zero
__device__ __forceinline__ long long elem_offset(const Operand& o, int row, int tap, int k, int K) {
  if (row >= o.rows || k >= K) return -1;
  long long r = row, kk = k;
  if (o.map_axis == 1) { r = map_index(o, row, tap); if (r < 0) return -1; }
  if (o.map_axis == 2) { kk = map_index(o, k, tap); if (kk < 0) return -1; }
  return (long long)tap * o.tap_stride + r * o.ld_r + kk * o.ld_k;
}

template <typename T>
struct Staged {
  typename Vec<T>::type v[BM * BK / Vec<T>::N / NT];
};

// The index map without its division: (b, t) of the mapped index are kept per thread (rows of a k-contiguous
// operand never change during the K loop; the k index of a row-contiguous one advances by BK per step), so the
// loop body only does the multiply-add of the map.  Integer divisions per vector per step used to dominate the loop.
__device__ __forceinline__ long long map_from_bt(const Operand& o, int b, int t, int tap, int shift) {      // shift = shift_of(o, b), hoisted
  int s = t * o.map_mul + tap * o.map_tapmul + shift;
  if (o.map_reflect) {
    if (s < 0) s = -s;
    if (s >= o.map_Lsrc) s = 2 * (o.map_Lsrc - 1) - s;
  }
  if (s < 0) return -1;
  if (o.map_div > 1) {
    const int q = s / o.map_div;          // map_div is a small constant per launch; only the strided gradients pay this
    if (q * o.map_div != s) return -1;
    s = q;
  }
  if (s >= o.map_Lsrc) return -1;
  return (long long)b * o.map_Lsrc + s;
}

template <typename T>
struct Pre {            // per-thread (b, t) of the mapped index of every vector this thread fetches
  static constexpr int NV = BM * BK / Vec<T>::N / NT;
  int b[NV], t[NV], s[NV];   // s = shift_of(o, b)
  bool hoisted;         // false: fall back to map_index (division) every step
};

template <typename T>
__device__ __forceinline__ void pre_init(Pre<T>& p, const Operand& o, int row0, int k_first, bool k_monotonic, int tid) {
  constexpr int V = Vec<T>::N;
  const bool kc = (o.ld_k == 1), rc = (o.ld_r == 1) && !kc;
  p.hoisted = (!rc && o.map_axis == 1) || (rc && o.map_axis == 2 && k_monotonic);
#pragma unroll
  for (int i = 0; i < Pre<T>::NV; ++i) {
    const int v = tid + i * NT;
    const int idx = !rc ? row0 + v / (BK / V) : k_first + v / (BM / V);
    p.b[i] = p.hoisted ? idx / o.map_L : 0;
    p.t[i] = p.hoisted ? idx - p.b[i] * o.map_L : 0;
    p.s[i] = p.hoisted ? shift_of(o, p.b[i]) : 0;
  }
}

// global -> registers for one (tap, k0) step of one operand (64 rows x 32 k)
template <typename T>
__device__ __forceinline__ void fetch(Staged<T>& st, Pre<T>& pre, const Operand& o, const T* base, int row0, int tap, int k0, int K, int tid) {
  constexpr int V = Vec<T>::N;
  constexpr int NV = BM * BK / V / NT;
  typedef typename Vec<T>::type vec_t;
  const bool kc = (o.ld_k == 1), rc = (o.ld_r == 1) && !kc;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = tid + i * NT;
    vec_t val;
#pragma unroll
    for (int j = 0; j < V; ++j) val[j] = (T)0.f;
    if (!rc) {
      // vector along k: row = v / (BK / V)
      const int r = row0 + v / (BK / V), k = k0 + (v % (BK / V)) * V;
      bool done = false;
      if (kc && o.map_axis != 2 && k + V <= K) {
        long long off;
        if (r >= o.rows) off = -1;
        else if (o.map_axis == 1 && pre.hoisted) {
          const long long rr = map_from_bt(o, pre.b[i], pre.t[i], tap, pre.s[i]);
          off = rr < 0 ? -1 : (long long)tap * o.tap_stride + rr * o.ld_r + k;
        } else {
          off = elem_offset(o, r, tap, k, K);
        }
        if (off < 0) done = true;
        else if ((((unsigned long long)(base + off)) & 15) == 0) { val = *reinterpret_cast<const vec_t*>(base + off); done = true; }
      }
      if (!done) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
          const long long off = elem_offset(o, r, tap, k + j, K);
          if (off >= 0) val[j] = base[off];
        }
      }
    } else {
      // vector along rows at one k: k = v / (BM / V)
      const int k = k0 + v / (BM / V), r = row0 + (v % (BM / V)) * V;
      bool done = false;
      if (o.map_axis != 1 && r + V <= o.rows) {
        long long off;
        if (k >= K) off = -1;
        else if (o.map_axis == 2 && pre.hoisted) {
          const long long kk = map_from_bt(o, pre.b[i], pre.t[i], tap, pre.s[i]);
          off = kk < 0 ? -1 : (long long)tap * o.tap_stride + r + kk * o.ld_k;
        } else {
          off = elem_offset(o, r, tap, k, K);
        }
        if (off < 0) done = true;
        else if ((((unsigned long long)(base + off)) & 15) == 0) { val = *reinterpret_cast<const vec_t*>(base + off); done = true; }
      }
      if (!done) {
#pragma unroll
        for (int j = 0; j < V; ++j) {
          const long long off = elem_offset(o, r + j, tap, k, K);
          if (off >= 0) val[j] = base[off];
        }
      }
      if (o.map_axis == 2 && pre.hoisted) {        // next step of this K slice: k += BK
        pre.t[i] += BK;
        if (pre.t[i] >= o.map_L) {
          while (pre.t[i] >= o.map_L) { pre.t[i] -= o.map_L; ++pre.b[i]; }
          pre.s[i] = shift_of(o, pre.b[i]);
        }
      }
    }
    st.v[i] = val;
  }
}

// registers -> LDS tile [64][PITCH] (k contiguous)
template <typename T, int PITCH>
__device__ __forceinline__ void stash(const Staged<T>& st, const Operand& o, T* tile, int tid) {
  constexpr int V = Vec<T>::N;
  constexpr int NV = BM * BK / V / NT;
  typedef typename Vec<T>::type vec_t;
  const bool rc = (o.ld_r == 1) && (o.ld_k != 1);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = tid + i * NT;
    if (!rc) {
      const int r = v / (BK / V), k = (v % (BK / V)) * V;
      *reinterpret_cast<vec_t*>(tile + r * PITCH + k) = st.v[i];
    } else {
      const int k = v / (BM / V), r = (v % (BM / V)) * V;
#pragma unroll
      for (int j = 0; j < V; ++j) tile[(r + j) * PITCH + k] = st.v[i][j];
    }
  }
}

__device__ __forceinline__ void mma8(f32x4& acc, const bf16_t* a, const bf16_t* b) {
  const bf16x8 fa = *reinterpret_cast<const bf16x8*>(a);
  const bf16x8 fb = *reinterpret_cast<const bf16x8*>(b);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma8(f32x4& acc, const float* a, const float* b) {
  float fa[8], fb[8];
  load8(a, fa);
  load8(b, fb);
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j], fb[j], acc, 0, 0, 0);
}

// ---- register-direct path: both operands K-contiguous (forward conv / linear, data gradient over transposed weights) ----
// A lane's MFMA fragment (row lane % 16, 8 consecutive k at (lane / 16) * 8) IS 16 contiguous bytes of its row, so the
// fragments come straight from global memory through buffer descriptors (rows in the padding / beyond M, N, K read as
// zero through an out-of-range offset): no LDS, no barriers in the K loop, PF steps of loads in flight per wave.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned D_OOB = 0x80000000u;
constexpr int D_RSRC_FLAGS = 0x00020000;

template <typename T> struct DFrag;
template <> struct DFrag<bf16_t> { typedef bf16x8 type; static constexpr int PF = 4; };
template <> struct DFrag<float> { typedef f32x8 type; static constexpr int PF = 2; };

__device__ __forceinline__ void dload(bf16x8& f, __amdgpu_buffer_rsrc_t r, unsigned voff) {
  f = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}
__device__ __forceinline__ void dload(f32x8& f, __amdgpu_buffer_rsrc_t r, unsigned voff) {
  const u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0);
  const u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(r, voff == D_OOB ? D_OOB : voff + 16u, 0, 0);
#pragma unroll
  for (int j = 0; j < 4; ++j) { f.v[j] = __uint_as_float(lo[j]); f.v[4 + j] = __uint_as_float(hi[j]); }
}
__device__ __forceinline__ void dmma(f32x4& acc, const bf16x8& a, const bf16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void dmma(f32x4& acc, const f32x8& a, const f32x8& b) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], acc, 0, 0, 0);
}

template <typename T, int PFX = 0>
__device__ __forceinline__ void direct_loop(const GemmDev& g, const T* abase, const T* bbase, int m0, int n0, int wm, int wn, int lane,
                                            int s_begin, int s_end, int ksteps, f32x4 (&acc)[2][2]) {
  typedef typename DFrag<T>::type Frag;
  constexpr int PF = PFX > 0 ? PFX : DFrag<T>::PF;
  constexpr unsigned ES = sizeof(T);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(abase), 0, 0x7fffffff, D_RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(bbase), 0, 0x7fffffff, D_RSRC_FLAGS);
  const int li = lane & 15, kq = (lane >> 4) * 8;
  int a_b[2], a_t[2], a_s[2], a_row[2];
  bool a_ok[2];
  unsigned b_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = m0 + wm * 32 + i * 16 + li;
    a_ok[i] = r < g.M;
    a_row[i] = r;
    a_b[i] = (g.a.map_axis == 1) ? r / g.a.map_L : 0;
    a_t[i] = (g.a.map_axis == 1) ? r - a_b[i] * g.a.map_L : 0;
    a_s[i] = (g.a.map_axis == 1 && a_ok[i]) ? shift_of(g.a, a_b[i]) : 0;
    const int n = n0 + wn * 32 + i * 16 + li;
    b_off[i] = n < g.N ? (unsigned)((long long)n * g.b.ld_r) * ES : D_OOB;
  }
  Frag fa[PF][2], fb[PF][2];
  // The byte offsets of this lane's rows are functions of the TAP only: they are computed when the walk enters a tap (one division
  // per kernel, the 64-bit products and the index map once per tap) and a step adds its k.  Recomputed per step they were the
  // kernel: ~100 vector instructions with quarter-rate 64-bit multiplies per step and lane, on 4 waves per SIMD.
  int s_next = s_begin;
  int cur_tap = s_begin / ksteps;
  int kidx = s_begin - cur_tap * ksteps;
  unsigned a_base[2], b_base[2];
  auto enter_tap = [&](int tap) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      long long row = a_row[i];
      if (g.a.map_axis == 1) row = map_from_bt(g.a, a_b[i], a_t[i], tap, a_s[i]);
      a_base[i] = (a_ok[i] && row >= 0) ? (unsigned)(((long long)tap * g.a.tap_stride + row * g.a.ld_r) * ES) : D_OOB;
      b_base[i] = b_off[i] != D_OOB ? b_off[i] + (unsigned)((long long)tap * g.b.tap_stride * ES) : D_OOB;
    }
  };
  enter_tap(cur_tap);
  auto issue = [&](Frag (&xa)[2], Frag (&xb)[2]) {
    const int k = kidx * BK + kq;
    const bool live = s_next < s_end && k < g.K;
    const unsigned kb = (unsigned)k * ES;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      dload(xa[i], ra, (live && a_base[i] != D_OOB) ? a_base[i] + kb : D_OOB);
      dload(xb[i], rb, (live && b_base[i] != D_OOB) ? b_base[i] + kb : D_OOB);
    }
    ++s_next;
    if (++kidx >= ksteps) { kidx = 0; enter_tap(++cur_tap); }
  };
#pragma unroll
  for (int u = 0; u < PF; ++u) issue(fa[u], fb[u]);
  for (int c = s_begin; c < s_end; c += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      if (c + u < s_end) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) dmma(acc[mi][ni], fa[u][mi], fb[u][ni]);
      }
      issue(fa[u], fb[u]);
    }
  }
}

// ---- K loop of the weight gradient (bf16): C[m][n] += sum_k A[k][m] * B[map(k, tap)][n], both operands stored [k][row] with the rows
// contiguous (dY and the layer's input as they lie in memory).  The generic staging above transposes on the way INTO LDS: eight 2-byte
// scattered writes per 16-byte vector, most of them on two banks, behind address arithmetic written for every layout -- ~1.3 us of CU
// time per 32-k step, which is what every weight-gradient launch of the pass was made of.  Here a step's tiles go into LDS as they
// are (one 16-byte global load + two 8-byte LDS stores per thread and operand, rows 136 bytes apart), and the transposition happens
// on the way OUT: a lane's MFMA fragment (row i, 8 consecutive k) is two hardware transpose reads (ds_read_b64_tr_b16) down one
// column.
constexpr int WP = 68;                           // LDS row pitch in elements (64 + 4)
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 tr_pair(const bf16_t* lo, const bf16_t* hi) {
  typedef __attribute__((address_space(3))) s16x4* lds_p;
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lo));
  const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(hi));
  const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ void wgrad_loop(const GemmDev& g, const bf16_t* abase, const bf16_t* bbase, int m0, int n0, int wm, int wn,
                                           int tap, int s_begin, int s_end, bool do_rowsum, float& rsum, bf16_t* lds,
                                           f32x4 (&acc)[2][2]) {
  const int tid = threadIdx.x, lane = tid & 63;
  bf16_t* At = lds;
  bf16_t* Bt = lds + BK * WP;
  const int kk = tid >> 3, cv = (tid & 7) * 8;           // this thread's slot in a step: row kk, columns cv .. cv + 7 of both tiles
  const bool a_col = m0 + cv + 8 <= (int)g.a.ld_k, b_col = n0 + cv + 8 <= (int)g.b.ld_k;     // (a tile may hang over the matrix)
  const bf16_t* ap = abase + m0 + cv;
  const bf16_t* bp = bbase + n0 + cv;
  int k = s_begin * BK + kk;
  int kb = 0, kt = k;                                   // (batch element, position) of k under B's index map
  int ks = 0;
  if (g.b.map_axis == 2) { kb = k / g.b.map_L; kt = k - kb * g.b.map_L; ks = k < g.K ? shift_of(g.b, kb) : 0; }
  const bf16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  constexpr int WST = 4;                                 // steps of global loads in flight (one 16-byte vector per operand each)
  bf16x8 ra[WST], rb[WST];
  auto fetch_ab = [&](bf16x8& xa, bf16x8& xb) {          // (called in step order)
    xa = zero;
    xb = zero;
    if (k < g.K) {
      if (a_col) xa = *reinterpret_cast<const bf16x8*>(ap + (long long)k * g.a.ld_k);
      long long row = k;
      if (g.b.map_axis == 2) row = map_from_bt(g.b, kb, kt, tap, ks);
      if (b_col && row >= 0) xb = *reinterpret_cast<const bf16x8*>(bp + row * g.b.ld_k);
    }
    k += BK;
    if (g.b.map_axis == 2) {
      kt += BK;
      if (kt >= g.b.map_L) {
        while (kt >= g.b.map_L) { kt -= g.b.map_L; ++kb; }
        ks = k < g.K ? shift_of(g.b, kb) : 0;
      }
    }
  };
#pragma unroll
  for (int u = 0; u < WST; ++u)
    if (s_begin + u < s_end) fetch_ab(ra[u], rb[u]);
  const int li = lane & 15, kq = (lane >> 4) * 8;
  typedef unsigned long long u64;
  float rs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int s0 = s_begin; s0 < s_end; s0 += WST) {
#pragma unroll
    for (int u = 0; u < WST; ++u) {
      const int s = s0 + u;
      if (s >= s_end) break;
      if (do_rowsum) {                                   // bias gradient: column sums of A, from the registers that hold its rows
#pragma unroll
        for (int j = 0; j < 8; ++j) rs[j] += (float)ra[u][j];
      }
      {
        const u64* pa = reinterpret_cast<const u64*>(&ra[u]);
        const u64* pb = reinterpret_cast<const u64*>(&rb[u]);
        u64* da = reinterpret_cast<u64*>(At + kk * WP + cv);
        u64* db = reinterpret_cast<u64*>(Bt + kk * WP + cv);
        da[0] = pa[0]; da[1] = pa[1];
        db[0] = pb[0]; db[1] = pb[1];
      }
      __syncthreads();
      if (s + WST < s_end) fetch_ab(ra[u], rb[u]);
      // ds_read_b64_tr_b16: the sixteen lanes of a k group each name 4 consecutive columns of one of 4 consecutive rows (lane p: row
      // p / 4, columns 4 (p % 4) ..) and receive column p of that 4 x 16 block, i.e. 4 consecutive k of their own row of the fragment
      bf16x8 fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int ao = (kq + (li >> 2)) * WP + wm * 32 + i * 16 + (li & 3) * 4;
        const int bo = (kq + (li >> 2)) * WP + wn * 32 + i * 16 + (li & 3) * 4;
        fa[i] = tr_pair(At + ao, At + ao + 4 * WP);
        fb[i] = tr_pair(Bt + bo, Bt + bo + 4 * WP);
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
      __syncthreads();
    }
  }
  if (do_rowsum) {                                       // the 32 threads of a column group meet in LDS once, in a fixed order
    float* red = reinterpret_cast<float*>(lds);          // [32][64] float32 = 8 KB over the two tiles (the loop is done with them)
#pragma unroll
    for (int j = 0; j < 8; ++j) red[kk * BM + cv + j] = rs[j];
    __syncthreads();
    if (tid < BM) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < BK; ++r) t += red[r * BM + tid];
      rsum = t;
    }
  }
}

// one workgroup of the 64 x 64 form: (bx, by, bz) = its position in the grid jen1_train_gemm would launch
// LEAN: only the register-direct K loop is compiled in (the caller guarantees g.direct and no float32 read-modify-write epilogue):
// the full body holds the staging registers of every path at once (224 VGPRs: 2 waves per SIMD), the lean one fits 4 waves per SIMD,
// which is what the many-row forward / data-gradient products of the long levels need to hide their K walk's round trips.
template <typename T, bool LEAN = false>
__device__ __forceinline__ void gemm_body(const GemmDev& g, int bx, int by, int bz, T* As, T* Bs) {
  constexpr int PITCH = BK + 16 / (int)sizeof(T);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = g.tall ? wave : wave >> 1, wn = g.tall ? 0 : wave & 1;
  const int m0 = bx * (g.tall ? 2 * BM : BM), n0 = by * (g.tall ? BN / 2 : BN);
  int z = bz;
  const int split = z % g.splitk;
  z /= g.splitk;
  int tap_z = 0;
  if (g.taps_in_z) { tap_z = z % g.taps; z /= g.taps; }
  const T* abase = reinterpret_cast<const T*>(g.a.p) + (long long)(z / g.a_zdiv) * g.a_zs0 + (long long)(z % g.a_zdiv) * g.a_zs1;
  const T* bbase = reinterpret_cast<const T*>(g.b.p) + (long long)(z / g.b_zdiv) * g.b_zs0 + (long long)(z % g.b_zdiv) * g.b_zs1;

  const int ksteps = (g.K + BK - 1) / BK;
  const int ntap = g.taps_in_z ? 1 : g.taps;
  const int total = ksteps * ntap;
  const int per = (total + g.splitk - 1) / g.splitk;
  const int s_begin = split * per, s_end = min(total, s_begin + per);

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // the bias of this lane's two output columns is requested before the K loop (its latency hides behind it)
  float bias_v[2] = {0.f, 0.f};
  if (g.bias != nullptr) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int n = n0 + wn * 32 + ni * 16 + (lane & 15);
      bias_v[ni] = n < g.N ? g.bias[n] : 0.f;
    }
  }

  // bias gradient: the workgroups of the first N tile and tap 0 also sum their A rows over K
  const bool do_rowsum = g.rowsum != nullptr && by == 0 && tap_z == 0;
  float rsum = 0.f;

  // plain read-modify-write of a float32 C (the weight gradient accumulating into .grad): the tile's old values are requested HERE,
  // all sixteen at once and ahead of the K loop.  Read in the epilogue, each load would sit behind the previous element's store
  // (the compiler cannot reorder them: same array) -- sixteen memory round trips in a row, ~15 us, the floor of every weight-
  // gradient launch before this.
  char* cb = reinterpret_cast<char*>(g.c);
  const long long coff = (long long)(z / g.c_zdiv) * g.c_zs0 + (long long)(z % g.c_zdiv) * g.c_zs1 + (long long)tap_z * g.c_tap_stride;
  const bool rmw32 = !LEAN && g.c_f32 && !g.atomic && g.accumulate;
  float cold[2][2][4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int n = n0 + wn * 32 + ni * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 32 + mi * 16 + (lane >> 4) * 4 + r;
        cold[mi][ni][r] = 0.f;
        if (rmw32 && n < g.N && m < g.M) cold[mi][ni][r] = reinterpret_cast<const float*>(cb)[coff + (long long)m * g.ldc_m + (long long)n * g.ldc_n];
      }
    }

  bool fastw = false;
  if constexpr (sizeof(T) == 2) fastw = g.fastw != 0;
  if (LEAN || g.direct) {
    direct_loop<T, (LEAN && sizeof(T) == 2) ? JEN1_LEAN_PF : 0>(g, abase, bbase, m0, n0, wm, wn, lane, s_begin, s_end, ksteps, acc);
  } else if constexpr (LEAN) {
  } else if (fastw) {
    if constexpr (sizeof(T) == 2) wgrad_loop(g, abase, bbase, m0, n0, wm, wn, tap_z, s_begin, s_end, do_rowsum, rsum, As, acc);
  } else {
  // ST steps of global loads in flight per thread (a step is one 16-byte vector per operand in bf16, two in float32).  Measured on
  // the weight gradients (tools/wgrad_shapes.py): ST = 2 changes nothing, ST = 4 is SLOWER (512 x 512 x 192: 13.4 -> 17.9 us) -- a step's
  // ~1.3 us is address arithmetic and the transposing LDS writes, not the memory round trip.
  constexpr int ST = 1;
  Staged<T> sa[ST], sb[ST];
  Pre<T> pa, pb;
  // with one matrix per tap (weight gradient) the steps of this K slice are consecutive in k: s -> k0 = s * BK
  pre_init<T>(pa, g.a, m0, s_begin * BK, g.taps_in_z != 0, tid);
  pre_init<T>(pb, g.b, n0, s_begin * BK, g.taps_in_z != 0, tid);
  auto fetch_step = [&](int s, Staged<T>& xa, Staged<T>& xb) {      // (called in step order: the hoisted indices advance per call)
    const int tap = g.taps_in_z ? tap_z : s / ksteps;
    const int k0 = (g.taps_in_z ? s : s % ksteps) * BK;
    fetch<T>(xa, pa, g.a, abase, m0, tap, k0, g.K, tid);
    fetch<T>(xb, pb, g.b, bbase, n0, tap, k0, g.K, tid);
  };
#pragma unroll
  for (int u = 0; u < ST; ++u)
    if (s_begin + u < s_end) fetch_step(s_begin + u, sa[u], sb[u]);
  for (int s0 = s_begin; s0 < s_end; s0 += ST) {
#pragma unroll
    for (int u = 0; u < ST; ++u) {
      const int s = s0 + u;
      if (s >= s_end) break;
      stash<T, PITCH>(sa[u], g.a, As, tid);
      stash<T, PITCH>(sb[u], g.b, Bs, tid);
      __syncthreads();
      if (s + ST < s_end) fetch_step(s + ST, sa[u], sb[u]);
      if (do_rowsum && tid < BM) {
#pragma unroll
        for (int k = 0; k < BK; ++k) rsum += (float)As[tid * PITCH + k];
      }
      const int kq = (lane >> 4) * 8, rr = lane & 15;
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          mma8(acc[mi][ni], As + (wm * 32 + mi * 16 + rr) * PITCH + kq, Bs + (wn * 32 + ni * 16 + rr) * PITCH + kq);
      __syncthreads();
    }
  }
  }

  if (do_rowsum && tid < BM && m0 + tid < g.M) atomicAdd(g.rowsum + m0 + tid, g.alpha * rsum);

  // epilogue: acc[r] <-> (m = 4 * (lane / 16) + r, n = lane % 16) of the 16 x 16 tile
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int n = n0 + wn * 32 + ni * 16 + (lane & 15);
      if (n >= g.N) continue;
      const float bv = split == 0 ? bias_v[ni] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 32 + mi * 16 + (lane >> 4) * 4 + r;
        if (m >= g.M) continue;
        float v = g.alpha * acc[mi][ni][r] + bv;
        const long long off = coff + (long long)m * g.ldc_m + (long long)n * g.ldc_n;
        if (g.c_f32) {
          float* cp = reinterpret_cast<float*>(cb) + off;
          if (g.atomic) atomicAdd(cp, v);
          else *cp = v + cold[mi][ni][r];
        } else {
          T* cp = reinterpret_cast<T*>(cb) + off;
          if (g.accumulate) v += (float)*cp;
          if (g.res) v += (float)reinterpret_cast<const T*>(g.res)[off];
          *cp = (T)v;
        }
      }
    }
}

// ---- skinny form of the register-direct path: few rows, a big weight -- the deep levels of the pass (M = B x T' = 16 .. 384 rows
// against 512 .. 1024 x 1024 x 3 weights).  In 64 x 64 tiles such a GEMM is a handful of workgroups, so it ran split over K with float
// atomics into a float32 scratch plus a convert launch (14 + 4.6 us).  Here a workgroup owns 32 rows x 16 columns and ITS FOUR WAVES
// split K (step s goes to wave s mod 4, 8 steps of loads in flight per wave), meet in LDS in a fixed order and write C in its own
// dtype: N / 16 x M / 32 workgroups, no atomics, no scratch, no second launch.
template <typename T> struct SkinnyPF;
template <> struct SkinnyPF<bf16_t> { static constexpr int PF = 8; };
template <> struct SkinnyPF<float> { static constexpr int PF = 3; };

typedef float SkinnyRed[2][64][4];
constexpr int SKINNY_RED_BYTES = 3 * (int)sizeof(SkinnyRed);

// NW waves split the K steps of the workgroup's 32 x 16 tile (step s goes to wave s mod NW)
template <typename T, int NW = 4>
__device__ __forceinline__ void skinny_body(const GemmDev& g, int bx, int by, int bz, SkinnyRed* red) {
  typedef typename DFrag<T>::type Frag;
  constexpr int PF = (NW == 16 && sizeof(T) == 2) ? 6 : (NW == 8 && sizeof(T) == 2) ? JEN1_SKINNY_PF8 : SkinnyPF<T>::PF;      // (16 waves leave 128 registers per lane)
  constexpr unsigned ES = sizeof(T);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = bx * 32, n0 = by * 16;
  int z = bz;
  const int split = z % g.splitk;
  z /= g.splitk;
  const T* abase = reinterpret_cast<const T*>(g.a.p) + (long long)(z / g.a_zdiv) * g.a_zs0 + (long long)(z % g.a_zdiv) * g.a_zs1;
  const T* bbase = reinterpret_cast<const T*>(g.b.p) + (long long)(z / g.b_zdiv) * g.b_zs0 + (long long)(z % g.b_zdiv) * g.b_zs1;
  const int ksteps = (g.K + BK - 1) / BK;
  const int total = ksteps * g.taps;
  const int per = (total + g.splitk - 1) / g.splitk;
  const int s_begin = split * per, s_end = min(total, s_begin + per);
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(abase), 0, 0x7fffffff, D_RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(bbase), 0, 0x7fffffff, D_RSRC_FLAGS);
  const int li = lane & 15, kq = (lane >> 4) * 8;
  int a_b[2], a_t[2], a_s[2], a_row[2];
  bool a_ok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = m0 + i * 16 + li;
    a_ok[i] = r < g.M;
    a_row[i] = r;
    a_b[i] = (g.a.map_axis == 1) ? r / g.a.map_L : 0;
    a_t[i] = (g.a.map_axis == 1) ? r - a_b[i] * g.a.map_L : 0;
    a_s[i] = (g.a.map_axis == 1 && a_ok[i]) ? shift_of(g.a, a_b[i]) : 0;
  }
  const int nb = n0 + li;
  const unsigned b_off = nb < g.N ? (unsigned)((long long)nb * g.b.ld_r) * ES : D_OOB;
  float bias_v = 0.f;
  if (g.bias != nullptr && wave == 0 && split == 0) bias_v = nb < g.N ? g.bias[nb] : 0.f;
  f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  Frag fa[PF][2], fb[PF];
  // (row offsets per tap, k added per step: see direct_loop)
  int s_next = s_begin + wave;
  int cur_tap = s_next / ksteps;
  int kidx = s_next - cur_tap * ksteps;
  unsigned a_base[2], b_base;
  auto enter_tap = [&](int tap) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      long long row = a_row[i];
      if (g.a.map_axis == 1) row = map_from_bt(g.a, a_b[i], a_t[i], tap, a_s[i]);
      a_base[i] = (a_ok[i] && row >= 0) ? (unsigned)(((long long)tap * g.a.tap_stride + row * g.a.ld_r) * ES) : D_OOB;
    }
    b_base = b_off != D_OOB ? b_off + (unsigned)((long long)tap * g.b.tap_stride * ES) : D_OOB;
  };
  enter_tap(cur_tap);
  auto issue = [&](Frag (&xa)[2], Frag& xb) {
    const int k = kidx * BK + kq;
    const bool live = s_next < s_end && k < g.K;
    const unsigned kb = (unsigned)k * ES;
#pragma unroll
    for (int i = 0; i < 2; ++i) dload(xa[i], ra, (live && a_base[i] != D_OOB) ? a_base[i] + kb : D_OOB);
    dload(xb, rb, (live && b_base != D_OOB) ? b_base + kb : D_OOB);
    s_next += NW;
    kidx += NW;
    if (kidx >= ksteps) {
      do { kidx -= ksteps; ++cur_tap; } while (kidx >= ksteps);
      enter_tap(cur_tap);
    }
  };
#pragma unroll
  for (int u = 0; u < PF; ++u) issue(fa[u], fb[u]);
  for (int c = s_begin + wave; c < s_end; c += PF * NW) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      if (c + u * NW < s_end) {
        dmma(acc[0], fa[u][0], fb[u]);
        dmma(acc[1], fa[u][1], fb[u]);
      }
      issue(fa[u], fb[u]);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) *reinterpret_cast<float4*>(&red[wave - 1][mi][lane][0]) = make_float4(acc[mi][0], acc[mi][1], acc[mi][2], acc[mi][3]);
  }
  __syncthreads();
  if (wave != 0 || nb >= g.N) return;
  char* cb = reinterpret_cast<char*>(g.c);
  const long long coff = (long long)(z / g.c_zdiv) * g.c_zs0 + (long long)(z % g.c_zdiv) * g.c_zs1;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    float4 t = make_float4(acc[mi][0], acc[mi][1], acc[mi][2], acc[mi][3]);
#pragma unroll
    for (int w = 0; w < NW - 1; ++w) {
      const float4 o = *reinterpret_cast<const float4*>(&red[w][mi][lane][0]);
      t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
    }
    const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + mi * 16 + (lane >> 4) * 4 + r;
      if (m >= g.M) continue;
      float v = g.alpha * tv[r] + bias_v;
      const long long off = coff + (long long)m * g.ldc_m + (long long)nb * g.ldc_n;
      if (g.c_f32) {
        float* cp = reinterpret_cast<float*>(cb) + off;
        if (g.atomic) atomicAdd(cp, v);
        else { if (g.accumulate) v += *cp; *cp = v; }
      } else {
        T* cp = reinterpret_cast<T*>(cb) + off;
        if (g.accumulate) v += (float)*cp;
        if (g.res) v += (float)reinterpret_cast<const T*>(g.res)[off];
        *cp = (T)v;
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void train_gemm_kernel(const GemmDev g) {
  constexpr int PITCH = BK + 16 / (int)sizeof(T);
  __shared__ __attribute__((aligned(16))) T tiles[(BM + BN) * PITCH];      // A tile, then B tile (wgrad_loop lays its own two tiles over both)
  static_assert(sizeof(T) != 2 || 2 * BK * WP <= (BM + BN) * PITCH, "wgrad_loop's tiles must fit");
  jen1_prefetch_kernarg<sizeof(GemmDev)>();      // one batch of scalar loads instead of one round trip per argument line
  gemm_body<T>(g, blockIdx.x, blockIdx.y, blockIdx.z, tiles, tiles + BM * PITCH);
}

template <typename T>
__global__ __launch_bounds__(NT) void train_gemm_direct_kernel(const GemmDev g) {
  jen1_prefetch_kernarg<sizeof(GemmDev)>();
  gemm_body<T, true>(g, blockIdx.x, blockIdx.y, blockIdx.z, nullptr, nullptr);
}

template <typename T, int NW>
__global__ __launch_bounds__(NW * 64) void train_gemm_skinny_kernel(const GemmDev g) {
  __shared__ __attribute__((aligned(16))) SkinnyRed red[NW - 1];
  jen1_prefetch_kernarg<sizeof(GemmDev)>();
  skinny_body<T, NW>(g, blockIdx.x, blockIdx.y, blockIdx.z, red);
}

// ---- two independent products in ONE launch (jen1_train_gemm_pair): the weight gradient and the data gradient of a layer read the
// same dY and nothing orders them, but as two launches they cost the chain two launch latencies and each leaves most of the chip
// idle (16 .. 384 rows at the deep levels).  One grid holds the workgroups of the first product (64 x 64 form) and of the second
// (64 x 64 or skinny form); the launch lasts as long as the longer of the two.
struct PairDev {
  GemmDev g0, g1;
  int nb1, gx0, gy0, gx1, gy1;
};

template <typename T, bool SKINNY1>
__global__ __launch_bounds__(NT) void train_gemm_pair_kernel(const PairDev p) {
  constexpr int PITCH = BK + 16 / (int)sizeof(T);
  constexpr int AB = 2 * BM * PITCH * (int)sizeof(T);
  __shared__ __attribute__((aligned(16))) char lds[AB > SKINNY_RED_BYTES ? AB : SKINNY_RED_BYTES];
  jen1_prefetch_kernarg<sizeof(PairDev)>();
  T* As = reinterpret_cast<T*>(lds);
  T* Bs = As + BM * PITCH;
  // the SECOND product's workgroups come first in the dispatch order: it is the short chain-critical one (few workgroups, each a
  // long K walk), and starts at once while the first product's many short workgroups fill the rest of the chip behind it
  int id = blockIdx.x;
  if (id >= p.nb1) {
    id -= p.nb1;
    const int bx = id % p.gx0;
    id /= p.gx0;
    gemm_body<T>(p.g0, bx, id % p.gy0, id / p.gy0, As, Bs);
  } else {
    const int bx = id % p.gx1;
    id /= p.gx1;
    if constexpr (SKINNY1) skinny_body<T>(p.g1, bx, id % p.gy1, id / p.gy1, reinterpret_cast<SkinnyRed*>(lds));
    else gemm_body<T>(p.g1, bx, id % p.gy1, id / p.gy1, As, Bs);
  }
}

int check_operand(const jen1_gemm_operand& o, const char* name) {
  JEN1_CHECK(o.p != nullptr, "train_gemm: operand %s is NULL", name);
  JEN1_CHECK(o.zdiv >= 1, "train_gemm: operand %s: zdiv must be >= 1", name);
  JEN1_CHECK(o.map_axis >= 0 && o.map_axis <= 2, "train_gemm: operand %s: map_axis must be 0, 1 or 2", name);
  if (o.map_axis) {
    JEN1_CHECK(o.map_L >= 1 && o.map_Lsrc >= 1 && o.map_div >= 1 && o.map_mul >= 1,
               "train_gemm: operand %s: map_L, map_Lsrc, map_mul and map_div must be >= 1", name);
  }
  return 0;
}

Operand to_dev(const jen1_gemm_operand& o, int rows, const int32_t* shift_b) {
  Operand d;
  d.p = o.p; d.ld_r = o.ld_r; d.ld_k = o.ld_k; d.tap_stride = o.tap_stride;
  d.map_axis = o.map_axis; d.map_L = o.map_L; d.map_Lsrc = o.map_Lsrc; d.map_mul = o.map_mul;
  d.map_tapmul = o.map_tapmul; d.map_shift = o.map_shift; d.map_div = o.map_div; d.map_reflect = o.map_reflect ? 1 : 0; d.rows = rows;
  d.shift_b = (o.reserved == 1 && o.map_axis != 0) ? reinterpret_cast<const int*>(shift_b) : nullptr;
  return d;
}

// validates one call and lays out its launch: the device descriptor, the grid, and whether it is the skinny form
int prepare(const jen1_gemm_args* args, GemmDev& g, dim3& grid, bool& skinny) {
  JEN1_CHECK(args != nullptr, "train_gemm: args is NULL");
  const jen1_gemm_args& a = *args;
  JEN1_CHECK(a.dtype == JEN1_F32 || a.dtype == JEN1_BF16, "train_gemm: dtype must be JEN1_F32 or JEN1_BF16");
  JEN1_CHECK(a.M >= 1 && a.N >= 1 && a.K >= 1 && a.taps >= 1 && a.batches >= 1, "train_gemm: M, N, K, taps, batches must be >= 1");
  JEN1_CHECK(a.c != nullptr, "train_gemm: c is NULL");
  JEN1_CHECK(a.splitk >= 1, "train_gemm: splitk must be >= 1");
  JEN1_CHECK(a.splitk == 1 || a.atomic, "train_gemm: splitk > 1 needs the atomic epilogue");
  JEN1_CHECK(!a.atomic || a.c_f32, "train_gemm: the atomic epilogue needs a float32 C");
  JEN1_CHECK(a.c_zdiv >= 1, "train_gemm: c_zdiv must be >= 1");
  if (check_operand(a.a, "a") || check_operand(a.b, "b")) return 1;
  const long long gz = (long long)a.batches * (a.taps_in_z ? a.taps : 1) * a.splitk;
  JEN1_CHECK(gz <= 65535, "train_gemm: batches * taps * splitk = %lld exceeds the grid limit", gz);
  const int gy = (a.N + BN - 1) / BN;
  JEN1_CHECK(gy <= 65535, "train_gemm: N = %d is too large", a.N);
  g.a = to_dev(a.a, a.M, a.map_shift_b);
  g.b = to_dev(a.b, a.N, a.map_shift_b);
  JEN1_CHECK((a.a.reserved != 1 && a.b.reserved != 1) || a.map_shift_b != nullptr, "train_gemm: an operand asks for per-batch-element shifts but map_shift_b is NULL");
  g.c = a.c; g.bias = reinterpret_cast<const float*>(a.bias);
  g.rowsum = reinterpret_cast<float*>(a.rowsum);
  g.res = a.residual;
  JEN1_CHECK(a.residual == nullptr || (!a.atomic && !a.c_f32), "train_gemm: a residual needs the plain epilogue in C's dtype");
  JEN1_CHECK(a.rowsum == nullptr || a.taps_in_z, "train_gemm: rowsum rides on the per-tap (weight gradient) form only");
  g.ldc_m = a.ldc_m; g.ldc_n = a.ldc_n; g.c_tap_stride = a.c_tap_stride;
  g.a_zs0 = a.a.zs0; g.a_zs1 = a.a.zs1; g.b_zs0 = a.b.zs0; g.b_zs1 = a.b.zs1; g.c_zs0 = a.c_zs0; g.c_zs1 = a.c_zs1;
  g.a_zdiv = a.a.zdiv; g.b_zdiv = a.b.zdiv; g.c_zdiv = a.c_zdiv;
  g.M = a.M; g.N = a.N; g.K = a.K; g.taps = a.taps; g.taps_in_z = a.taps_in_z ? 1 : 0; g.splitk = a.splitk;
  g.atomic = a.atomic ? 1 : 0; g.accumulate = a.accumulate ? 1 : 0; g.c_f32 = a.c_f32 ? 1 : 0; g.alpha = a.alpha;
  {
    // register-direct path: K-contiguous operands whose rows start on 16-byte boundaries and whose extents fit the
    // 31-bit offsets of a buffer descriptor; the mapped axis of A may only be its rows; B is a plain matrix per tap
    static const bool no_direct = getenv("JEN1_TRAIN_GEMM_NO_DIRECT") != nullptr;     // tuning / A-B switch, read once
    const long long es = a.dtype == JEN1_F32 ? 4 : 2;
    const long long vec = 16 / es;
    auto aligned = [&](const jen1_gemm_operand& o) {
      return o.ld_k == 1 && o.ld_r % vec == 0 && o.tap_stride % vec == 0 && o.zs0 % vec == 0 && o.zs1 % vec == 0 &&
             ((uintptr_t)o.p & 15) == 0;
    };
    const long long a_rows = a.a.map_axis == 1 ? (long long)((a.M + a.a.map_L - 1) / a.a.map_L) * a.a.map_Lsrc : a.M;
    const long long a_span = ((long long)(a.taps - 1) * (a.a.tap_stride > 0 ? a.a.tap_stride : 0) + a_rows * a.a.ld_r + a.K) * es;
    const long long b_span = ((long long)(a.taps - 1) * a.b.tap_stride + (long long)a.N * a.b.ld_r + a.K) * es;
    g.direct = (!a.taps_in_z && a.rowsum == nullptr && aligned(a.a) && aligned(a.b) && a.a.map_axis != 2 && a.b.map_axis == 0 &&
                a.K % vec == 0 && a.a.tap_stride >= 0 && a.b.tap_stride >= 0 && a_span < (1ll << 31) && b_span < (1ll << 31) &&
                !no_direct) ? 1 : 0;
  }
  // narrow outputs (N <= 32, e.g. the last stages of the SEANet decoder): in the 2 x 2 wave layout half of the waves
  // would only multiply padding; on the direct path the four waves stack along M instead
  g.tall = (g.direct && a.N <= 32) ? 1 : 0;
  {
    // the weight-gradient layout in bf16: both operands [k][row] with contiguous rows on 16-byte boundaries, one matrix per tap, A
    // unmapped, B unmapped or mapped along k
    static const bool no_fastw = getenv("JEN1_TRAIN_GEMM_NO_FASTW") != nullptr;       // tuning / A-B switch, read once
    auto rows_ok = [&](const jen1_gemm_operand& o) {
      return o.ld_r == 1 && o.ld_k >= 8 && o.ld_k % 8 == 0 && o.zs0 % 8 == 0 && o.zs1 % 8 == 0 && ((uintptr_t)o.p & 15) == 0 && !o.map_reflect;
    };
    g.fastw = (a.dtype == JEN1_BF16 && a.taps_in_z && !g.direct && rows_ok(a.a) && rows_ok(a.b) && a.a.map_axis == 0 &&
               (a.b.map_axis == 0 || a.b.map_axis == 2) && a.a.tap_stride == 0 && a.b.tap_stride == 0 && !no_fastw) ? 1 : 0;
  }
  skinny = a.reserved == 1 && g.direct && !g.tall;
  if (skinny) {
    // the caller asked for the skinny form (few rows, a big weight: see train_gemm_skinny_kernel) and the operands allow it
    const long long wgs = (long long)((a.M + 31) / 32) * ((a.N + 15) / 16);
    JEN1_CHECK(wgs * gz <= (1ll << 24) && (a.N + 15) / 16 <= 65535, "train_gemm: skinny grid too large");
    grid = dim3((a.M + 31) / 32, (a.N + 15) / 16, (unsigned)gz);
    return 0;
  }
  grid = dim3(g.tall ? (a.M + 2 * BM - 1) / (2 * BM) : (a.M + BM - 1) / BM, g.tall ? 1 : gy, (unsigned)gz);
  return 0;
}

}  // namespace

extern "C" int jen1_train_gemm(const jen1_gemm_args* args, void* stream) {
  GemmDev g;
  dim3 grid;
  bool skinny = false;
  if (prepare(args, g, grid, skinny)) return 1;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const bool f32 = args->dtype == JEN1_F32;
  if (skinny) {
    // waves per workgroup by the length of the K walk: a wave keeps 8 (bf16) / 3 (float32) steps of loads in flight, and a tile with
    // 96 steps (3 taps x 1024 channels) on 4 waves is three memory round trips in a row where 16 waves make it one
    const int steps = ((args->K + BK - 1) / BK) * args->taps / args->splitk;
    const int nw = steps >= 64 ? JEN1_SKINNY_NW96 : steps >= 48 ? 8 : 4;      // (32 steps on 8 waves measured slower than on 4: 7.1 vs 5.6 us)
#define JEN1_SKINNY(TT, NWV) hipLaunchKernelGGL((train_gemm_skinny_kernel<TT, NWV>), grid, dim3(NWV * 64), 0, s, g)
    if (f32) { if (nw == 16) JEN1_SKINNY(float, 16); else if (nw == 8) JEN1_SKINNY(float, 8); else JEN1_SKINNY(float, 4); }
    else { if (nw == 16) JEN1_SKINNY(bf16_t, 16); else if (nw == 8) JEN1_SKINNY(bf16_t, 8); else JEN1_SKINNY(bf16_t, 4); }
#undef JEN1_SKINNY
  } else if (g.direct && !(g.c_f32 && g.accumulate && !g.atomic) && g.rowsum == nullptr) {
    if (f32) hipLaunchKernelGGL(train_gemm_direct_kernel<float>, grid, dim3(NT), 0, s, g);
    else hipLaunchKernelGGL(train_gemm_direct_kernel<bf16_t>, grid, dim3(NT), 0, s, g);
  } else {
    if (f32) hipLaunchKernelGGL(train_gemm_kernel<float>, grid, dim3(NT), 0, s, g);
    else hipLaunchKernelGGL(train_gemm_kernel<bf16_t>, grid, dim3(NT), 0, s, g);
  }
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_train_gemm_pair(const jen1_gemm_args* first, const jen1_gemm_args* second, void* stream) {
  PairDev p;
  dim3 g0, g1;
  bool sk0 = false, sk1 = false;
  if (prepare(first, p.g0, g0, sk0) || prepare(second, p.g1, g1, sk1)) return 1;
  JEN1_CHECK(first->dtype == second->dtype, "train_gemm_pair: both products must have one dtype");
  JEN1_CHECK(!sk0, "train_gemm_pair: the first product must be of the 64 x 64 form (the skinny form is for the second)");
  const long long nb0 = (long long)g0.x * g0.y * g0.z, nb1 = (long long)g1.x * g1.y * g1.z;
  JEN1_CHECK(nb0 + nb1 < (1ll << 31), "train_gemm_pair: grid too large");
  p.nb1 = (int)nb1; p.gx0 = (int)g0.x; p.gy0 = (int)g0.y; p.gx1 = (int)g1.x; p.gy1 = (int)g1.y;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((unsigned)(nb0 + nb1));
  const bool f32 = first->dtype == JEN1_F32;
  if (sk1) {
    if (f32) hipLaunchKernelGGL((train_gemm_pair_kernel<float, true>), grid, dim3(NT), 0, s, p);
    else hipLaunchKernelGGL((train_gemm_pair_kernel<bf16_t, true>), grid, dim3(NT), 0, s, p);
  } else {
    if (f32) hipLaunchKernelGGL((train_gemm_pair_kernel<float, false>), grid, dim3(NT), 0, s, p);
    else hipLaunchKernelGGL((train_gemm_pair_kernel<bf16_t, false>), grid, dim3(NT), 0, s, p);
  }
  JEN1_HIP(hipGetLastError());
  return 0;
}
