// Large-M matrix-core GEMM of the JEN-1 hot path on gfx950 (MI355X).  C ABI: include/jen1_hip.h (jen1_big_gemm,
// jen1_standardize_rows, jen1_kv_fixed_fill).
//
// The one shape class of the denoiser where "fraction of the MFMA peak" means something: the bias-free
// cross-attention projection ``to_kv`` = Linear(1024 -> 2C) over the text context (reference jen1/model/blocks.py:402-407,
// :427-434; 45.9 % of the as-written FLOPs).  Sampling projects the 128 text tokens of every batch element ONCE per
// conditioning for all 13 cross-attention layers (engine.Plan.set_context): one grouped GEMM  [B*128, 1024] x [17408, 1024]^T
// whose column groups are the layers (LayerNorm gamma / beta folded into the weights at pack time, the standardisation shared:
// jen1_standardize_rows), masked rows (blocks.py:431-434 multiplies K and V by the padding mask) and a row map into the
// [2B][129][2C] K/V caches in the epilogue.  Training runs the same kernel for the forward and data-gradient products of the
// 2B*129-row context (jen1_amd/train.py), and the TN form below for the weight gradient.
//
// Structure (cdna_hip_programming.md section 5):
//   * 128 x 128 output tile per 256-thread workgroup, K step 128 bytes (64 bf16 / 32 float32), 2 x 2 waves of 64 x 64;
//     v_mfma_f32_32x32x16_bf16 (float32 mode: v_mfma_f32_16x16x4_f32, exact fp32), accumulators in registers;
//   * both operands go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no staging registers, rows past the matrix
//     read as zeros through the buffer descriptor), two LDS buffers per operand (64 KiB: two workgroups per CU), the loads of
//     tile t+1 in flight across the barrier while tile t is multiplied (counted vmcnt, raw s_barrier);
//   * the LDS image is lane-linear per DMA instruction, so the bank swizzle sits on the SOURCE address: 16-byte chunk c of row r
//     is stored at chunk c ^ ((r >> 1) & 7) of the 128-byte row; a fragment read (ds_read_b128, 16 rows x one k group per lane
//     group) then touches every bank once;
//   * workgroup ids are remapped so that the workgroups of one XCD walk M under the same weight tile: every weight byte is read
//     from HBM once, the activations (2 - 4 MB) stay in the L2s.
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int BM = 128, BN = 128, ROWB = 128;          // tile rows; bytes of one tile row (the K step)
constexpr int TILE_B = BM * ROWB;                      // 16 KiB per operand tile
constexpr int RSRC_FLAGS = 0x00020000;

struct BGroup {                 // one column group of the stacked B operand = one output tensor (a layer)
  void* c;                      // output rows: c + row * ldc (+ column n - n0), in the compute dtype or float32
  const float* bias;            // [N] or null
  int32_t n0, N, ldc, pad;
};
static_assert(sizeof(BGroup) == 32, "group table entry");

struct BArgs {
  const void* a;                // activations [M][lda], K contiguous
  const void* b;                // weights [Ntot][ldb], K contiguous (all groups stacked)
  const BGroup* groups;         // device table, n0 ascending, every n0 and N a multiple of 128 unless n_groups == 1
  const float* row_scale;       // indexed by OUTPUT row, or null: the result row is multiplied by it (the padding mask)
  int32_t M, Ntot, K, lda, ldb, n_groups;
  int32_t rows_in, rows_out;    // output row of GEMM row m: (m / rows_in) * rows_out + m % rows_in  (rows_in = 0: m itself)
  int32_t c_f32, tiles_m, tiles_n, accumulate;
  float inv_rows_in, alpha;
  int32_t n_begin, pad0;        // first weight row (output column) of this launch: a product may be split by columns over two launches
  BGroup inl;                   // the one group of a launch whose ``groups`` is null (no device table: capturable without a copy)
};

template <typename T> struct Elem;
template <> struct Elem<bf16_t> { static constexpr int ES = 2, BK = 64; };
template <> struct Elem<float> { static constexpr int ES = 4, BK = 32; };

// XCD-aware, bijective: the launch's workgroup ids are dealt round-robin to the 8 XCDs; give every XCD a contiguous run of
// tile ids (cdna_hip_programming.md T1)
__device__ __forceinline__ int xcd_remap(int id, int n) {
  const int q = n >> 3, r = n & 7, x = id & 7, k = id >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

__device__ __forceinline__ void store_out(bf16_t* p, const float (&v)[4]) { store4(p, v); }
__device__ __forceinline__ void store_out(float* p, const float (&v)[4]) { store4(p, v); }

// ---------------------------------------------------------------------------------------------------------------------
// NT form: C[m][n] = alpha * sum_k A[m][k] B[n][k]  (+ bias[n]) (* row_scale[m])
// WM waves along M (2: 128 x 128 tile, 256 threads; 4: 256 x 128 tile, 512 threads), 2 along N, 64 x 64 per wave.
// NSTAGE = 3: ONE barrier per K step -- behind the barrier of step t every wave has finished step t - 1, so the stage it read is
// refilled with tile t + 2 right there; two tiles in flight while one is multiplied.  NSTAGE = 2: the classic double buffer, two
// barriers per step, half the LDS.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int WM, int NSTAGE>
__global__ __launch_bounds__(512) void big_gemm_nt_kernel(const BArgs g) {
  constexpr int ES = Elem<T>::ES, BK = Elem<T>::BK;
  constexpr bool F32 = is_f32<T>::value;
  constexpr int NWV = WM * 2;                          // waves
  constexpr int TM = WM * 64;                          // tile rows of the activation operand
  constexpr int A_B = TM * ROWB, B_B = BN * ROWB;      // bytes of the two operand tiles of a stage
  constexpr int STAGE = A_B + B_B;
  constexpr int PA = TM / 8 / NWV, PB = BN / 8 / NWV;  // LDS-DMA instructions per wave, tile and operand (8 rows each)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NSTAGE * STAGE];

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
  const int tn = tile / g.tiles_m, tm = tile - tn * g.tiles_m;         // consecutive ids: same weight tile, next M tile
  const int m0 = tm * TM, n0 = g.n_begin + tn * BN;

  // ---- LDS-DMA addressing: instruction i of wave w fills rows (i * NWV + w) * 8 .. + 8 of a tile; lane l lands at row + (l >> 3),
  // physical chunk l & 7, and fetches the global chunk (l & 7) ^ ((row >> 1) & 7) of that row --------------------------------
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(g.a) + (size_t)m0 * (size_t)g.lda * ES), 0,
      (int)((size_t)((g.M - m0) < TM ? (g.M - m0) : TM) * (size_t)g.lda * ES), RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(g.b) + (size_t)n0 * (size_t)g.ldb * ES), 0,
      (int)((size_t)((g.Ntot - n0) < BN ? (g.Ntot - n0) : BN) * (size_t)g.ldb * ES), RSRC_FLAGS);
  unsigned voa[4], vob[4];          // (fixed bounds: a call of the LDS-DMA builtin with type-dependent operands does not instantiate in the host pass)
  static_assert(PA <= 4 && PB <= 4, "DMA pieces per wave");
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int row = (i * NWV + w) * 8 + (lane >> 3);
    voa[i] = (unsigned)row * (unsigned)(g.lda * ES) + (unsigned)((lane & 7) ^ ((row >> 1) & 7)) * 16u;
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int row = (i * NWV + w) * 8 + (lane >> 3);
    vob[i] = (unsigned)row * (unsigned)(g.ldb * ES) + (unsigned)((lane & 7) ^ ((row >> 1) & 7)) * 16u;
  }
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  lds_u8* const lds0 = (lds_u8*)smem;
#define BG_ISSUE(stage_, kt_)                                                                                                        \
  do {                                                                                                                               \
    const unsigned so_ = (unsigned)(kt_) * (unsigned)ROWB;                                                                           \
    lds_u8* const sb_ = lds0 + (stage_) * STAGE + w * 1024;                                                                          \
    _Pragma("unroll") for (int i_ = 0; i_ < PA; ++i_)                                                                                \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(sb_ + i_ * NWV * 1024), 16, voa[i_], so_, 0, 0);                    \
    _Pragma("unroll") for (int i_ = 0; i_ < PB; ++i_)                                                                                \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(sb_ + A_B + i_ * NWV * 1024), 16, vob[i_], so_, 0, 0);              \
  } while (0)

  // ---- fragment addressing.  The weight tile is the MFMA's A operand (rows n), the activation tile its B operand (rows m): a
  // lane then holds 4 consecutive n of one output row m -- 8-byte (bf16) stores -----------------------------------------------
  const int wm = w % WM, wn = w / WM;
  constexpr int SUB = F32 ? 16 : 32;                   // rows of one MFMA tile
  constexpr int NSUB = 64 / SUB;                       // tiles per wave and operand
  constexpr int KG = F32 ? 4 : 2;                      // k groups (16-byte chunks) one MFMA k block spans
  constexpr int KB = 8 / KG;                           // k blocks per tile row
  const int fi = lane & (SUB - 1), fg = lane / SUB;
  const int swz = (fi >> 1) & 7;                       // (row >> 1) & 7: the wave's row bases are multiples of 16
  const unsigned fa = (unsigned)(wn * 64 + fi) * ROWB + A_B;         // weight tile rows of this lane
  const unsigned fb = (unsigned)(wm * 64 + fi) * ROWB;               // activation tile rows
  unsigned xo[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) xo[kb] = (unsigned)((kb * KG + fg) ^ swz) * 16u;

  typedef typename std::conditional<F32, f32x4, f32x16>::type Acc;
  Acc acc[NSUB][NSUB];
#pragma unroll
  for (int i = 0; i < NSUB; ++i)
#pragma unroll
    for (int j = 0; j < NSUB; ++j)
#pragma unroll
      for (int r = 0; r < (F32 ? 4 : 16); ++r) acc[i][j][r] = 0.f;

  const int KT = g.K / BK;
  BG_ISSUE(0, 0);
  if (NSTAGE == 3 && KT > 1) BG_ISSUE(1, 1);
  int stage = 0;
  for (int kt = 0; kt < KT; ++kt) {
    if constexpr (NSTAGE == 2) {
      // two stages (64 KiB at WM = 2: two workgroups per CU cover each other's waits): tile kt + 1 is requested into the other
      // stage, which every wave left behind the closing barrier of step kt - 1
      if (kt + 1 < KT) BG_ISSUE(stage ^ 1, kt + 1);
    }
    // this wave's pieces of tile kt have landed (tile kt + 1 may still be in flight) ...
    if (kt + 1 < KT) {
      if constexpr (PA + PB == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                                // ... and everybody else's; every wave is done with step kt - 1
    if constexpr (NSTAGE == 3) {
      if (kt + 2 < KT) BG_ISSUE(stage == 0 ? 2 : stage - 1, kt + 2);  // refill the stage step kt - 1 read
    }
    const unsigned char* base = smem + stage * STAGE;
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      u32x4 wa[NSUB], xb[NSUB];
#pragma unroll
      for (int s = 0; s < NSUB; ++s) {
        wa[s] = *reinterpret_cast<const u32x4*>(base + fa + s * SUB * ROWB + xo[kb]);
        xb[s] = *reinterpret_cast<const u32x4*>(base + fb + s * SUB * ROWB + xo[kb]);
      }
#ifdef JEN1_BGEMM_SETPRIO
      __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
      for (int i = 0; i < NSUB; ++i)
#pragma unroll
        for (int j = 0; j < NSUB; ++j) {
          if constexpr (F32) {
            // lane group fg holds k = 4 fg .. 4 fg + 3 of the 16-float block in BOTH operands: step s multiplies element s
#pragma unroll
            for (int s = 0; s < 4; ++s)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(wa[i][s]), __uint_as_float(xb[j][s]), acc[i][j], 0, 0, 0);
          } else {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wa[i]), __builtin_bit_cast(bf16x8, xb[j]), acc[i][j], 0, 0, 0);
          }
        }
#ifdef JEN1_BGEMM_SETPRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    }
    if constexpr (NSTAGE == 2) __builtin_amdgcn_s_barrier();     // the stage is free for the loads of tile kt + 2
    stage = stage == NSTAGE - 1 ? 0 : stage + 1;
  }

#undef BG_ISSUE
  // ---- epilogue ---------------------------------------------------------------------------------------------------------
  // group of this column tile (groups never straddle a tile unless there is only one)
  int gi = 0;
  if (g.groups)
    for (int k = 1; k < g.n_groups; ++k) gi += (n0 >= g.groups[k].n0) ? 1 : 0;
  const BGroup grp = g.groups ? g.groups[gi] : g.inl;
  T* cT = reinterpret_cast<T*>(grp.c);
  float* cF = reinterpret_cast<float*>(grp.c);
#pragma unroll
  for (int j = 0; j < NSUB; ++j) {
    // C/D layout: 32x32: column (operand B index, here m) = lane & 31, rows (here n) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5);
    //             16x16: column m = lane & 15, rows n = 4 (lane >> 4) + r
    const int m = m0 + wm * 64 + j * SUB + fi;
    if (m >= g.M) continue;
    int orow = m;
    if (g.rows_in > 0) {
      const int q = (int)(((float)m + 0.5f) * g.inv_rows_in);
      orow = q * g.rows_out + (m - q * g.rows_in);
    }
    const float rs = g.row_scale ? g.row_scale[orow] : 1.0f;
#pragma unroll
    for (int i = 0; i < NSUB; ++i) {
#pragma unroll
      for (int q4 = 0; q4 < (F32 ? 1 : 4); ++q4) {
        const int nl = F32 ? (wn * 64 + i * 16 + fg * 4) : (wn * 64 + i * 32 + q4 * 8 + fg * 4);
        const int n = n0 + nl - grp.n0;                              // column inside the group
        if (n >= grp.N) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][q4 * 4 + r] * g.alpha;
        if (grp.bias) {
          const float4 bb = *reinterpret_cast<const float4*>(grp.bias + n);
          v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= rs;
        const size_t off = (size_t)orow * (size_t)grp.ldc + (size_t)n;
        if (g.c_f32) {
          if (g.accumulate) {
            float o[4];
            load4(cF + off, o);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += o[r];
          }
          store_out(cF + off, v);
        } else {
          if (g.accumulate) {
            float o[4];
            load4(cT + off, o);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += o[r];
          }
          store_out(cT + off, v);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Row-wise epilogue of the large tile forms (bf16 output, no accumulation): the finished tile goes through LDS -- the stages are free
// behind the K loop -- and leaves as whole rows, 16 bytes per lane with consecutive lanes along a row.  Measured with the stores removed:
// the direct form (a lane's 4 consecutive n of ONE row: 8-byte stores to 16 - 32 different rows per wave instruction) cost 17 of the
// stacked projection's 54 us and 13 of 4096^3's 123.
//   rowwise_tables  per 16-column block of the tile: output pointer, bias, pitch, first column inside its group, the group's width; per
//                   tile row: the mapped output row (-1 past M) and the row scale
//   rowwise_store   LDS tile [TM][CP bytes] -> global rows
// ---------------------------------------------------------------------------------------------------------------------
struct RowTab {
  bf16_t* c;
  const float* bias;
  int ldc, n_first, n_lim, pad;
};

template <int TM, int NB16>
__device__ __forceinline__ void rowwise_tables(const BArgs& g, int m0, int n0, int tid, RowTab* gt, int* orl, float* rsl) {
  if (tid < NB16) {
    const int nb16 = n0 + tid * 16;
    RowTab e;
    e.c = nullptr; e.bias = nullptr; e.ldc = 0; e.n_first = 0; e.n_lim = 0; e.pad = 0;
    if (nb16 < g.Ntot) {
      int gi = 0;
      if (g.groups)
        for (int k = 1; k < g.n_groups; ++k) gi += (nb16 >= g.groups[k].n0) ? 1 : 0;
      const BGroup grp = g.groups ? g.groups[gi] : g.inl;
      e.c = reinterpret_cast<bf16_t*>(grp.c); e.bias = grp.bias; e.ldc = grp.ldc; e.n_first = nb16 - grp.n0; e.n_lim = grp.N;
    }
    gt[tid] = e;
  }
  if (tid < TM) {
    const int m = m0 + tid;
    int orow = -1;
    float rs = 1.0f;
    if (m < g.M) {
      orow = m;
      if (g.rows_in > 0) {
        const int q = (int)(((float)m + 0.5f) * g.inv_rows_in);
        orow = q * g.rows_out + (m - q * g.rows_in);
      }
      if (g.row_scale) rs = g.row_scale[orow];
    }
    orl[tid] = orow;
    rsl[tid] = rs;
  }
}

template <int TM, int TN, int NTHR>
__device__ __forceinline__ void rowwise_store(const unsigned char* ct, int CP, const RowTab* gt, const int* orl, int tid) {
  constexpr int CPR = TN / 8;                         // 16-byte chunks per tile row
  for (int ch = tid; ch < TM * CPR; ch += NTHR) {
    const int row = ch / CPR, c8 = ch - row * CPR;
    const int orow = orl[row];
    if (orow < 0) continue;
    const RowTab e = gt[c8 >> 1];
    const int n = e.n_first + (c8 & 1) * 8;
    if (e.c == nullptr || n >= e.n_lim) continue;
    const uint4 val = *reinterpret_cast<const uint4*>(ct + row * CP + c8 * 16);
    bf16_t* dst = e.c + (size_t)orow * (size_t)e.ldc + (size_t)n;
    if (n + 8 <= e.n_lim && (((size_t)dst) & 15) == 0) {
      *reinterpret_cast<uint4*>(dst) = val;
    } else {                                          // ragged group end / unaligned row pitch: 8-byte halves (widths are multiples of 4 here)
      *reinterpret_cast<uint2*>(dst) = make_uint2(val.x, val.y);
      if (n + 4 < e.n_lim) *reinterpret_cast<uint2*>(dst + 4) = make_uint2(val.z, val.w);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// NT form, 256 x 256 output tile (bf16): 8 waves = 2 along m (128 activation rows each) x 4 along n (64 weight rows each), so a wave
// owns 2 x 4 MFMA tiles of 32 x 32 (128 accumulator registers) and feeds 8 MFMAs from 6 fragment reads per k block -- the 64 x 64 waves
// of the kernel above feed 4 from 4, and issue one LDS-DMA piece per 2 MFMAs where this one issues one per 4 (the two ratios
// DESIGN.md 4c names as what is left to the vendor kernel on big products).  Two LDS stages of 64 KiB (one workgroup, two waves per
// SIMD), ONE barrier per K step: behind the barrier of step kt every wave has left step kt - 1, so the stage that step read is refilled
// with tile kt + 1 right there and has the whole step to land.  Same swizzle, same epilogue, same column groups (256-column
// multiples here) as above.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void big_gemm_nt256_kernel(const BArgs g) {
  typedef bf16_t T;
  constexpr int ES = 2, BK = 64, NWV = 8, TM = 256, TN = 256;
  constexpr int A_B = TM * ROWB, B_B = TN * ROWB, STAGE = A_B + B_B;      // 32 KiB + 32 KiB
  constexpr int PA = TM / 8 / NWV, PB = TN / 8 / NWV;                     // 4 + 4 LDS-DMA pieces per wave and K step
  constexpr int NSA = 2, NSB = 4;                                         // MFMA tiles of a wave: weight rows (n) x activation rows (m)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE + 12288];      // (+12 KiB: the row-wise epilogue's output tile)

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
  // Tile order inside an XCD's run of ids.  The 32 CUs of an XCD work on ~32 consecutive ids at a time and share one 4 MiB L2: as a
  // column of 32 M tiles under one weight tile they pull 32 + 1 operand slices per K step through it, as a bm x bn block (bm * bn = 32)
  // bm + bn.  Measured without the MFMAs (-DJEN1_BG256_NOMMA) the kernel moved its operands at 7.7 TB/s whatever the tile -- the
  // traffic behind the L2s is the bound of big products, not the matrix cores.
  int tn, tm;
  {
    const int bm = (g.tiles_m & 7) == 0 ? 8 : ((g.tiles_m & 3) == 0 ? 4 : ((g.tiles_m & 1) == 0 ? 2 : 1));
    const int bn = 32 / bm;
    if (g.tiles_n % bn == 0 && bm > 1) {
      const int grp = tile >> 5, r = tile & 31;
      const int gpm = g.tiles_m / bm;                   // blocks along m
      const int gn_ = grp / gpm, gm_ = grp - gn_ * gpm;
      tm = gm_ * bm + (r % bm);
      tn = gn_ * bn + (r / bm);
    } else {
      tn = tile / g.tiles_m;
      tm = tile - tn * g.tiles_m;                       // consecutive ids: same weight tile, next M tile
    }
  }
  const int m0 = tm * TM, n0 = g.n_begin + tn * TN;

  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(g.a) + (size_t)m0 * (size_t)g.lda * ES), 0,
      (int)((size_t)((g.M - m0) < TM ? (g.M - m0) : TM) * (size_t)g.lda * ES), RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(g.b) + (size_t)n0 * (size_t)g.ldb * ES), 0,
      (int)((size_t)((g.Ntot - n0) < TN ? (g.Ntot - n0) : TN) * (size_t)g.ldb * ES), RSRC_FLAGS);
  unsigned voa[PA], vob[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int row = (i * NWV + w) * 8 + (lane >> 3);
    voa[i] = (unsigned)row * (unsigned)(g.lda * ES) + (unsigned)((lane & 7) ^ ((row >> 1) & 7)) * 16u;
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int row = (i * NWV + w) * 8 + (lane >> 3);
    vob[i] = (unsigned)row * (unsigned)(g.ldb * ES) + (unsigned)((lane & 7) ^ ((row >> 1) & 7)) * 16u;
  }
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  lds_u8* const lds0 = (lds_u8*)smem;
#define BG_ISSUE(stage_, kt_)                                                                                                        \
  do {                                                                                                                               \
    const unsigned so_ = (unsigned)(kt_) * (unsigned)ROWB;                                                                           \
    lds_u8* const sb_ = lds0 + (stage_) * STAGE + w * 1024;                                                                          \
    _Pragma("unroll") for (int i_ = 0; i_ < PA; ++i_)                                                                                \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(sb_ + i_ * NWV * 1024), 16, voa[i_], so_, 0, 0);                    \
    _Pragma("unroll") for (int i_ = 0; i_ < PB; ++i_)                                                                                \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(sb_ + A_B + i_ * NWV * 1024), 16, vob[i_], so_, 0, 0);              \
  } while (0)

  const int wm = w & 1, wn = w >> 1;
  const int fi = lane & 31, fg = lane >> 5;
  const int swz = (fi >> 1) & 7;
  const unsigned fa = (unsigned)(wn * (NSA * 32) + fi) * ROWB + A_B;      // weight tile rows of this lane
  const unsigned fb = (unsigned)(wm * (NSB * 32) + fi) * ROWB;            // activation tile rows
  unsigned xo[4];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) xo[kb] = (unsigned)((kb * 2 + fg) ^ swz) * 16u;

  f32x16 acc[NSA][NSB];
#pragma unroll
  for (int i = 0; i < NSA; ++i)
#pragma unroll
    for (int j = 0; j < NSB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int KT = g.K / BK;
  BG_ISSUE(0, 0);
  int stage = 0;
  for (int kt = 0; kt < KT; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this wave's pieces of tile kt have landed ...
#ifndef JEN1_BG256_NOBAR
    __builtin_amdgcn_s_barrier();                                // ... and everybody else's; every wave has left step kt - 1
#endif
#ifndef JEN1_BG256_NODMA
    if (kt + 1 < KT) BG_ISSUE(stage ^ 1, kt + 1);                // refill the stage step kt - 1 read; a whole step to land
#endif
    const unsigned char* base = smem + stage * STAGE;
#ifdef JEN1_BG256_NOMMA
    if (g.alpha == 1234.5f)
#endif
    {
      // fragments of k block kb + 1 are requested BEFORE the MFMAs of k block kb (two register sets): left to itself the compiler
      // requests a block's fragments right in front of its MFMAs and the matrix cores idle for every LDS round trip
      u32x4 wa[2][NSA], xb[2][NSB];
#pragma unroll
      for (int s = 0; s < NSA; ++s) wa[0][s] = *reinterpret_cast<const u32x4*>(base + fa + s * 32 * ROWB + xo[0]);
#pragma unroll
      for (int s = 0; s < NSB; ++s) xb[0][s] = *reinterpret_cast<const u32x4*>(base + fb + s * 32 * ROWB + xo[0]);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const int c = kb & 1, nx = c ^ 1;
        if (kb + 1 < 4) {
#pragma unroll
          for (int s = 0; s < NSA; ++s) wa[nx][s] = *reinterpret_cast<const u32x4*>(base + fa + s * 32 * ROWB + xo[kb + 1]);
#pragma unroll
          for (int s = 0; s < NSB; ++s) xb[nx][s] = *reinterpret_cast<const u32x4*>(base + fb + s * 32 * ROWB + xo[kb + 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NSB; ++j)
#pragma unroll
          for (int i = 0; i < NSA; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wa[c][i]), __builtin_bit_cast(bf16x8, xb[c][j]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    stage ^= 1;
  }
#undef BG_ISSUE

  // ---- epilogue (as above; C/D layout 32x32: column (here m) = lane & 31, rows (here n) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) ----
  if (!g.c_f32 && !g.accumulate && (g.groups != nullptr || (g.inl.N & 3) == 0)) {      // (group widths are multiples of 128)
    // bf16 output without accumulation: row-wise through LDS (rowwise_tables / rowwise_store above)
    constexpr int CP = TN * 2 + 16;
    constexpr int NB16 = TN / 16;
    __syncthreads();
    unsigned char* ct = smem;
    RowTab* gt = reinterpret_cast<RowTab*>(smem + TM * CP);
    int* orl = reinterpret_cast<int*>(smem + TM * CP + NB16 * (int)sizeof(RowTab));
    float* rsl = reinterpret_cast<float*>(orl + TM);
    static_assert(TM * CP + NB16 * (int)sizeof(RowTab) + TM * 8 <= 2 * STAGE + 12288, "epilogue tile must fit the LDS allocation");
    rowwise_tables<TM, NB16>(g, m0, n0, tid, gt, orl, rsl);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NSB; ++j) {
      const int ml = wm * (NSB * 32) + j * 32 + fi;
      const float rs = rsl[ml];
#pragma unroll
      for (int i = 0; i < NSA; ++i) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int nl = wn * (NSA * 32) + i * 32 + q4 * 8 + fg * 4;
          const RowTab e = gt[nl >> 4];
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[i][j][q4 * 4 + r] * g.alpha;
          const int nn = e.n_first + (nl & 15);
          if (e.bias && nn < e.n_lim) {
            const float4 bb = *reinterpret_cast<const float4*>(e.bias + nn);
            v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] *= rs;
          store4(reinterpret_cast<T*>(ct + ml * CP) + nl, v);
        }
      }
    }
    __syncthreads();
    rowwise_store<TM, TN, NWV * 64>(ct, CP, gt, orl, tid);
    return;
  }
  int gi = 0;
  if (g.groups)
    for (int k = 1; k < g.n_groups; ++k) gi += (n0 >= g.groups[k].n0) ? 1 : 0;
  const BGroup grp = g.groups ? g.groups[gi] : g.inl;
  T* cT = reinterpret_cast<T*>(grp.c);
  float* cF = reinterpret_cast<float*>(grp.c);
#pragma unroll
  for (int j = 0; j < NSB; ++j) {
    const int m = m0 + wm * (NSB * 32) + j * 32 + fi;
    if (m >= g.M) continue;
    int orow = m;
    if (g.rows_in > 0) {
      const int q = (int)(((float)m + 0.5f) * g.inv_rows_in);
      orow = q * g.rows_out + (m - q * g.rows_in);
    }
    const float rs = g.row_scale ? g.row_scale[orow] : 1.0f;
#pragma unroll
    for (int i = 0; i < NSA; ++i) {
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int nl = wn * (NSA * 32) + i * 32 + q4 * 8 + fg * 4;
        const int n = n0 + nl - grp.n0;
        if (n >= grp.N) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][q4 * 4 + r] * g.alpha;
        if (grp.bias) {
          const float4 bb = *reinterpret_cast<const float4*>(grp.bias + n);
          v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= rs;
        const size_t off = (size_t)orow * (size_t)grp.ldc + (size_t)n;
        if (g.c_f32) {
          if (g.accumulate) {
            float o[4];
            load4(cF + off, o);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += o[r];
          }
          store_out(cF + off, v);
        } else {
          if (g.accumulate) {
            float o[4];
            load4(cT + off, o);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += o[r];
          }
          store_out(cT + off, v);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// NT form with FOUR LDS stages (bf16, v_mfma_f32_16x16x32_bf16): 256 x TN output tile (built for TN = 272), K step 32 (64-byte tile rows).
// Why: with two 64 KiB stages ONE tile is in flight per CU while one is multiplied, and a K step then lasts as long as the delivery of
// 64 KB to every CU at once (~3.3 us on the stacked projection, whatever the MFMAs do: 0.9 us) -- the 256 x 256 form above moved operands
// at 5 - 6 TB/s where the 128 x 128 form (two workgroups per CU) reached 10.  Here a stage is 32 / 33 KiB and THREE tiles are in flight.
// 8 waves stacked along m (32 activation rows each), every wave walks all TN / 16 weight tiles: 32 / 34 accumulators of 4 registers.
// TN = 272: the stacked text-context projection has N = 17408 = 64 x 272 columns -- 4 x 64 = 256 tiles at B = 8: one round of one tile
// per CU (256-wide tiles: 1.06 rounds); such a column tile may straddle two groups, so the epilogue looks the group up per 16 columns.
// LDS image: 64-byte rows, 16-byte chunk c of row r at chunk c ^ S[(r >> 2) & 3], S = {0, 2, 3, 1}: the four 16-lane groups of a
// ds_read_b128 (MI355X_MICROARCH.md LDS table) then hold 16 distinct (row mod 4, chunk) pairs = all 64 banks once.
// ---------------------------------------------------------------------------------------------------------------------
template <int TN>
__global__ __launch_bounds__(512) void big_gemm_nt_s4_kernel(const BArgs g) {
  typedef bf16_t T;
  constexpr int ES = 2, BK4 = 32, RB = 64, NWV = 8, TM = 256, NST = 4;
  constexpr int A_B = TM * RB, B_B = TN * RB, STAGE = A_B + B_B;          // 16 KiB + 16 / 17 KiB
  constexpr int NPA = TM / 16, NPB = TN / 16, NP = NPA + NPB;             // LDS-DMA pieces of 16 rows per K step: 16 + 16 / 17
  constexpr int NI = TN / 16, NJ = 2;
  static_assert(NPA == 2 * NWV && NPB >= 2 * NWV && NPB <= 2 * NWV + 1, "piece distribution");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * STAGE + 16384];      // (+16 KiB: the epilogue's output tile is a little larger)

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
  int tn, tm;
  {
    const int bm = (g.tiles_m & 7) == 0 ? 8 : ((g.tiles_m & 3) == 0 ? 4 : ((g.tiles_m & 1) == 0 ? 2 : 1));
    const int bn = 32 / bm;
    if (g.tiles_n % bn == 0 && bm > 1) {
      const int grp = tile >> 5, r = tile & 31;
      const int gpm = g.tiles_m / bm;
      const int gn_ = grp / gpm, gm_ = grp - gn_ * gpm;
      tm = gm_ * bm + (r % bm);
      tn = gn_ * bn + (r / bm);
    } else {
      tn = tile / g.tiles_m;
      tm = tile - tn * g.tiles_m;
    }
  }
  const int m0 = tm * TM, n0 = g.n_begin + tn * TN;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(g.a) + (size_t)m0 * (size_t)g.lda * ES), 0,
      (int)((size_t)((g.M - m0) < TM ? (g.M - m0) : TM) * (size_t)g.lda * ES), RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(g.b) + (size_t)n0 * (size_t)g.ldb * ES), 0,
      (int)((size_t)((g.Ntot - n0) < TN ? (g.Ntot - n0) : TN) * (size_t)g.ldb * ES), RSRC_FLAGS);
  // wave w: activation pieces w and w + 8, weight pieces w and w + 8 (+ weight piece 16 for wave 0 when TN = 272); lane l of a piece lands at
  // row l >> 2, physical chunk l & 3 and fetches the global chunk (l & 3) ^ S[(row >> 2) & 3]
  const unsigned sw4 = 0x1320u;                                            // S[k] = (sw4 >> (4 k)) & 3  ->  {0, 2, 3, 1}
  unsigned vo[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int pc = j < 2 ? w + NWV * j : (j < 4 ? w + NWV * (j - 2) : 2 * NWV);
    const bool isb = j >= 2;
    const int row = pc * 16 + (lane >> 2);
    const unsigned sx = (sw4 >> (4 * ((row >> 2) & 3))) & 3u;
    vo[j] = (unsigned)row * (unsigned)((isb ? g.ldb : g.lda) * ES) + (((unsigned)(lane & 3)) ^ sx) * 16u;
  }
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  lds_u8* const lds0 = (lds_u8*)smem;
  const bool extra = (NPB > 2 * NWV) && w == 0;
#define S4_ISSUE(stage_, kt_)                                                                                                        \
  do {                                                                                                                               \
    const unsigned so_ = (unsigned)(kt_) * (unsigned)RB;                                                                             \
    lds_u8* const sb_ = lds0 + (stage_) * STAGE + w * 1024;                                                                          \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(sb_), 16, vo[0], so_, 0, 0);                                             \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(sb_ + NWV * 1024), 16, vo[1], so_, 0, 0);                                \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(sb_ + A_B), 16, vo[2], so_, 0, 0);                                       \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(sb_ + A_B + NWV * 1024), 16, vo[3], so_, 0, 0);                          \
    if (extra) __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(sb_ + A_B + 2 * NWV * 1024), 16, vo[4], so_, 0, 0);          \
  } while (0)

  const int fi = lane & 15, fg = lane >> 4;
  const unsigned xo = (((unsigned)fg) ^ ((sw4 >> (4 * ((fi >> 2) & 3))) & 3u)) * 16u;      // (tile bases are multiples of 16 rows)
  const unsigned fa = (unsigned)fi * RB + A_B + xo;                        // weight tile i: + i * 16 rows
  const unsigned fb = (unsigned)(w * 32 + fi) * RB + xo;                   // activation tile j: + j * 16 rows

  f32x4 acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int KT = g.K / BK4;
  S4_ISSUE(0, 0);
  if (KT > 1) S4_ISSUE(1, 1);
  if (KT > 2) S4_ISSUE(2, 2);
  for (int kt = 0; kt < KT; ++kt) {
    // this wave's pieces of tile kt have landed; the (up to two) later tiles may still fly: 4 pieces each (5 for wave 0 at TN = 272)
    const int later = KT - 1 - kt;
    if (later >= 2) { if (extra) asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    else if (later == 1) { if (extra) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                          // everybody's pieces; every wave has left step kt - 1
    if (kt + 3 < KT) S4_ISSUE((kt + 3) & 3, kt + 3);                       // into the stage step kt - 1 read
    const unsigned char* base = smem + (kt & 3) * STAGE;
    u32x4 xb[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) xb[j] = *reinterpret_cast<const u32x4*>(base + fb + j * 16 * RB);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const u32x4 wa = *reinterpret_cast<const u32x4*>(base + fa + i * 16 * RB);
#pragma unroll
      for (int j = 0; j < NJ; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa), __builtin_bit_cast(bf16x8, xb[j]), acc[i][j], 0, 0, 0);
    }
  }
#undef S4_ISSUE

  // ---- epilogue, bf16 output without accumulation: row-wise through LDS (rowwise_tables / rowwise_store above) ------------------------
  constexpr int CP = TN * 2 + 16;                     // bytes of a tile row in LDS (+16: the 16 rows of a store instruction hit 16 bank groups)
  if (!g.c_f32 && !g.accumulate && (g.groups != nullptr || (g.inl.N & 3) == 0)) {      // (group widths are multiples of 128)
    __syncthreads();                                  // every wave has left the K loop: the stages are free
    unsigned char* ct = smem;
    RowTab* gt = reinterpret_cast<RowTab*>(smem + TM * CP);
    int* orl = reinterpret_cast<int*>(smem + TM * CP + NI * (int)sizeof(RowTab));
    float* rsl = reinterpret_cast<float*>(orl + TM);
    static_assert(TM * CP + NI * (int)sizeof(RowTab) + TM * 8 <= NST * STAGE + 16384, "epilogue tile must fit the LDS allocation");
    rowwise_tables<TM, NI>(g, m0, n0, tid, gt, orl, rsl);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int ml = w * 32 + j * 16 + fi;
      const float rs = rsl[ml];
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const RowTab e = gt[i];
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * g.alpha;
        const int nn = e.n_first + fg * 4;
        if (e.bias && nn < e.n_lim) {
          const float4 bb = *reinterpret_cast<const float4*>(e.bias + nn);
          v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= rs;
        store4(reinterpret_cast<T*>(ct + ml * CP) + i * 16 + fg * 4, v);
      }
    }
    __syncthreads();
    rowwise_store<TM, TN, NWV * 64>(ct, CP, gt, orl, tid);
    return;
  }
  // ---- epilogue: C/D layout 16x16: column (here m) = lane & 15, rows (here n) = 4 (lane >> 4) + r ---------------------------------
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int m = m0 + w * 32 + j * 16 + fi;
    if (m >= g.M) continue;
    int orow = m;
    if (g.rows_in > 0) {
      const int q = (int)(((float)m + 0.5f) * g.inv_rows_in);
      orow = q * g.rows_out + (m - q * g.rows_in);
    }
    const float rs = g.row_scale ? g.row_scale[orow] : 1.0f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int nb16 = n0 + i * 16;                                        // first column of this 16-column block (inside ONE group)
      if (nb16 >= g.Ntot) continue;
      int gi = 0;
      if (g.groups)
        for (int k = 1; k < g.n_groups; ++k) gi += (nb16 >= g.groups[k].n0) ? 1 : 0;
      const BGroup grp = g.groups ? g.groups[gi] : g.inl;
      const int n = nb16 + fg * 4 - grp.n0;
      if (n >= grp.N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * g.alpha;
      if (grp.bias) {
        const float4 bb = *reinterpret_cast<const float4*>(grp.bias + n);
        v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= rs;
      const size_t off = (size_t)orow * (size_t)grp.ldc + (size_t)n;
      if (g.c_f32) {
        float* cF = reinterpret_cast<float*>(grp.c);
        if (g.accumulate) {
          float o[4];
          load4(cF + off, o);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += o[r];
        }
        store_out(cF + off, v);
      } else {
        T* cT = reinterpret_cast<T*>(grp.c);
        if (g.accumulate) {
          float o[4];
          load4(cT + off, o);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += o[r];
        }
        store_out(cT + off, v);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Convolution form of the NT product (bf16): y[b T_out + t][n] = sum_tap sum_c x[b T_in + t stride + tap - pad][c] W_tap[n][c]
// (+ bias[n]) (+ residual[row][n]) -- the forward and (stride 1, taps reversed) data-gradient passes of _Conv1d over many rows
// (blocks.py:34-53; the long levels of the training pass).  The 128 x 128 / two-stage structure of big_gemm_nt_kernel<bf16, 2, 2>; the
// activation tile of K step kt is the rows of ONE tap (ci is a multiple of the 64-channel step), fetched through the row map by
// the LDS-DMA (rows outside [0, T_in) get an offset beyond the descriptor and arrive as zeros); the weight tile comes from that
// tap's [N][ldw] matrix.
// ---------------------------------------------------------------------------------------------------------------------
struct CArgs {
  const void* x;                // [B T_in][ldx]
  const void* w;                // [taps][N][ldw], tap matrices w_tap_stride elements apart
  const float* bias;            // [N] or null
  const void* residual;         // [M][ldy] or null
  void* y;                      // [M][ldy]
  int32_t M, N, ldx, ldw, ldy, tiles_m, tiles_n;
  int32_t taps, ci, T_out, T_in, stride, pad, rows_x, tap_rev, w_tap_stride;
  float inv_T_out;
  const int32_t* shift_b;       // per batch element, added to the row shift tap - pad (a pass that mixes causal and centred padding), or null
  int32_t div;                  // > 1: the mapped position must be a multiple of div and is divided by it (ConvTranspose1d forward: the
  float inv_div;                //      taps that do not meet an input sample read zeros)
};

__global__ __launch_bounds__(256, 2) void big_gemm_conv_kernel(const CArgs g) {
  constexpr int ES = 2, BK = 64, NWV = 4, TM = 128;
  constexpr int A_B = TM * ROWB, STAGE = A_B + BN * ROWB;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = xcd_remap(blockIdx.x, g.tiles_m * g.tiles_n);
  const int tn = tile / g.tiles_m, tm = tile - tn * g.tiles_m;
  const int m0 = tm * TM, n0 = tn * BN;
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.x), 0, (int)((size_t)g.rows_x * (size_t)g.ldx * ES), RSRC_FLAGS);
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.w), 0, (int)((size_t)g.taps * (size_t)g.w_tap_stride * ES), RSRC_FLAGS);
  // instruction i of wave w fills rows (i * 4 + w) * 8 .. + 8 of a tile; lane l lands at row + (l >> 3), physical chunk l & 7, and
  // fetches chunk (l & 7) ^ ((row >> 1) & 7) of that row.  The activation rows go through the map: (batch element, position) of the
  // lane's four rows are fixed, the tap's shift changes with the K step
  int xb[4], xt[4];
  unsigned xc[4], vob[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (i * NWV + w) * 8 + (lane >> 3);
    const unsigned chunk = (unsigned)((lane & 7) ^ ((row >> 1) & 7)) * 16u;
    const int m = m0 + row;
    const int b = (int)(((float)m + 0.5f) * g.inv_T_out);             // m < 2^22: exact
    xb[i] = m < g.M ? b * g.T_in : -(1 << 28);                           // (rows past M: never valid)
    xt[i] = (m - b * g.T_out) * g.stride + ((g.shift_b && m < g.M) ? g.shift_b[b] : 0);
    xc[i] = chunk;
    vob[i] = (unsigned)(n0 + row) * (unsigned)(g.ldw * ES) + chunk;
  }
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  lds_u8* const lds0 = (lds_u8*)smem;
  const int spt = (g.ci + BK - 1) / BK;                                   // K steps per tap (the last one may be ragged: ci % 8 == 0)
  const unsigned ci_b = (unsigned)(g.ci * ES);
  auto a_off = [&](int i, int shift, unsigned cin_b) -> unsigned {
    int xp = xt[i] + shift;
    bool ok = xp >= 0 && xb[i] >= 0 && cin_b + xc[i] < ci_b;
    if (g.div > 1) {
      const int q = (int)(((float)xp + 0.5f) * g.inv_div);            // xp < 2^22: exact
      ok = ok && q * g.div == xp;
      xp = q;
    }
    ok = ok && xp < g.T_in;
    return ok ? (unsigned)(xb[i] + xp) * (unsigned)(g.ldx * ES) + xc[i] + cin_b : 0x7ffffff0u;
  };
#define CG_ISSUE(stage_, kt_)                                                                                                        \
  do {                                                                                                                               \
    const int tap_ = (kt_) / spt;                                                                                                    \
    const unsigned cin_b_ = (unsigned)(((kt_) - tap_ * spt) * BK * ES);                                                              \
    const unsigned sob_ = (unsigned)((g.tap_rev ? g.taps - 1 - tap_ : tap_) * g.w_tap_stride * ES) + cin_b_;                         \
    lds_u8* const sb_ = lds0 + (stage_) * STAGE + w * 1024;                                                                          \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                                                 \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(sb_ + i_ * NWV * 1024), 16, a_off(i_, tap_ - g.pad, cin_b_), 0u, 0, 0); \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                                                 \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(sb_ + A_B + i_ * NWV * 1024), 16,                                   \
                                                 cin_b_ + xc[i_] < ci_b ? vob[i_] : 0x7ffffff0u, sob_, 0, 0);                        \
  } while (0)

  // fragments: the weight tile is the MFMA's A operand (rows n), the activation tile its B operand (rows m)
  const int wm = w & 1, wn = w >> 1;
  const int fi = lane & 31, fg = lane >> 5;
  const int swz = (fi >> 1) & 7;
  const unsigned fa = (unsigned)(wn * 64 + fi) * ROWB + A_B;
  const unsigned fb = (unsigned)(wm * 64 + fi) * ROWB;
  unsigned xo[4];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) xo[kb] = (unsigned)((kb * 2 + fg) ^ swz) * 16u;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int KT = g.taps * spt;
  CG_ISSUE(0, 0);
  int stage = 0;
  for (int kt = 0; kt < KT; ++kt) {
    if (kt + 1 < KT) {
      CG_ISSUE(stage ^ 1, kt + 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const unsigned char* base = smem + stage * STAGE;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      u32x4 wa[2], xv[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        wa[q] = *reinterpret_cast<const u32x4*>(base + fa + q * 32 * ROWB + xo[kb]);
        xv[q] = *reinterpret_cast<const u32x4*>(base + fb + q * 32 * ROWB + xo[kb]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wa[i]), __builtin_bit_cast(bf16x8, xv[j]), acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_barrier();
    stage ^= 1;
  }
#undef CG_ISSUE
  // C / D layout of the 32 x 32 MFMA: column (here m) = lane & 31, rows (here n) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  bf16_t* y = reinterpret_cast<bf16_t*>(g.y);
  const bf16_t* res = reinterpret_cast<const bf16_t*>(g.residual);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + wm * 64 + j * 32 + fi;
    if (m >= g.M) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int n = n0 + wn * 64 + i * 32 + q4 * 8 + fg * 4;
        if (n >= g.N) continue;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][q4 * 4 + r];
        if (g.bias) {
          const float4 bb = *reinterpret_cast<const float4*>(g.bias + n);
          v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
        }
        const size_t off = (size_t)m * (size_t)g.ldy + (size_t)n;
        if (res) {
          float o[4];
          load4(res + off, o);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += o[r];
        }
        store_out(y + off, v);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// TN form (weight gradients, bf16): C[n][k] += alpha * sum_m A[m][n] B[m][k], float32 atomics (the reduction over m is split
// across workgroups), both operands with the reduction index as their ROW index ([m][cols], cols contiguous: dY and the layer's
// input as they lie in memory).  Tiles go into LDS as they are (LDS-DMA: 64 reduction rows x 128 columns per operand and stage,
// two stages) and are transposed on the reads: a lane's MFMA fragment -- 8 consecutive m of one column -- is two
// ds_read_b64_tr_b16 (each: a 16-lane group names 4 rows x 16 columns and receives them column-major).  LDS image of a tile:
// [64 m][8 blocks of 16 columns]; block c of row m is stored at block c ^ (2 (m & 3)): the 32 lanes of one read pass (two adjacent
// column blocks x 4 rows) then cover the 8 block positions = all 64 banks once.
// ---------------------------------------------------------------------------------------------------------------------
struct TArgs {
  const void* a;                // [M][lda]: columns n
  const void* b;                // [M][ldb]: columns k
  float* c;                     // [N][ldc] float32, accumulated with atomics
  int32_t M, N, K, lda, ldb, ldc, tiles_n, tiles_k, splits, mt_per_split;
  float alpha;
  // convolution form (jen1_big_gemm_tn_conv): K = taps * ci columns, column block [tap ci, (tap + 1) ci) reads row
  // b T_in + t stride + tap - pad of the input for reduction row m = b T_out + t (zero outside [0, T_in)), C is [N][ci][taps]
  int32_t conv, taps, ci, T_out, T_in, stride, pad, rows_b;
  float inv_T_out;
  float* bias_grad;             // [N] += column sums of A (the bias gradient), or NULL
  // taps > 1: a column tile holds cpt 8-channel chunks of EVERY tap ([tap][chunk][8]: taps * cpt <= 16 chunks), so that a workgroup owns
  // runs of cpt * 8 * taps consecutive floats of C and adds them with coalesced atomics (a tile of ONE tap would add floats `taps` apart,
  // in lines it shares with the other taps' workgroups: 41.6 against 16 us at 24 000 x 128 x (3 x 128))
  int32_t cpt, main_blocks, bias_blocks, bias_rows;
  const int32_t* shift_b;       // convolution form: per batch element, added to the row shift tap - pad, or null
  int32_t store, pad1;          // plain form, unsplit reduction: C = (not +=) the product, plain stores (jen1_big_gemm_tn_store)
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 tr_pair8(const unsigned char* lo) {       // rows r .. r + 3 and r + 4 .. r + 7 (1 KiB apart)
  typedef __attribute__((address_space(3))) s16x4* lds_p;
  const s16x4 x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lo));
  const s16x4 y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(lo + 4 * 256));
  const s16x8 v = __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(256, 2) void big_gemm_tn_kernel(const TArgs g) {
  constexpr int TROW = 256;                           // bytes of a tile row: 128 bf16 columns
  constexpr int TB = 64 * TROW;                       // 16 KiB per operand tile (64 reduction rows)
  __shared__ __attribute__((aligned(1024))) unsigned char smem[4 * TB];     // [stage][A | B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  if ((int)blockIdx.x >= g.main_blocks) {
    // bias gradient: the column sums of A over a slice of the rows, by workgroups of their own (every reduction slice adding its
    // partial sums would put ~100 workgroups' atomics on the same four lines: +23 us)
    float* colsum = reinterpret_cast<float*>(smem);                  // [16][128]
    const int j = (int)blockIdx.x - g.main_blocks;
    const int r0 = j * g.bias_rows;
    int r1 = r0 + g.bias_rows;
    r1 = r1 < g.M ? r1 : g.M;
    const int cg = tid & 15, rl = tid >> 4;
    const bf16_t* A = reinterpret_cast<const bf16_t*>(g.a);
    for (int nb = 0; nb < g.N; nb += 128) {
      float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const int c = nb + cg * 8;
      if (c < g.N) {
        // eight rows in flight per thread: the slice is a latency chain otherwise (one 16-byte load per trip: 35 us for 1 600 rows)
        for (int r = r0 + rl; r < r1; r += 16 * 8) {
          bf16x8 v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int ru = r + 16 * u;
            v[u] = *reinterpret_cast<const bf16x8*>(A + (size_t)(ru < r1 ? ru : r0) * (size_t)g.lda + c);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (r + 16 * u < r1) {
#pragma unroll
              for (int e = 0; e < 8; ++e) acc8[e] += (float)v[u][e];
            }
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 8; ++e) colsum[rl * 128 + cg * 8 + e] = acc8[e];
      __syncthreads();
      if (tid < 128 && nb + tid < g.N) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += colsum[q * 128 + tid];
        unsafeAtomicAdd(g.bias_grad + nb + tid, t * g.alpha);
      }
    }
    return;
  }
  int id = xcd_remap(blockIdx.x, g.main_blocks);
  const int split = id % g.splits;
  id /= g.splits;
  const int tk = id % g.tiles_k, tn = id / g.tiles_k;
  const int n0 = tn * 128, k0 = tk * 128;
  const int mt0 = split * g.mt_per_split;
  const int MT_all = (g.M + 63) / 64;
  int mt1 = mt0 + g.mt_per_split;
  mt1 = mt1 < MT_all ? mt1 : MT_all;
  if (mt0 >= mt1) return;

  // descriptors start at the tile's first column; rows past M are beyond num_records and read as zeros
  const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(g.a) + (size_t)n0 * 2), 0,
      (int)(((size_t)g.M * (size_t)g.lda - (size_t)n0) * 2), RSRC_FLAGS);
  const bool conv = g.conv != 0;
  const bool inter = conv && g.taps > 1;                      // interleaved taps: tile tk = input chunks [tk cpt, (tk + 1) cpt) of every tap
  const int kb0 = inter ? 0 : k0;                             // first input column behind the descriptor
  const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(reinterpret_cast<const unsigned char*>(g.b) + (size_t)kb0 * 2), 0,
      (int)(((size_t)(conv ? g.rows_b : g.M) * (size_t)g.ldb - (size_t)kb0) * 2), RSRC_FLAGS);
  // this lane's SOURCE chunk position in a tile row is the same for all its DMA instructions ((row & 3) = (lane >> 4) & 3): its tap,
  // its input chunk and whether that chunk exists
  const int pc_src = ((((lane >> 1) & 7) ^ (2 * ((lane >> 4) & 3))) << 1) | (lane & 1);
  const int tap_l = inter ? pc_src / g.cpt : 0;
  const int cc_l = inter ? tk * g.cpt + (pc_src - tap_l * g.cpt) : 0;
  const bool ok_l = !inter || (tap_l < g.taps && cc_l * 8 < g.ci);
  const int shift = tap_l - g.pad;
  // instruction i of wave w fills rows (i * 4 + w) * 4 .. + 4 of a tile; lane l lands at row + (l >> 4), 16-byte chunk l & 15 =
  // (block l >> 1 & 7, half l & 1) and fetches block ((l >> 1) & 7) ^ (2 (row & 3)) of that row
  unsigned voa[4], vob[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (i * 4 + w) * 4 + (lane >> 4);
    const int blk = ((lane >> 1) & 7) ^ (2 * (row & 3));
    const unsigned col = (unsigned)(blk * 32 + (lane & 1) * 16);
    voa[i] = (unsigned)row * (unsigned)(g.lda * 2) + col;
    vob[i] = inter ? (unsigned)(cc_l * 16) : (conv ? col : (unsigned)row * (unsigned)(g.ldb * 2) + col);
  }
  // convolution form: the input row of reduction row m (an offset far outside the descriptor reads as zeros)
  auto conv_off = [&](int mt, int i) -> unsigned {
    const int m = mt * 64 + (i * 4 + w) * 4 + (lane >> 4);
    const int b = (int)(((float)m + 0.5f) * g.inv_T_out);     // m < 2^22: exact
    const bool mv = m < g.M;
    const int xp = (m - b * g.T_out) * g.stride + shift + ((g.shift_b && mv) ? g.shift_b[b] : 0);
    const bool ok = xp >= 0 && xp < g.T_in && mv && ok_l;
    return ok ? (unsigned)(b * g.T_in + xp) * (unsigned)(g.ldb * 2) + vob[i] : 0x7ffffff0u;
  };
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  lds_u8* const lds0 = (lds_u8*)smem;
#define TN_ISSUE(stage_, mt_)                                                                                                  \
  do {                                                                                                                         \
    const unsigned soa_ = (unsigned)(mt_) * 64u * (unsigned)(g.lda * 2), sob_ = (unsigned)(mt_) * 64u * (unsigned)(g.ldb * 2);   \
    lds_u8* const sb_ = lds0 + (stage_) * 2 * TB + w * 1024;                                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                                           \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(sb_ + i_ * 4096), 16, voa[i_], soa_, 0, 0);                   \
    if (conv) {                                                                                                                \
      _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                                         \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(sb_ + TB + i_ * 4096), 16, conv_off((mt_), i_), 0u, 0, 0);  \
    } else {                                                                                                                   \
      _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                                         \
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(sb_ + TB + i_ * 4096), 16, vob[i_], sob_, 0, 0);            \
    }                                                                                                                          \
  } while (0)

  // fragment addressing: wave (wn, wk) owns 64 n x 64 k; lane = (g2 = lane >> 5: rows 8 g2 .. + 8 of a 16-row block, h = bit 4:
  // which 16-column half of the 32-column MFMA tile, p = lane & 15: row p >> 2 (+ 4 in the second read), columns 4 (p & 3) ..)
  const int wn = w >> 1, wk = w & 1;
  const int g2 = lane >> 5, h = (lane >> 4) & 1, p = lane & 15;
  const int x = 2 * (p >> 2);                                            // the row swizzle of this lane's rows ((row & 3) = p >> 2)
  const unsigned rowb = (unsigned)(8 * g2 + (p >> 2)) * TROW + (unsigned)(p & 3) * 8u;
  unsigned fa[2], fb[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    fa[s] = rowb + (unsigned)(((wn * 4 + s * 2 + h) ^ x) * 32);
    fb[s] = rowb + (unsigned)(((wk * 4 + s * 2 + h) ^ x) * 32) + TB;
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  TN_ISSUE(0, mt0);
  int stage = 0;
  for (int mt = mt0; mt < mt1; ++mt) {
    if (mt + 1 < mt1) {
      TN_ISSUE(stage ^ 1, mt + 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const unsigned char* base = smem + stage * 2 * TB;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {                                     // 16 reduction rows per MFMA
      bf16x8 va[2], vb[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        va[s] = tr_pair8(base + fa[s] + kb * 16 * TROW);
        vb[s] = tr_pair8(base + fb[s] + kb * 16 * TROW);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[i], vb[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_barrier();
    stage ^= 1;
  }
#undef TN_ISSUE
  if (inter) {
    // the tile through LDS: [128 n][128 columns = tap, chunk, channel] -> per n one run of cpt * 8 * taps floats in C's order
    // (channel-major, tap-minor), added with coalesced atomics
    float* ct = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int kc = wk * 64 + j * 32 + (lane & 31);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) ct[(wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g2) * 128 + kc] = acc[i][j][r] * g.alpha;
    }
    __syncthreads();
    const int ch0 = tk * g.cpt * 8;                                       // first input channel of the tile
    int nch = g.ci - ch0;
    nch = nch < g.cpt * 8 ? nch : g.cpt * 8;
    const int run = nch * g.taps;
    for (int nl = w; nl < 128; nl += 4) {
      const int n = n0 + nl;
      if (n >= g.N) break;
      float* dst = g.c + (size_t)n * (size_t)g.ldc + (size_t)ch0 * (size_t)g.taps;
      for (int e = lane; e < run; e += 64) {
        const int c = (int)(((float)e + 0.5f) / (float)g.taps), tp = e - c * g.taps;
        unsafeAtomicAdd(dst + e, ct[nl * 128 + tp * g.cpt * 8 + c]);
      }
    }
    return;
  }
  // D[n][k]: column (B operand index) k = lane & 31, rows n = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int k = k0 + wk * 64 + j * 32 + (lane & 31);
    if (k >= g.K) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g2;
        if (n >= g.N) continue;
        if (g.store) g.c[(size_t)n * (size_t)g.ldc + (size_t)k] = acc[i][j][r] * g.alpha;
        else unsafeAtomicAdd(g.c + (size_t)n * (size_t)g.ldc + (size_t)k, acc[i][j][r] * g.alpha);
      }
    }
  }
}

// standardise rows: y = (x - mean) / sqrt(var + eps) per row of C channels, x float32, y in the compute dtype; the statistics are
// taken over the values ROUNDED to the compute dtype (what the matrix cores see), var biased like nn.LayerNorm (blocks.py:400-401)
template <typename T>
__global__ __launch_bounds__(256) void standardize_rows_kernel(const float* __restrict__ x, T* __restrict__ y, int rows, int C, int ldx, int ldy, float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* xr = x + (size_t)row * ldx;
  float s = 0.f, q = 0.f;
  for (int c = lane * 4; c < C; c += 256) {
    float v[4];
    load4(xr + c, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] = (float)(T)v[j];
      s += v[j];
      q += v[j] * v[j];
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_xor(s, off);
    q += __shfl_xor(q, off);
  }
  const float mean = s / (float)C;
  float var = q / (float)C - mean * mean;
  var = var < 0.f ? 0.f : var;
  const float rstd = is_f32<T>::value ? 1.0f / sqrtf(var + eps) : rsqrtf(var + eps);
  T* yr = y + (size_t)row * ldy;
  for (int c = lane * 4; c < C; c += 256) {
    float v[4];
    load4(xr + c, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = ((float)(T)v[j] - mean) * rstd;
    store4(yr + c, v);
  }
}

struct FixedEntry {             // one layer: out[b][n][0..C2) = fixed[n][0..C2) * mask[b][n]
  const void* fixed;            // [rows][C2] in the compute dtype
  void* out;                    // first unconditional slot of the layer's K/V cache: [B][rows][C2]
  int32_t C2, pad;
  int64_t pad2;
};
static_assert(sizeof(FixedEntry) == 32, "fixed-slot table entry");

template <typename T>
__global__ __launch_bounds__(256) void kv_fixed_fill_kernel(const FixedEntry* __restrict__ tab, const float* __restrict__ mask, int B, int rows) {
  const FixedEntry e = tab[blockIdx.y];
  const int vpr = e.C2 >> 3;                          // 8-element vectors per row
  const int total = B * rows * vpr;
  const T* fx = reinterpret_cast<const T*>(e.fixed);
  T* out = reinterpret_cast<T*>(e.out);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int br = i / vpr, c = (i - br * vpr) * 8;
    const int b = br / rows, n = br - b * rows;
    float v[8];
    load8(fx + (size_t)n * e.C2 + c, v);
    const float mk = mask[br];
    // (x * mask in the compute dtype, as the torch expression this replaces rounded it)
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] *= mk;
    store8(out + (size_t)br * e.C2 + c, v);
  }
}

}  // namespace

extern "C" int jen1_big_gemm(const jen1_bgemm_args* a, void* stream) {
  JEN1_CHECK(a && a->a && a->b && ((a->groups && a->n_groups >= 1) || (a->c && a->n_groups <= 1)), "big_gemm: null argument");
  const int es = a->dtype == JEN1_F32 ? 4 : 2;
  const int bk = 128 / es;
  JEN1_CHECK(a->dtype == JEN1_F32 || a->dtype == JEN1_BF16, "big_gemm: dtype must be f32 or bf16");
  JEN1_CHECK(a->M >= 1 && a->Ntot >= 1 && a->K >= bk && a->K % bk == 0, "big_gemm: K=%d must be a positive multiple of %d", a->K, bk);
  JEN1_CHECK(a->lda >= a->K && a->ldb >= a->K && (a->lda * es) % 16 == 0 && (a->ldb * es) % 16 == 0, "big_gemm: row pitches must be 16-byte multiples >= K");
  JEN1_CHECK((int64_t)a->M * a->lda * es < ((int64_t)1 << 31) && (int64_t)a->Ntot * a->ldb * es < ((int64_t)1 << 31), "big_gemm: operand too large for 31-bit offsets");
  JEN1_CHECK(a->n_groups == 1 || a->Ntot % BN == 0, "big_gemm: stacked groups need 128-column multiples");
  BArgs g;
  memset(&g, 0, sizeof(g));
  g.a = a->a; g.b = a->b; g.groups = reinterpret_cast<const BGroup*>(a->groups); g.row_scale = a->row_scale;
  g.M = a->M; g.Ntot = a->Ntot; g.K = a->K; g.lda = a->lda; g.ldb = a->ldb; g.n_groups = a->groups ? a->n_groups : 1;
  if (!a->groups) { g.inl.c = a->c; g.inl.bias = a->bias; g.inl.n0 = 0; g.inl.N = a->Ntot; g.inl.ldc = a->ldc; }
  g.rows_in = a->rows_in; g.rows_out = a->rows_out; g.c_f32 = a->c_f32; g.accumulate = a->accumulate;
  g.inv_rows_in = a->rows_in > 0 ? 1.0f / (float)a->rows_in : 0.f;
  g.alpha = a->alpha;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // Three forms.  (1) 128 x 128 tiles, two LDS stages (64 KiB: two workgroups per CU cover each other's barriers and waits): products
  // of a few hundred tiles (the training shapes, 1024 .. 2064 rows against 512 .. 2048 columns).  (2) 256 x 128 tiles, 8 waves, three
  // stages: float32, and bf16 products whose groups are not 256-column multiples.  (3) 256 x 256 tiles (big_gemm_nt256_kernel), bf16:
  // products of >= ~200 such tiles.  A launch runs in rounds of one tile per CU, so the columns are cut where the big tiles fill whole
  // rounds and the rest goes to form (1) in a second launch: the stacked projection of a sampling plan (1024 x 17408: 4 x 68 tiles)
  // is one full round of 256 big tiles + 64 small ones instead of a second round that 16 of the 256 CUs work on.
  // JEN1_BGEMM_WM = 2 / 4 forces form (1) / (2), JEN1_BGEMM_T256 = 0 / 1 forbids / forces form (3) (tuning runs).
  static const int force_wm = getenv("JEN1_BGEMM_WM") ? atoi(getenv("JEN1_BGEMM_WM")) : 0;
  const char* e256 = getenv("JEN1_BGEMM_T256");      // (read per call: the parity tests switch it)
  const int force_t256 = e256 ? atoi(e256) : -1;
  static const int n_cu = [] { int d = 0, n = 0; if (hipGetDevice(&d) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess) n = 0; return n > 0 ? n : 256; }();
  const bool ok256 = a->dtype == JEN1_BF16 && !force_wm && force_t256 != 0 && (g.n_groups == 1 || (a->group_align >= 256 && a->group_align % 256 == 0));
  int n_big = 0;                                        // columns [0, n_big) on the 256 x 256 form
  if (ok256) {
    const int tm = (a->M + 255) / 256, tn_all = (a->Ntot + 255) / 256;
    if (force_t256 == 1) {
      n_big = a->Ntot;
    } else if (tm * tn_all >= (n_cu * 3) / 4 && a->K >= 512) {
      // whole rounds of big tiles; the last round may stay if it is at least 3/4 full, and a remainder narrower than 256 columns rides along
      const int rounds = (tm * tn_all) / n_cu, tail = tm * tn_all - rounds * n_cu;
      int tn_big = tn_all;
      if (tail > 0 && tail * 4 < n_cu * 3 && rounds >= 1) tn_big = (rounds * n_cu) / tm;
      n_big = tn_big >= tn_all ? a->Ntot : tn_big * 256;
      if (g.n_groups > 1 && n_big < a->Ntot && n_big % a->group_align != 0) n_big -= n_big % a->group_align;      // (cut on a group boundary multiple)
    }
  }
  {
    // the four-stage form (big_gemm_nt_s4_kernel<272>): column counts that are multiples of 272 when that tiling fills its rounds of one
    // tile per CU better than 256-wide tiles (the stacked projection, M = 1024: 256 tiles = exactly one round against 272 = 1.06: 54 -> 47 us;
    // the 256-wide instance of the same kernel measured slower than the two-stage 32 x 32 form everywhere and is not built).
    // JEN1_BGEMM_S4 = 0: never; 272: whenever the shape allows (tuning / parity tests)
    const char* es4 = getenv("JEN1_BGEMM_S4");
    const int force_s4 = es4 ? atoi(es4) : -1;
    if (a->dtype == JEN1_BF16 && !force_wm && force_s4 != 0 && force_t256 != 1 && a->K >= 32 && a->K % 32 == 0 && a->Ntot % 272 == 0 &&
        (g.n_groups == 1 || (a->group_align >= 16 && a->group_align % 16 == 0))) {
      const int tm = (a->M + 255) / 256;
      const long long t272 = (long long)tm * (a->Ntot / 272), t256 = (long long)tm * ((a->Ntot + 255) / 256);
      // rounds of one tile per CU x tile width: which tiling fills the chip better
      const double c272 = (double)((t272 + n_cu - 1) / n_cu) * 272.0, c256 = (double)((t256 + n_cu - 1) / n_cu) * 256.0;
      if (force_s4 == 272 || (t272 >= (n_cu * 3) / 4 && c272 < c256 && a->K >= 512)) {
        BArgs gc = g;
        gc.tiles_m = tm;
        gc.tiles_n = a->Ntot / 272;
        hipLaunchKernelGGL(big_gemm_nt_s4_kernel<272>, dim3(gc.tiles_m * gc.tiles_n), dim3(512), 0, s, gc);
        JEN1_HIP(hipGetLastError());
        return 0;
      }
    }
  }
  if (n_big > 0) {
    BArgs gb = g;
    gb.Ntot = n_big;
    gb.tiles_m = (a->M + 255) / 256;
    gb.tiles_n = (n_big + 255) / 256;
    hipLaunchKernelGGL(big_gemm_nt256_kernel, dim3(gb.tiles_m * gb.tiles_n), dim3(512), 0, s, gb);
    JEN1_HIP(hipGetLastError());
    if (n_big >= a->Ntot) return 0;
    g.n_begin = n_big;
  }
  const int n_rest = a->Ntot - g.n_begin;
  const int t256 = ((a->M + 255) / 256) * ((n_rest + BN - 1) / BN);
  const int wm = force_wm ? (force_wm == 5 ? 4 : force_wm) : ((t256 >= 512 && a->K >= 2048) ? 4 : 2);
  g.tiles_m = (a->M + wm * 64 - 1) / (wm * 64);
  g.tiles_n = (n_rest + BN - 1) / BN;
  const dim3 grid(g.tiles_m * g.tiles_n);
  if (wm == 4 && force_wm == 5) {                       // (tuning: 256 x 128 tiles on two stages)
    hipLaunchKernelGGL((big_gemm_nt_kernel<bf16_t, 4, 2>), grid, dim3(512), 0, s, g);
  } else if (wm == 4) {
    if (a->dtype == JEN1_F32) hipLaunchKernelGGL((big_gemm_nt_kernel<float, 4, 3>), grid, dim3(512), 0, s, g);
    else hipLaunchKernelGGL((big_gemm_nt_kernel<bf16_t, 4, 3>), grid, dim3(512), 0, s, g);
  } else {
    if (a->dtype == JEN1_F32) hipLaunchKernelGGL((big_gemm_nt_kernel<float, 2, 2>), grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL((big_gemm_nt_kernel<bf16_t, 2, 2>), grid, dim3(256), 0, s, g);
  }
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_big_gemm_conv(const void* x, const void* w, const float* bias, const void* residual, void* y, int B, int T_in, int T_out, int ci,
                                  int co, int taps, int stride, int pad, int tap_rev, int ld_x, int ld_w, int w_tap_stride, int ld_y,
                                  const int32_t* shift_b, int div, void* stream) {
  JEN1_CHECK(x && w && y && B >= 1 && T_in >= 1 && T_out >= 1 && taps >= 1 && stride >= 1, "big_gemm_conv: bad arguments");
  JEN1_CHECK(ci >= 8 && ci % 8 == 0, "big_gemm_conv: the input channels (%d) must be a multiple of 8 (the columns [ci, pitch) are not read)", ci);
  JEN1_CHECK(co >= 4 && co % 4 == 0 && ld_x >= ci && ld_w >= ci && ld_y >= co && ld_x % 8 == 0 && ld_w % 8 == 0 && ld_y % 4 == 0,
             "big_gemm_conv: widths / pitches (ci, ld_x, ld_w multiples of 8 elements; co, ld_y of 4)");
  JEN1_CHECK(w_tap_stride >= co * ld_w || taps == 1, "big_gemm_conv: tap matrices overlap");
  const int64_t M = (int64_t)B * T_out, Mx = (int64_t)B * T_in;
  JEN1_CHECK(M < ((int64_t)1 << 22) && Mx * ld_x * 2 < 0x7ffffff0ll && (int64_t)taps * w_tap_stride * 2 < ((int64_t)1 << 31) && M * ld_y * 2 < ((int64_t)1 << 40),
             "big_gemm_conv: operand too large");
  CArgs g;
  memset(&g, 0, sizeof(g));
  g.x = x; g.w = w; g.bias = bias; g.residual = residual; g.y = y;
  g.M = (int)M; g.N = co; g.ldx = ld_x; g.ldw = ld_w; g.ldy = ld_y;
  g.tiles_m = (int)((M + 127) / 128); g.tiles_n = (co + BN - 1) / BN;
  g.taps = taps; g.ci = ci; g.T_out = T_out; g.T_in = T_in; g.stride = stride; g.pad = pad; g.rows_x = (int)Mx; g.tap_rev = tap_rev ? 1 : 0;
  g.w_tap_stride = w_tap_stride; g.inv_T_out = 1.0f / (float)T_out; g.shift_b = shift_b;
  g.div = div > 1 ? div : 1; g.inv_div = 1.0f / (float)g.div;
  hipLaunchKernelGGL(big_gemm_conv_kernel, dim3(g.tiles_m * g.tiles_n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g);
  JEN1_HIP(hipGetLastError());
  return 0;
}

// slices of the reduction of a TN product
static int tn_splits(int tiles, int MT) {
  // slices of the reduction: every slice adds its 128 x 128 partial with float atomics, and the atomics (slices x N x K of them, at
  // ~280 per ns through the L2s) are what a small output costs -- 24 000 rows x 128 x 128 on 375 slices: 27.8 us, 6 of them atomics
  // alone would take.  Cost model  t(s) = a ceil(MT / s) + b s tiles  (a = 0.5 us per 64-row step of a workgroup, b = 0.058 us per
  // tile of atomics): the s with the smallest t, among those that put at least half of the CUs to work
  int splits = 1;
  {
    const int s_min = (128 + tiles - 1) / tiles;
    float best = 1e30f;
    for (int s = s_min < MT ? s_min : MT; s <= MT; ++s) {
      const float t = 0.5f * (float)((MT + s - 1) / s) + 0.058f * (float)s * (float)tiles;
      if (t < best) { best = t; splits = s; }
    }
  }
  static const int force_splits = getenv("JEN1_BGEMM_TN_SPLITS") ? atoi(getenv("JEN1_BGEMM_TN_SPLITS")) : 0;      // tuning
  if (force_splits > 0) splits = force_splits;
  splits = splits < 1 ? 1 : (splits > MT ? MT : splits);
  return splits;
}

static int big_gemm_tn_launch(const void* a, const void* b, float* c, int M, int N, int K, int lda, int ldb, int ldc, float alpha, int store, void* stream);
extern "C" int jen1_big_gemm_tn(const void* a, const void* b, float* c, int M, int N, int K, int lda, int ldb, int ldc, float alpha, void* stream) {
  return big_gemm_tn_launch(a, b, c, M, N, K, lda, ldb, ldc, alpha, 0, stream);
}
extern "C" int jen1_big_gemm_tn_store(const void* a, const void* b, float* c, int M, int N, int K, int lda, int ldb, int ldc, float alpha, void* stream) {
  return big_gemm_tn_launch(a, b, c, M, N, K, lda, ldb, ldc, alpha, 1, stream);
}
static int big_gemm_tn_launch(const void* a, const void* b, float* c, int M, int N, int K, int lda, int ldb, int ldc, float alpha, int store, void* stream) {
  JEN1_CHECK(a && b && c && M >= 1 && N >= 1 && K >= 1, "big_gemm_tn: bad arguments");
  JEN1_CHECK(lda >= N && ldb >= K && lda % 8 == 0 && ldb % 8 == 0 && N % 8 == 0 && K % 8 == 0, "big_gemm_tn: widths and pitches must be multiples of 8 elements");
  JEN1_CHECK((N % 128 == 0 || lda >= ((N + 127) / 128) * 128) && (K % 128 == 0 || ldb >= ((K + 127) / 128) * 128),
             "big_gemm_tn: a partial last column tile must still lie inside the row pitch");
  JEN1_CHECK((int64_t)M * lda * 2 < ((int64_t)1 << 31) && (int64_t)M * ldb * 2 < ((int64_t)1 << 31), "big_gemm_tn: operand too large for 31-bit offsets");
  TArgs g;
  memset(&g, 0, sizeof(g));
  g.a = a; g.b = b; g.c = c; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.alpha = alpha;
  g.tiles_n = (N + 127) / 128;
  g.tiles_k = (K + 127) / 128;
  const int tiles = g.tiles_n * g.tiles_k, MT = (M + 63) / 64;
  const int splits = store ? 1 : tn_splits(tiles, MT);      // (store: every workgroup owns its output tile; no atomics, no zero-fill before)
  g.store = store;
  g.mt_per_split = (MT + splits - 1) / splits;
  g.splits = (MT + g.mt_per_split - 1) / g.mt_per_split;
  g.main_blocks = tiles * g.splits;
  hipLaunchKernelGGL(big_gemm_tn_kernel, dim3(g.main_blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g);
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_big_gemm_tn_conv(const void* dy, const void* x, float* gw, float* gb, int B, int T_out, int T_in, int co, int ci, int taps,
                                     int stride, int pad, int ld_dy, int ld_x, float alpha, const int32_t* shift_b, void* stream) {
  JEN1_CHECK(dy && x && gw && B >= 1 && T_out >= 1 && T_in >= 1 && co >= 8 && ci >= 8 && taps >= 1 && stride >= 1, "big_gemm_tn_conv: bad arguments");
  // (ci itself may be ragged, e.g. 257 latent + context channels: its last 8-channel chunk is read up to the pitch and only the real
  // channels are added to gw)
  JEN1_CHECK(co % 8 == 0 && ci >= 1 && ld_dy >= co && ld_x >= ((ci + 7) / 8) * 8 && ld_dy % 8 == 0 && ld_x % 8 == 0,
             "big_gemm_tn_conv: co and the pitches must be multiples of 8 elements, ld_x >= ci rounded up to 8");
  JEN1_CHECK(taps <= 16, "big_gemm_tn_conv: at most 16 taps (a column tile holds an 8-channel chunk of every tap)");
  // (one tap, ragged ci: the last column tile reads past the row's end into the next row -- finite values that only reach columns
  // k >= ci, which the epilogue drops; past the last row the descriptor returns zeros)
  JEN1_CHECK(co % 128 == 0 || ld_dy >= ((co + 127) / 128) * 128, "big_gemm_tn_conv: a partial last column tile of dy must still lie inside the row pitch");
  const int64_t M = (int64_t)B * T_out, Mx = (int64_t)B * T_in;
  JEN1_CHECK(M < ((int64_t)1 << 22) && M * ld_dy * 2 < ((int64_t)1 << 31) && Mx * ld_x * 2 < 0x7ffffff0ll, "big_gemm_tn_conv: operand too large");
  TArgs g;
  memset(&g, 0, sizeof(g));
  g.a = dy; g.b = x; g.c = gw; g.M = (int)M; g.N = co; g.K = taps * ci; g.lda = ld_dy; g.ldb = ld_x; g.ldc = ci * taps; g.alpha = alpha;
  g.conv = 1; g.taps = taps; g.ci = ci; g.T_out = T_out; g.T_in = T_in; g.stride = stride; g.pad = pad; g.rows_b = (int)Mx;
  g.inv_T_out = 1.0f / (float)T_out; g.bias_grad = gb; g.shift_b = shift_b;
  g.tiles_n = (co + 127) / 128;
  g.cpt = taps > 1 ? 16 / taps : 16;
  g.tiles_k = taps > 1 ? ((ci + 7) / 8 + g.cpt - 1) / g.cpt : (ci + 127) / 128;
  const int tiles = g.tiles_n * g.tiles_k, MT = (int)((M + 63) / 64);
  int splits = tn_splits(tiles, MT);
  g.mt_per_split = (MT + splits - 1) / splits;
  g.splits = (MT + g.mt_per_split - 1) / g.mt_per_split;
  g.main_blocks = tiles * g.splits;
  if (gb) {
    int nbk = (int)(M / 1536);
    nbk = nbk < 1 ? 1 : (nbk > 32 ? 32 : nbk);
    g.bias_rows = (int)((M + nbk - 1) / nbk);
    g.bias_blocks = (int)((M + g.bias_rows - 1) / g.bias_rows);
  }
  hipLaunchKernelGGL(big_gemm_tn_kernel, dim3(g.main_blocks + g.bias_blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g);
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_standardize_rows(const float* x, void* y, int rows, int C, int ldx, int ldy, float eps, int dtype, void* stream) {
  JEN1_CHECK(x && y && rows >= 1 && C >= 4 && C % 4 == 0 && ldx >= C && ldy >= C && ldx % 4 == 0 && ldy % 4 == 0, "standardize_rows: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid((rows + 3) / 4);
  if (dtype == JEN1_F32) hipLaunchKernelGGL(standardize_rows_kernel<float>, grid, dim3(256), 0, s, x, (float*)y, rows, C, ldx, ldy, eps);
  else if (dtype == JEN1_BF16) hipLaunchKernelGGL(standardize_rows_kernel<bf16_t>, grid, dim3(256), 0, s, x, (bf16_t*)y, rows, C, ldx, ldy, eps);
  else return jen1_set_error("standardize_rows: bad dtype");
  JEN1_HIP(hipGetLastError());
  return 0;
}

extern "C" int jen1_kv_fixed_fill(const void* table_dev, int n_layers, const float* mask, int B, int rows, int dtype, void* stream) {
  JEN1_CHECK(table_dev && mask && n_layers >= 1 && B >= 1 && rows >= 1, "kv_fixed_fill: bad arguments");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const dim3 grid(64, n_layers);
  if (dtype == JEN1_F32) hipLaunchKernelGGL(kv_fixed_fill_kernel<float>, grid, dim3(256), 0, s, reinterpret_cast<const FixedEntry*>(table_dev), mask, B, rows);
  else if (dtype == JEN1_BF16) hipLaunchKernelGGL(kv_fixed_fill_kernel<bf16_t>, grid, dim3(256), 0, s, reinterpret_cast<const FixedEntry*>(table_dev), mask, B, rows);
  else return jen1_set_error("kv_fixed_fill: bad dtype");
  JEN1_HIP(hipGetLastError());
  return 0;
}
