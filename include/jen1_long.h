/*
 * jen1_long.h -- C ABI of the sample-resident long-level kernel of libjen1_hip.so (gfx950 / MI355X).
 *
 * The LONG levels of the JEN-1 UNet -- to_in, levels 0..2 of the down path and levels 2..0 + to_out of the up path
 * (T' = 1500 / 375 / 94 at T = 1500; reference jen1/model/blocks.py:98-145 ConvBlock1d, :168-231 ResnetBlock1d,
 * :540-650 DownsampleBlock1d, :653-764 UpsampleBlock1d, jen1/model/model.py:243-262) -- are a chain of ~39 dependent
 * convolutions with a GroupNorm (a reduction over the whole sample) between any two of them.  One launch per layer
 * costs ~10 us each at 4.6 % of the HBM roofline (profiles/r05_tile_gemm_stage_profile.txt: a serial chain of kernel
 * boundary, argument fetch, statistics round trip, staging, weight ring and epilogue).  Here each half of the chain is
 * ONE launch of `nwg` resident 512-thread workgroups (one per CU) that walk a device-resident list of *phases*, one
 * per convolution:
 *
 *   * every GroupNorm statistic and every convolution is per sample, so a sample is given to a fixed GROUP of
 *     G = nwg / B workgroups -- sample b owns workgroups b, b + B, b + 2B, ...: with B = 8 on the 8 XCDs of an MI355X
 *     (workgroups are dealt round-robin over the XCDs) that is exactly the 32 CUs of XCD b, so a sample's activations,
 *     its statistics exchange and the layer's weights stay in ONE L2;
 *   * within the group, workgroup k of every phase computes the same slice of the sample -- M block k % mblocks (128
 *     GEMM rows) x position tile k / mblocks (tb = ceil(L_out / tiles) positions) -- so the window a workgroup reads is
 *     what it and its two neighbours wrote one phase earlier;
 *   * the 128 x K weight slice of the unit (<= 24 k-steps = the whole slice of a 128-channel k = 3 layer) is requested
 *     into registers right behind the previous unit's stores, BEFORE the dependency wait: the next layer's weights
 *     arrive while the exchange is in flight (what the round-3 tile phases lacked);
 *   * GroupNorm statistics travel as per-unit partial sums: wave w of a unit owns 16 output channels and writes ONE
 *     8-byte (sum, sum of squares) word; the consumer's wave w reads the words of slot w of all units of the sample
 *     (one coalesced load), adds them with a fixed cross-lane tree and the affine pair y = silu(A x + S) of every
 *     channel follows -- no float atomics: the long levels are bit-reproducible;
 *   * activations and partials travel through global memory as 8-byte write-through words that start each step
 *     POISONED (all ones, jen1_deep_poison): the data is its own arrival flag exactly as in jen1_deep.h ("Reserved
 *     word" there applies unchanged), there is no counter, drain or barrier between producer and consumer.
 *
 * The static unit -> workgroup map needs every workgroup of the launch resident at once (like jen1_deep_run_mode with
 * tickets = 0): the caller runs at most one such launch per device at a time.  A dependency wait that exceeds its bound
 * (~170 ms) raises the error word (1 + phase) and releases every waiter instead of hanging.
 *
 * Plain pointers and sizes only; all pointers are device pointers unless noted.
 */
#ifndef JEN1_LONG_H
#define JEN1_LONG_H

#include <stdint.h>

#include "jen1_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define JEN1_LONG_THREADS 512
#define JEN1_LONG_DESC_BYTES 512      /* one phase descriptor: two dwords per lane of a wave */
#define JEN1_LONG_MAX_PHASES 128
#define JEN1_LONG_MAX_NF 6            /* 16-position fragments per unit: a tile has at most 96 positions */
#define JEN1_LONG_BM 128              /* GEMM rows per unit: 8 waves x one 16-row M tile */
#define JEN1_LONG_MAX_KS 64           /* k-steps (32 input channels of one tap each) of a phase */

typedef struct jen1_long_src {
  const void* x;              /* [B][L_in][ld] in the launch dtype */
  int32_t ld;
  int32_t C;                  /* channels read (multiple of 32; 0 = source absent) */
} jen1_long_src;

/* One convolution of a long level.  Field order is ABI (the kernel reads the descriptor as 128 dwords). */
typedef struct jen1_long_phase {
  /* sources 0 / 1: the channel concat the taps run over (normalised when pro_mode != NONE: ConvBlock1d over [x, skip * 2^-1/2],
   * blocks.py:137-145, :732-734); sources 2 / 3: raw extra K segments at row shift 0 (a 1x1 shortcut riding on the block's
   * second conv, blocks.py:229-231) */
  jen1_long_src src[4];
  /* GroupNorm statistics of sources 0 / 1.  st_entries = 0: totals [B][32][2] (sum, sumsq per fine group of ld / 32 channels,
   * complete before the launch: jen1_conv_args.gn_stats*).  st_entries > 0: partials [B][8][st_entries][2] written by the phase
   * that produced the source -- slot s of entry e holds channels 16 (s + 8 ((e % st_mblocks) * 8 % st_nsub / 8)) .. + 16 */
  const float* st[2];
  const void* w;              /* packed weight [k-step][M/16][64 lanes][8] (jen1_amd/packing.py: pack_gemm_weight) */
  const float* bias;          /* [out_C] or NULL */
  const void* residual;       /* same row mapping as y, or NULL */
  void* y;
  float* out_part;            /* [B][8][out_entries][2] partial sums of this phase's output, or NULL */
  const float* p1;            /* y = xhat * p1[row * p_ld + c] + p2[row * p_ld + c]: GroupNorm gamma / beta (p_ld = 0) or the */
  const float* p2;            /* fused GroupNorm-FiLM table gamma (scale + 1) / beta (scale + 1) + shift (blocks.py:141-143) */
  const int32_t* film_row;    /* p_ld > 0: table row of sample b (NULL: b) ... */
  const int32_t* film_step;   /* ... or one row for everybody, read from this device scalar */
  uint32_t w_bytes;
  int32_t st_entries[2], st_mblocks[2], st_nsub[2];
  int32_t live_mask;          /* bits 0..3: source k was produced inside THIS launch (polled); bits 4, 5: st[0], st[1]; bit 8: the residual */
  int32_t B, G;               /* samples; workgroups per sample */
  int32_t L_in, L_out, stride, taps, pad_left;
  int32_t tb, tiles_t, mblocks, NF, rows_in;      /* unit k: M block k % mblocks, positions [k / mblocks * tb, + tb) */
  int32_t cmain, call, pitch, kch, KS, MT;        /* c0 + c1; + extra channels; LDS row pitch; cmain / 32; k-steps; M / 16 */
  int32_t pro_mode, gn_groups, gn_cpg, p_ld;
  int32_t out_C, ps_f, ps_off, L_y, y_brows, y_row0, ld_y, ld_res;
  int32_t out_entries;        /* tiles_t * mblocks: units of this phase per sample */
  int32_t st_gran[2];         /* channels per statistics entry of source k as the consumer sees them (16, or ld / 32 for totals) */
  int32_t tab_off, st_off, lds_bytes;             /* LDS byte offsets behind the staged tile; total */
  float inv_count, gn_eps, src1_scale, inv_vpr;
  /* log2 of mblocks, out_C, gn_cpg and st_gran[k] (all powers of two in a phase that jen1_long_phase_conv accepts; gn_cpg / st_gran only
   * when gn_groups > 1): the unit's index arithmetic is shifts */
  int32_t mb_shift, outc_shift, cpg_shift, gran_shift[2];
  /* element offset into the staged tile of k-step ks: tap * pitch + 32 * chunk in (tap, chunk) order, then the extra chunks at the
   * centre row (pad_left * pitch + cmain + 32 j) */
  uint16_t koff[JEN1_LONG_MAX_KS];
  int32_t reserved_[5];
} jen1_long_phase;

/* geometry of a phase with M GEMM rows and L_out GEMM positions on G workgroups per sample: mblocks = M / 128, tiles = G / mblocks
 * position tiles of tb = ceil(L_out / tiles) positions.  Returns 0, or non-zero (jen1_last_error) when it does not fit
 * (M not a multiple of 128, more M blocks than workgroups, a tile of more than 96 positions). */
int jen1_long_geometry(int M, int L_out, int G, int* mblocks, int* tiles_t, int* tb);

/* HOST helper: fill *out (host memory, sizeof = JEN1_LONG_DESC_BYTES) for one layer.  Reads from `a` what jen1_deep_phase_tile reads:
 * x0 / x1 / c0 / c1 / ld0 / ld1 / src1_scale, taps / stride / pad_left, seg[0..nseg) as raw extra K segments at row shift 0, the
 * packed weight, bias, residual, y and its row mapping incl. the sub-pixel form of ConvTranspose1d, pro_mode NONE / GN / GN_SILU
 * with gn_* and the FUSED GroupNorm-FiLM table in `film`, live_mask (bit 0 x0, bit 1 x1 (or seg[0] when c1 = 0), following bits the
 * extra segments, bit 8 the residual), dtype JEN1_F32 / JEN1_BF16, B / L_in / L_out.
 * G: workgroups per sample.  st0 / st1 with (entries, mblocks, nsub): statistics of x0 / x1 in the partial form described above,
 * or entries = 0 for totals; st_live: bit k = the partials of source k are written inside this launch.  out_part (or NULL) receives
 * this phase's own partials ([B][8][jen1_long_geometry tiles * mblocks][2] floats).
 * Returns non-zero (message in jen1_last_error) when the layer does not fit the kernel. */
int jen1_long_phase_conv(const jen1_conv_args* a, int G, const float* st0, int st0_entries, int st0_mblocks, int st0_nsub, const float* st1,
                         int st1_entries, int st1_mblocks, int st1_nsub, int st_live, float* out_part, jen1_long_phase* out);

/* dynamic LDS bytes / units per sample of a filled descriptor (host helpers for bindings that treat the descriptor as opaque bytes) */
int jen1_long_phase_lds(const jen1_long_phase* p);
int jen1_long_phase_units(const jen1_long_phase* p);

/* enqueue the launch: descs_dev = n_phases descriptors (device copy of what jen1_long_phase_conv filled), B samples on nwg workgroups
 * (sample b on workgroups b, b + B, ...; G = nwg / B must equal the G the phases were built for), lds_bytes = the largest
 * jen1_long_phase.lds_bytes, err = one uint32 (zeroed once by the caller: the first time-out of any launch since then stays visible).
 * ticket = NULL: the static unit -> workgroup map (fastest; correct only while every workgroup of the launch is resident: the caller
 * runs at most ONE static persistent launch per device at a time, like jen1_deep_run_mode with tickets = 0).  ticket = one uint32 that
 * is ZERO when the launch starts: units are handed out by ticket, the launch makes progress with any number of resident workgroups
 * and may share the GPU with other persistent launches.
 * local = 1 (B a multiple of 8, nwg a multiple of B and of 8; `ticket` must point at 16 ZERO words): the static form with the groups
 * formed from where the workgroups actually run -- every workgroup reads its XCD (HW_REG_XCC_ID) and takes a number on it (words
 * ticket[8 + xcd]); XCD x hosts samples x, x + 8, ... -- so that a sample's group sits on ONE XCD by construction and the phases'
 * outputs can be stored PLAIN: they stay in that XCD's L2, where the readers' L1-bypassing polls find them (a hand-off costs ~0.3 us
 * instead of ~0.5 - 0.6 written through), and are written back when the kernel ends.  HIP promises nothing about workgroup -> XCD
 * placement; what this form needs is only that every XCD holds nwg / 8 workgroups of the launch, which one workgroup per CU and full
 * residency imply.  A workgroup that finds its XCD full raises the error word (0x40000000 | workgroup).
 * Every tensor and every partial array the phases write must be poisoned (jen1_deep_poison) between the previous launch's last
 * reader and this launch.  Capturable. */
int jen1_long_run(const void* descs_dev, int n_phases, int B, uint32_t* err, uint32_t* ticket, int nwg, int lds_bytes, int dtype, int local,
                  void* stream);

/* one launch of the kernel's shape (nwg workgroups x 512 threads, one per CU): out_dev[i] = the XCD (HW_REG_XCC_ID) workgroup i ran on */
int jen1_long_census(int* out_dev, int nwg, void* stream);

/* tuning builds (-DJEN1_LONG_PROFILE): per (phase, workgroup) 8 stamps of the 100 MHz counter */
int jen1_long_debug_buffer(void* p);

#ifdef __cplusplus
}
#endif
#endif /* JEN1_LONG_H */
