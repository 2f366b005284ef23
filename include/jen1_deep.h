/*
 * jen1_deep.h -- C ABI of the persistent deep-level kernel of libjen1_hip.so (gfx950).
 *
 * The levels of the JEN-1 UNet with T' <= 24 positions (down 3..8, bottleneck, up 0..5:
 * reference jen1/model/model.py:246-259, jen1/model/blocks.py:540-830) hold 288 M of the
 * 296 M parameters and are a chain of ~165 tiny dependent layers.  Run as one launch per
 * layer, every layer pays a kernel boundary, an argument fetch and a cold prefetch ring.
 * Here the whole chain is ONE launch of `nwg` resident workgroups (one per CU, 512
 * threads) that walk a device-resident list of *phases*:
 *
 *   JEN1_DEEP_GEMM   conv / transposed conv / linear as a K-segment GEMM (same algebra as
 *                    jen1_conv_gemm's direct mode: _Conv1d blocks.py:34-53, Upsample1d :69-95,
 *                    ResnetBlock1d :219-231, Attention projections :427-429, FeedForward
 *                    :440-446) whose prologue -- GroupNorm(+FiLM)(+SiLU), blocks.py:137-145,
 *                    :509 -- is applied by the consuming workgroup while it stages the (tiny)
 *                    activation tile in LDS: the group statistics are computed there, in a
 *                    fixed order (no float atomics, bit-reproducible);
 *   JEN1_DEEP_ATTN   AttentionBase.forward's math path (blocks.py:355-380) with the deferred
 *                    LayerNorm finish of jen1_attention_fin; LayerNorm row statistics are
 *                    recomputed by the consumer from the rows themselves;
 *   JEN1_DEEP_STATS  GroupNorm fine-group sums of the last tensor of the chain, for the
 *                    launch-per-layer kernels that consume it after the persistent launch.
 *   JEN1_DEEP_TILE   the same convolution algebra for the LONG levels (T' = 1500 / 375 at
 *                    T = 1500: ConvBlock1d blocks.py:137-145 over [x, skip], strided down
 *                    convs, sub-pixel up convs, 1x1 shortcut as extra K segments): a unit is
 *                    ALL output channels (or a block of 128 / 256 of them) x a tile of 16..64
 *                    positions of one batch element, like jen1_conv_gemm's T* tiles.  A batch
 *                    element is far too long to be re-read by every consumer, so GroupNorm
 *                    statistics travel as per-tile partial sums: every producing unit writes
 *                    (sum, sumsq) of its tile per statistics group, the consumer adds the
 *                    partials of its batch element in a fixed order (bit-reproducible, no
 *                    float atomics).  The partials start poisoned like every tensor of the
 *                    launch, so reading them complete IS the wait for the whole batch element.
 *
 * A unit of a phase (16 output rows x the positions of a few batch elements; one
 * (batch element, head, 32-query chunk)) is done by one workgroup.  Results travel through
 * global memory as 8-byte write-through (sc1) words; the tensors a phase produces are
 * POISONED (all bytes 0xFF, jen1_deep_poison) before the launch, and a consumer wave repeats
 * its L1-bypassing (sc1) loads until no word is the sentinel: the data is its own arrival
 * flag, there is no counter, no drain and no barrier between producer and consumer
 * (1.2 - 1.35 us per all-to-all stage on 256 workgroups against 3.05 us for the counter
 * protocol of round 2: tools/microbench/flagchain.hip).  Weight slices are requested BEFORE
 * the dependency wait, so their HBM latency hides behind the exchange.
 *
 * Reserved word (contract of the protocol).  The all-ones 8-byte word -- four bf16 or two
 * float32 NaNs with the sign bit and every mantissa bit set -- means "not stored yet" and is
 * never delivered as data.  No finite value encodes as it, and every store of the launch
 * breaks exactly that pattern (it clears the lowest payload bit of the word's first element:
 * still a NaN, no longer the sentinel), so NaN / Inf activations (NaN weights, overflow) flow
 * through the program as NaNs like they do through the reference: they cannot make a
 * consumer wait.  A consumer that still does
 * not see its operands complete within JEN1_DEEP_POLL_LIMIT polls (~170 ms: a lost producer,
 * a launch that is not fully resident next to a static schedule) gives up, writes 1 + phase
 * to the error word (jen1_deep_run_err) and releases every other waiter; the results of
 * that launch are garbage, the host raises Jen1HipError and clears the word, and the next
 * launch is clean (tests/test_gpu_deep.py::test_nan_activations_flow_through_the_launch,
 * ::test_time_out_is_reported_and_cleared).
 *
 * Plain pointers and sizes only; all pointers are device pointers unless noted.
 */
#ifndef JEN1_DEEP_H
#define JEN1_DEEP_H

#include <stdint.h>

#include "jen1_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define JEN1_DEEP_GEMM 0
#define JEN1_DEEP_ATTN 1
#define JEN1_DEEP_STATS 2
#define JEN1_DEEP_TILE 3

#define JEN1_DEEP_MAX_SRC 4
#define JEN1_DEEP_MAX_SEG 12
#ifndef JEN1_DEEP_THREADS
#define JEN1_DEEP_THREADS 512      /* 8 waves: 256 registers per lane hold the weight ring and the staging vectors */
#endif
#define JEN1_DEEP_BLOB_BYTES 4096   /* device image of one phase: descriptor + per-wave K-chunk lists */
#define JEN1_DEEP_MAX_PHASES 256
#ifndef JEN1_DEEP_SHARDS
#define JEN1_DEEP_SHARDS 32         /* arrival counters per phase (<= 63: one wave polls them; 8 / 16 / 32 / 48: 1113 / 1070 / 1039 / 1037 us) */
#endif
#ifndef JEN1_DEEP_SHARD_WORDS
#define JEN1_DEEP_SHARD_WORDS 32    /* uint32 words between two shard counters (128 B apart: one line each; 64 B apart is slower) */
#endif

/* one source tensor of a GEMM phase: channels [coff, coff + C) of the staged tile */
typedef struct jen1_deep_src {
  const void* x;              /* [B][L_in][ld] in the launch dtype */
  int32_t ld;
  int32_t C;                  /* channels read (multiple of 32) */
  int32_t coff;               /* first channel of this source in the staged tile */
  float scale;                /* multiplies the raw values (skip scale of the up path, blocks.py:682,732-734) */
} jen1_deep_src;

/* one K segment: chunks [gend_prev, gend) of the flat packed weight multiply the staged channels
 * coff + 32*(g - gend_prev) .. of input row  q*stride + shift  (zero outside [0, L_in)) */
typedef struct jen1_deep_seg {
  int32_t coff, shift, gend, reserved;
} jen1_deep_seg;

/* the part of a phase a GEMM unit reads as ONE batch of scalar loads (384 bytes = 6 x 64): scheduling, GEMM geometry,
 * prologue / epilogue scalars and the source table.  Field order is ABI. */
typedef struct jen1_deep_hot {
  /* ---- scheduling ---- */
  int32_t kind;               /* JEN1_DEEP_* */
  int32_t n_units;
  int32_t rot;                /* unit u runs on workgroup (u + rot) % nwg          (set by jen1_deep_link) */
  int32_t dep;                /* phase whose completion this phase waits for, -1 = none  (jen1_deep_link) */
  int32_t dep_units;          /* its n_units                                             (jen1_deep_link) */
  int32_t lds_bytes;          /* dynamic LDS a unit of this phase needs */
  int32_t dtype;
  int32_t ntrips;             /* GEMM: trips of JEN1_DEEP_MAXV vectors per lane that stage the normalised part (1 for short rows) */
  /* ---- GEMM ---- */
  const void* w;              /* flat packed weight [chunk][M/16][64 lanes][8] */
  const float* bias;          /* [out_C] or NULL */
  const void* residual;       /* same row mapping as y, or NULL */
  void* y;
  const float* p1;            /* y = xhat * p1[row * p_ld + c] + p2[row * p_ld + c]: GroupNorm gamma / beta (p_ld = 0), or the */
  const float* p2;            /* GroupNorm-FiLM table  gamma (scale + 1) / beta (scale + 1) + shift  (blocks.py:141-143) */
  const int32_t* film_row;    /* p_ld > 0: table row of batch element b (NULL: b) ... */
  const int32_t* film_step;   /* ... or one row for everybody, read from this device scalar */
  uint32_t w_bytes;
  int32_t G, MT, mt_split, g_split;
  int32_t B, L_in, L_out, stride, nb, NF, groups_n;   /* groups_n = ceil(B / nb) */
  int32_t nsrc, nseg, Ctot, pitch, R;                 /* R = nb * L_in staged rows (+1 zero row) */
  int32_t pro_mode, norm_C, gn_groups, gn_cpg, gran;
  float inv_count, gn_eps, inv_vpr, inv_Lin, inv_Lout, inv_groups;
  int32_t p_ld;
  int32_t lvn, lvr;           /* lvr: log2 of the 8-channel vectors per row of the raw part of the staged tile (lvn unused) */
  int32_t out_C, ps_f, ps_off, L_y, y_brows, y_row0, ld_y, ld_res, act, y_f32;
  int32_t part_off, stat_off, red_off;                /* LDS byte offsets behind the tile */
  jen1_deep_src src[JEN1_DEEP_MAX_SRC];
  int32_t red_bytes;          /* size of one K-reduction scratch at red_off (a second one follows it when mrep > 1) */
  /* staged tile: every batch element owns Lp = Hb + L_in + Ha rows (zero halo rows before / after: a conv tap is a plain row
   * offset), a block of zero rows behind them serves the columns that do not exist (zrow: its centre row) */
  int32_t Lp, Hb, zrow, Rtot;
  /* GroupNorm statistics without LDS: each (batch element of the unit, group) pair owns 2^lS consecutive lanes, a lane owns
   * one 8-channel column of the group (2^lvpg columns per row) */
  int32_t lS, lvpg, lgroups;
  /* a unit finishes mrep consecutive 16-row M tiles from ONE staged tile (jen1_deep_link: phases with more units than
   * workgroups; divides MT and mt_split): n_units = (MT / mrep) * groups_n */
  int32_t mrep;
  int32_t live_mask;          /* bit k: source k was produced inside this launch, bit 8: the residual (jen1_conv_args.live_mask) */
  const float* wscale;        /* JEN1_FP8: [M] scale of the e4m3 weight rows (jen1_conv_args.w_scale), else NULL */
  /* column chunks: a batch element with more than 64 output positions (the 94-position level at T = 1500) is cut into n_chunks
   * chunks of Lc positions; a unit computes ONE chunk but stages -- and normalises over -- the whole batch element (nb = 1).
   * Without chunking n_chunks = 1 and Lc = L_out.  n_units = (MT / mrep) * groups_n * n_chunks; inv_Lout is 1 / Lc. */
  int32_t n_chunks, Lc;
  float inv_nchunks;
  int32_t pad_hot_;
} jen1_deep_hot;

/* JEN1_DEEP_TILE: what a tile unit needs beyond the shared fields of jen1_deep_hot (w / bias / residual / y and its row mapping,
 * p1 / p2 / p_ld / film_row / film_step, pro_mode, gn_*, B / L_in / L_out / stride, src[], Ctot, pitch, live_mask) */
typedef struct jen1_deep_tile {
  /* GroupNorm statistics of the two normalised sources: st_tiles = 0: totals [B][32][2] (sum, sumsq per fine group of
   * ld / 32 channels: jen1_conv_args.gn_stats*, complete before the launch); st_tiles > 0: partials [B][st_tiles][st_nfg][2]
   * written by a tile phase (bit k of st_live: by a phase of THIS launch, polled) */
  const float* st[2];
  float* out_part;            /* [B][tiles_t * mblocks][out_nfg][2] partial sums of this phase's output, or NULL */
  int32_t st_tiles[2], st_nfg[2];
  int32_t st_live;
  int32_t cmain;              /* c0 + c1: the channels the taps run over (the extra segments follow at row shift 0) */
  int32_t taps, pad_left;
  int32_t tb, tiles_t;        /* positions per tile (multiple of 16 except the last tile's tail), tiles per batch element */
  int32_t BM, mblocks, MF;    /* output rows per unit, units per tile, 16-row M tiles per wave (1 or 2) */
  int32_t NF, KS, kch, rows_in;
  int32_t out_nfg, out_cps;   /* statistics groups of the output and channels per group (multiple of 16) */
  int32_t tab_off, red_off;   /* LDS byte offsets behind the staged tile: affine tables, reduction scratch */
  float inv_tiles_t, inv_bt, inv_cpg, inv_out_cps;
  float inv_cps[2];           /* 1 / channels per statistics group of source k */
  int32_t pad_[2];
} jen1_deep_tile;

typedef struct jen1_deep_phase {
  jen1_deep_hot h;
  jen1_deep_seg seg[JEN1_DEEP_MAX_SEG];
  /* ---- attention ---- */
  const void* q;              /* [B][Nq][ldq]; LayerNorm statistics are taken over its columns [0, ln_C) */
  const void* k;
  const void* v;
  void* out;
  const int32_t* kv_row;
  const void* kv_extra;
  const int32_t* extra_row;
  const int32_t* extra_step;
  const float* ln_u;
  const float* ln_b;
  int32_t ld_extra, kx_off, vx_off, H, d, Nq, Nk, ldq, q_off, ldkv, k_off, v_off, ldo, causal;
  int32_t ln_C, fin_q, fin_kv, kv_live, nqc, log2_vpr, vsep, pad1_;   /* kv_live: K/V were produced inside this launch (sc1 loads) */
  float scale, ln_eps, inv_H, inv_nqc;
  /* ---- stats ---- */
  const void* sx;             /* [B][L][ld] */
  float* sstats;              /* [B][32][2] */
  int32_t sL, sld, scpf, sgran;
  /* ---- tile ---- */
  jen1_deep_tile tl;
} jen1_deep_phase;

/* sizeof(jen1_deep_phase), for host bindings that treat it as opaque bytes */
int jen1_deep_phase_size(void);
/* 1 when the library was built with -DJEN1_DEEP_CHUNKS (column chunks of jen1_deep_hot.n_chunks: levels of more than 64 positions as
 * GEMM phases).  Off in the default build: measured at T = 1500 the 94-position level inside the launch gives 765 against 777 steps/s,
 * and the chunk decode alone costs every GEMM unit of every plan ~50 ns (8 us per launch). */
int jen1_deep_has_chunks(void);

/* HOST helpers: fill *out (host memory) for one layer.  Return non-zero (message in jen1_last_error) when the layer does
 * not fit the persistent kernel (LDS, register-resident staging vectors, unsupported option); the caller then keeps the
 * level on the launch-per-layer path.
 *
 * jen1_deep_phase_conv reads from `a`: x0/x1/c0/c1/ld0/ld1/src1_scale (the normalised or raw main sources), taps / stride /
 * pad_left, seg[0..nseg) as EXTRA raw K segments appended after the taps, w (flat packed), bias, residual, y and its row
 * mapping, pro_mode (NONE / GN / GN_SILU) with the gn_* fields (statistics pointers are ignored: the consumer computes
 * them), act, m_split / k_split, dtype, B / L_in / L_out.  When `film` is set it must be the FUSED GroupNorm-FiLM table
 * (gamma (scale + 1) at film_off + c, beta (scale + 1) + shift at film_off + film_C + c; rows picked by film_row / film_step
 * as in jen1_conv_args).  nb_max > 0 caps the batch elements per unit. */
int jen1_deep_phase_conv(const jen1_conv_args* a, int nb_max, jen1_deep_phase* out);

int jen1_deep_phase_attention(const void* q, const void* k, const void* v, void* out_t, const int32_t* kv_row, const void* kv_extra,
                              const int32_t* extra_row, const int32_t* extra_step, int ld_extra, int kx_off, int vx_off, int B, int H,
                              int d, int Nq, int Nk, int ldq, int q_off, int ldkv, int k_off, int v_off, int ldo, int causal,
                              float scale, const float* ln_u, const float* ln_b, int ln_C, float ln_eps, int finish_q,
                              int finish_kv, int kv_live /* bit 0: k / v, bit 1: q produced inside this launch */, int dtype, jen1_deep_phase* out);

int jen1_deep_phase_stats(const void* x, float* stats, int B, int L, int ld, int dtype, jen1_deep_phase* out);

/* One layer of a long level as a tile phase.  Reads from `a` what jen1_deep_phase_conv reads (sources, taps / stride / pad_left,
 * seg[0..nseg) as raw extra K segments at row shift 0, packed weight, bias, residual, y and its row mapping incl. the sub-pixel
 * form, pro_mode NONE / GN / GN_SILU with gn_* and the FUSED GroupNorm-FiLM table, live_mask, dtype JEN1_F32 / JEN1_BF16).
 * tb: positions per tile (16, 32, 48 or 64); bm: output rows per unit (128 or 256; M is cut into M / bm blocks).
 * st0 / st1 with their tile counts and group counts: the statistics of x0 / x1 as described at jen1_deep_tile (st_live: bit k = the
 * partials of source k are written inside this launch).  out_part (or NULL) receives this phase's own partials with out_nfg groups
 * per tile ([B][jen1_deep_tile_count][out_nfg][2] floats; out_C / out_nfg must be a multiple of 16). */
int jen1_deep_phase_tile(const jen1_conv_args* a, int tb, int bm, const float* st0, int st0_tiles, int st0_nfg, const float* st1,
                         int st1_tiles, int st1_nfg, int st_live, float* out_part, int out_nfg, jen1_deep_phase* out);
/* tile entries per batch element of the partial array a phase built with (L_out, tb, M, bm) writes: ceil(L_out / tb) * (M / bm) */
int jen1_deep_tile_count(int L_out, int tb, int M, int bm);

/* chain the phases (dep = previous phase, rotation of the unit -> workgroup map) and build the device image:
 * blobs   n_phases * jen1_deep_blob_bytes() bytes (HOST memory): per phase the descriptor followed by the per-wave lists of the
 *         K chunks that can touch a real input row ({chunk index, staged column | row shift << 16}, dealt round-robin);
 * headers n_phases * 16 bytes (HOST memory): {n_units, rot, kind, 0} per phase.
 * Both are then copied to the device by the caller.  Returns the dynamic LDS bytes of the launch, or a negative value. */
int jen1_deep_link(jen1_deep_phase* phases, int n_phases, int nwg, void* blobs, void* headers);
int jen1_deep_blob_bytes(void);

/* Poison the tensors the launch produces: table_dev = n device entries {uint64 pointer, uint64 bytes (multiple of 16)}, one per
 * tensor written by a phase (the caller collects them while it records the phases).  Must run after the previous launch's last
 * reader and complete before the launch starts (same stream: the first node of the step).  Also zeroes sync[0], the launch's ticket
 * counter (units are handed to workgroups by ticket: the launch makes progress with any number of resident workgroups).  Capturable. */
int jen1_deep_poison(const void* table_dev, int n, uint32_t* sync, void* stream);
/* the same launch also zeroes the caller's per-step scratch (zero_bytes at zero_ptr, multiples of 16): the statistics arena reset and
 * the poisoning are one node at the head of the step instead of two */
int jen1_deep_poison_zero(const void* table_dev, int n, uint32_t* sync, void* zero_ptr, int64_t zero_bytes, void* stream);

/* bytes of the synchronisation area: word 0 is the ticket counter (zero when the launch starts: jen1_deep_poison), the rest is unused */
int64_t jen1_deep_sync_bytes(int n_phases);

/* workgroups the launch should use on the current device (one per CU, all resident) */
int jen1_deep_num_workgroups(void);

/* enqueue the persistent launch.  blobs_dev / headers_dev: device copies of what jen1_deep_link produced.  sync: zeroed area
 * of jen1_deep_sync_bytes(n_phases).  After completion sync word jen1_deep_error_word(n_phases) is non-zero if a dependency
 * wait timed out (1 + phase index). */
int jen1_deep_run(const void* blobs_dev, const void* headers_dev, int n_phases, uint32_t* sync, int nwg, int lds_bytes, int dtype,
                  void* stream);
/* the same with the error word outside the synchronisation area: `sync` is zeroed before every launch, `err` (one uint32, zeroed
 * once by the caller) keeps the first time-out of ANY launch since then, so a host that replays the launch many times (a sampling
 * run) checks it once at the end */
int jen1_deep_run_err(const void* blobs_dev, const void* headers_dev, int n_phases, uint32_t* sync, uint32_t* err, int nwg, int lds_bytes,
                      int dtype, void* stream);
int jen1_deep_error_word(int n_phases);
/* the same with the scheduling form chosen by the caller.  tickets = 1 (what jen1_deep_run / jen1_deep_run_err use): units are handed
 * to workgroups by ticket from sync[0]; the launch makes progress with ANY number of resident workgroups, so persistent launches that
 * share the GPU (other streams, other processes) cannot deadlock each other.  tickets = 0: the static unit -> workgroup map, ~9 %
 * faster (889 against 971 us per launch at B = 8, T = 1500), correct only while all nwg workgroups are resident together: for a caller
 * that runs at most ONE such launch per device at a time.  Either way a dependency wait that exceeds its bound (~170 ms) raises the
 * error word instead of hanging. */
int jen1_deep_run_mode(const void* blobs_dev, const void* headers_dev, int n_phases, uint32_t* sync, uint32_t* err, int nwg, int lds_bytes,
                       int dtype, int tickets, void* stream);
/* the same for a program of the given unit kinds (kind_mask: bit k = the program holds phases of kind JEN1_DEEP_<k>).  Two kernels
 * exist: GEMM + attention + statistics phases (what jen1_deep_run_mode launches) and tile + statistics phases; a caller records the
 * long levels and the deep levels as separate programs (one launch each, back to back on the stream). */
int jen1_deep_run_kinds(const void* blobs_dev, const void* headers_dev, int n_phases, uint32_t* sync, uint32_t* err, int nwg, int lds_bytes,
                        int dtype, int tickets, int kind_mask, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* JEN1_DEEP_H */
