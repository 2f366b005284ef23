/*
 * jen1_hip.h -- C ABI of libjen1_hip.so: the MI355X (gfx950) kernels behind the
 * JEN-1 denoiser hot path.
 *
 * The reference (0417keito/JEN-1-pytorch) is pure Python/PyTorch: it has no FFI or
 * operator plug-in interface; the seam is the nn.Module call contract between
 * GaussianDiffusion and UNetCFG1d.  This header is the
 * boundary a maintainer would bind to replace the stock ATen ops that path
 * dispatches: every entry point cites the reference
 * lines whose arithmetic it replaces.  Plain pointers and sizes only: no torch
 * types.  All pointers are DEVICE pointers unless noted; every call enqueues on
 * `stream` (a hipStream_t passed as void*) and returns immediately, so the calls
 * are hipGraph-capturable.  Return value: 0 on success, non-zero on error, with
 * a message available from jen1_last_error().
 *
 * Layout: activations are channel-last  [B][L][Cp]  (Cp = channels padded to a
 * multiple of 32, padding lanes hold zeros) in JEN1_F32 or JEN1_BF16.  The
 * reference layout [B][C][T] float32 appears only at the two ends of the
 * denoiser (jen1_pack_input / jen1_cfg_ddim_step / jen1_unpack_output).
 */
#ifndef JEN1_HIP_H
#define JEN1_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JEN1_F32 0
#define JEN1_BF16 1
/* OCP e4m3 operands for the matrix cores of the persistent deep-level launch (jen1_deep.h) and of jen1_attention_fin: activations
 * stay JEN1_BF16 in memory, packed weights are one byte per element with a float32 scale per output row */
#define JEN1_FP8 2

/* prologue applied to the GEMM's activation operand while it is staged in LDS */
#define JEN1_PRO_NONE 0
#define JEN1_PRO_GN 1        /* GroupNorm (+FiLM)            blocks.py:140-143, :530 */
#define JEN1_PRO_GN_SILU 2   /* GroupNorm (+FiLM) + SiLU      blocks.py:140-144      */
#define JEN1_PRO_LN 3        /* LayerNorm                     blocks.py:427          */
#define JEN1_PRO_SILU 4      /* SiLU only                     blocks.py:156-159      */

#define JEN1_ACT_NONE 0
#define JEN1_ACT_GELU 1      /* exact erf GELU                blocks.py:444, model.py:77-98 */

/* tile configurations of jen1_conv_gemm: BM x BN output tile per 256-thread workgroup.
 * W* = "wide": 4 waves x 16*MF output rows, K not split (activation-heavy levels);
 * S* = "streaming": one 16-row M tile, the 4 waves split K and reduce through LDS (deep levels,
 *      pure weight streaming: 16-row tiles give >= 64 workgroups at C_out = 1024). */
#define JEN1_CFG_W64x64 0
#define JEN1_CFG_W128x64 1
#define JEN1_CFG_S16x64 2
#define JEN1_CFG_S16x32 3
#define JEN1_CFG_S16x16 4
/* T* = "tile": the lean kernel of the long levels (T' >= 64): 4 waves x 16*MF output rows, one batch element per
 *      tile (nb = 1), GroupNorm / SiLU / no prologue, no split-K; BM = all output channels when M <= 256 */
#define JEN1_CFG_T128x64 5
#define JEN1_CFG_T128x32 6
#define JEN1_CFG_T128x16 7
#define JEN1_CFG_T256x32 8
#define JEN1_CFG_T256x16 9
#define JEN1_CFG_T64x64 10
#define JEN1_NUM_CFG 11

/*
 * Fused implicit-GEMM 1-D convolution / linear layer.
 *
 *   y[b, q*ps_f + m/out_C - ps_off, m % out_C] =
 *       rowscale * ( act( sum_{tap, c} W[tap][m][c] * pro(x[b, q*stride + tap - pad_left, c]) + bias ) + residual )
 *
 * with zero padding outside [0, L_in) applied AFTER the prologue, x the channel
 * concatenation of x0 (c0 channels) and x1 (c1 channels, scaled by src1_scale).
 * It replaces, in one launch:
 *   _Conv1d pad + nn.Conv1d                      jen1/model/blocks.py:34-53
 *   Upsample1d / nn.ConvTranspose1d (sub-pixel)  blocks.py:69-95   (ps_f = stride of the transposed conv)
 *   ConvBlock1d GroupNorm -> FiLM -> SiLU -> conv blocks.py:137-145
 *   skip concat + crop                           blocks.py:732-734, utils/module.py:186-204
 *   ResnetBlock1d residual add                   blocks.py:231
 *   nn.Linear with LayerNorm prologue, GELU, residual   blocks.py:427-429, :440-446, :485-488
 *   MappingToScaleShift (SiLU -> Linear)         blocks.py:148-165
 * and accumulates the statistics the NEXT norm layer needs (GroupNorm fine-group
 * sums, LayerNorm row sums) in its epilogue.
 */
/* One K segment of a direct-mode (streaming) GEMM: the K axis of the packed weight is the
 * concatenation of the segments' 32-channel chunks, segment s contributes
 *     sum_{c < 32*kch} W_s[m][c] * x[b, q*stride + shift, c]       (zero outside [0, L_in)).
 * Conv taps (_Conv1d, blocks.py:42-53), the skip concat (blocks.py:732-734) and a ResnetBlock1d
 * 1x1 shortcut folded into its second conv (blocks.py:219-231) are all expressed this way. */
#define JEN1_MAX_SEG 16
typedef struct jen1_conv_seg {
  const void* x;             /* [B][L_in][ld] in the launch dtype */
  int32_t ld;                /* row pitch in elements (>= 32*kch, multiple of 8) */
  int32_t shift;             /* input row = q*stride + shift */
  int32_t kch;               /* 32-channel chunks */
  int32_t reserved;
} jen1_conv_seg;

typedef struct jen1_conv_args {
  const void* x0;            /* [B][L_in][ld0] */
  const void* x1;            /* optional second channel range, [B][L_in][ld1] */
  const void* w;             /* packed by jen1_pack_conv_weight_* (host side), see DESIGN.md */
  const float* bias;         /* [out_C] or NULL */
  const void* residual;      /* same row mapping as y, [..][ld_res] or NULL */
  void* y;
  const float* gn_stats0;    /* [B][32][2] (sum, sumsq) per fine group of x0 */
  const float* gn_stats1;    /* same for x1 */
  const float* gn_gamma;     /* [c0+c1] */
  const float* gn_beta;      /* [c0+c1] */
  const float* film;         /* [rows][film_ld] or NULL: scale at film_off+c, shift at film_off+film_C+c */
  const int32_t* film_row;   /* [B] row of `film` for each batch element, or NULL (row = b) */
  const float* ln_rowstats;  /* [B*L_in][2] (sum, sumsq) over ln_C channels */
  const float* ln_gamma;     /* [c0] or NULL (weights pre-folded) */
  const float* ln_beta;      /* [c0] or NULL */
  const float* row_scale;    /* [B*y_brows] or NULL: multiplies the finished output row */
  float* out_gn_stats;       /* [B][32][2] accumulated with atomics, or NULL */
  float* out_rowstats;       /* [B*y_brows][2] accumulated with atomics, or NULL */
  float* slab;               /* split-K partial sums workspace */
  uint32_t* counters;        /* split-K arrival counters (zero on entry, left zero on exit) */
  int32_t dtype;             /* JEN1_F32 / JEN1_BF16: element type of x0, x1, w, residual, y */
  int32_t B, L_in, L_out;    /* L_out = number of GEMM positions q per batch element */
  int32_t c0, c1, ld0, ld1;
  int32_t taps, stride, pad_left;
  int32_t M;                 /* GEMM rows = out_C * ps_f */
  int32_t out_C, ps_f, ps_off;
  int32_t L_y, y_brows, y_row0, ld_y, ld_res;
  int32_t y_f32;             /* store y as float32 even when dtype is bf16 */
  int32_t pro_mode;
  int32_t gn_groups, gn_cpg, gn_count;
  float gn_eps, src1_scale;
  int32_t film_off, film_C, film_ld;
  int32_t ln_C;
  float ln_eps;
  int32_t act;
  int32_t out_cpf;           /* channels per fine group of out_gn_stats (= padded out_C / 32) */
  int32_t tb, nb;            /* tile = nb batch elements x tb positions (nb*tb <= BN) */
  int32_t kc_stage;          /* 32-channel chunks staged in LDS at a time */
  int32_t splitk;
  int32_t cfg;               /* JEN1_CFG_* */
  int32_t direct;            /* 1: no LDS staging, activation fragments straight from global memory
                                (streaming cfgs only, pro_mode must be JEN1_PRO_NONE, src1_scale == 1) */
  const void* zeros;         /* direct mode: >= (c0+c1) zero elements; padding rows read from here */
  int32_t tiles_t;           /* derived by jen1_conv_gemm (callers leave 0) */
  float inv_tiles_t, inv_tb; /* derived by jen1_conv_gemm */
  const int32_t* film_step;  /* optional device scalar: when non-NULL every batch element uses FiLM row
                                film_step[0] (per-step tables of a sampler; overrides film_row) */
  const float* ln_u;         /* ln_fold: [M] row sums of the (gamma-folded, dtype-rounded) weights */
  int32_t ln_fold;           /* 1: LayerNorm applied in the epilogue instead of a prologue:
                                y = rstd_n * (acc - mean_n * ln_u[m]) + bias   (taps = 1 only; the row
                                statistics of the INPUT rows come from ln_rowstats / ln_C / ln_eps) */
  int32_t nseg;              /* direct mode: > 0 = explicit K segments below (x0/x1/c0/c1/taps/pad_left are
                                then ignored); 0 = segments derived as (tap, source) pairs */
  jen1_conv_seg seg[JEN1_MAX_SEG];
  int32_t m_split;           /* direct mode, 0 = off: a dual-range GEMM that produces two dependent layers in one
                                launch, e.g. [x3 | f] = [x2 + W_o a ; gelu(W_1 W_o a + W_1 x2 + b)]
                                (blocks.py:485-488, :440-446).  Output rows m < m_split sum only the first k_split
                                chunks of K, take the residual and feed out_rowstats; rows >= m_split sum all of K
                                and take `act`.  Multiple of 16. */
  int32_t k_split;           /* 32-channel chunks of K summed by the rows below m_split */
  const float* w_scale;      /* JEN1_FP8 (jen1_deep_phase_conv only): [M] float32, the scale of output row m -- `w` holds e4m3
                                bytes q with W[m][k] = w_scale[m] * q[m][k] in the same fragment order (8 bytes per lane and chunk) */
  int32_t live_mask;         /* jen1_deep_phase_conv only: which operands were produced by EARLIER PHASES OF THE SAME persistent
                                launch (they start poisoned and are polled, jen1_deep.h): bit 0 x0, bit 1 x1 (or seg[0] when c1 = 0),
                                following bits the extra segments in source order, bit 8 the residual */
  int32_t reserved_;
} jen1_conv_args;

int jen1_conv_gemm(const jen1_conv_args* args, void* stream);
/* dynamic LDS bytes jen1_conv_gemm will request for `args` (host helper, no launch) */
int64_t jen1_conv_gemm_lds_bytes(const jen1_conv_args* args);
/* BM / BN of a tile configuration */
int jen1_cfg_bm(int cfg);
int jen1_cfg_bn(int cfg);

/*
 * Normalise (+FiLM) (+SiLU) a small channel-last tensor ONCE, ahead of a streaming GEMM.
 * On the deep levels (T' <= 24) a GEMM has up to 192 M-tile workgroups that would each redo the
 * prologue of the same tiny activation tile; this pre-pass does it once.
 *   mode JEN1_PRO_GN / JEN1_PRO_GN_SILU: GroupNorm over the channel concat [x0, x1*src1_scale]
 *        (+ x*(scale+1)+shift)(+ SiLU)                       blocks.py:140-144, :530, :732-734
 *   mode JEN1_PRO_LN: LayerNorm (gamma/beta optional)           blocks.py:427
 * y: [B][L][c0+c1] in `dtype`.  Statistics come from the producers' epilogues (same layout as
 * jen1_conv_args.gn_stats* / ln_rowstats).
 */
typedef struct jen1_norm_args {
  const void* x0; const void* x1; void* y;
  const float* gn_stats0; const float* gn_stats1;
  const float* gamma; const float* beta;       /* [c0+c1]; may be NULL for LN (standardise only) */
  const float* film; const int32_t* film_row;
  const float* ln_rowstats;
  int32_t dtype, mode, B, L, c0, c1, ld0, ld1, ld_y;
  int32_t groups, cpg, count;
  float eps, src1_scale;
  int32_t film_off, film_C, film_ld;
  const int32_t* film_step;  /* see jen1_conv_args.film_step */
} jen1_norm_args;

int jen1_norm_apply(const jen1_norm_args* args, void* stream);

/*
 * Multi-head attention core, softmax in fp32 with wavefront-shuffle reductions.
 * q: [B][Nq][ldq] (head h at columns q_off + h*d ...), k/v: rows kv_row[b]*Nk .. +Nk of
 * [*][ldkv] at columns k_off / v_off.  Padding keys are NOT masked with -inf: the
 * reference zeroes K and V rows instead (done by the producer through row_scale).
 * Replaces AttentionBase.forward math path, jen1/model/blocks.py:355-380, incl.
 * causal_mask (:315-319).  out: [B][Nq][ldo] at columns h*d.
 * kv_extra / extra_row: when extra_row[b] >= 0 the LAST key/value row (index Nk-1) of batch
 * element b is read from kv_extra[extra_row[b]][kx_off / vx_off + h*d ...] instead: the text
 * tokens' K/V are step-invariant and cached, only the appended time token (model.py:315-316)
 * is projected per step.  extra_step (optional device scalar): rows with extra_row[b] >= 0 read row
 * extra_step[0] of kv_extra instead (per-step table of a sampler).
 */
int jen1_attention(const void* q, const void* k, const void* v, void* out, const int32_t* kv_row,
                   const void* kv_extra, const int32_t* extra_row, const int32_t* extra_step, int ld_extra, int kx_off, int vx_off,
                   int B, int H, int d, int Nq, int Nk, int ldq, int q_off, int ldkv, int k_off, int v_off,
                   int ldo, int causal, float scale, int dtype, void* stream);

/*
 * Same, with a deferred LayerNorm finish: when a launch produced raw = W' x for a LayerNorm-folded projection
 * together with x itself (jen1_conv_args.m_split), the row statistics of x are only complete after that launch, so
 *     q = rstd_row * (raw - mean_row * ln_u[col]) + ln_b[col]            (blocks.py:427-429)
 * is applied here while the operand is staged.  ln_rowstats: [rows][2] (sum, sumsq over ln_C channels) indexed by the
 * operand's row; ln_u / ln_b: indexed by the operand's COLUMN in its tensor (q_off + h*d + c, k_off + ..., v_off + ...).
 * finish_q applies it to Q, finish_kv to K and V (self-attention only: no kv_row / kv_extra, Nq == Nk).
 */
int jen1_attention_fin(const void* q, const void* k, const void* v, void* out, const int32_t* kv_row,
                       const void* kv_extra, const int32_t* extra_row, const int32_t* extra_step, int ld_extra, int kx_off, int vx_off,
                       int B, int H, int d, int Nq, int Nk, int ldq, int q_off, int ldkv, int k_off, int v_off,
                       int ldo, int causal, float scale, const float* ln_rowstats, const float* ln_u, const float* ln_b,
                       int ln_C, float ln_eps, int finish_q, int finish_kv, int dtype, void* stream);

/*
 * [B][C][T] float32 latents + [B][Cc][T] float32 context channels -> channel-last
 * [nrep*B][T][ld] (ld >= C+Cc, padded with zeros), replicated nrep times along the
 * batch (the CFG pair), plus GroupNorm fine-group sums.  Replaces torch.cat at
 * jen1/model/model.py:240 and :332-349.
 */
int jen1_pack_input(const float* x, const float* ctx, void* y, float* gn_stats, int B, int C, int Cc, int T,
                    int ld, int nrep, int dtype, void* stream);

/*
 * The same with the statistics in a FIXED summation order (no float atomics: two runs are bit-identical): every workgroup writes the
 * (sum, sumsq) of its 32 channels x 32 time steps to parts[B][ceil(T / 32)][ld][2]; jen1_gn_stats_from_parts then adds the time blocks
 * of every channel and the channels of every fine group in order and writes gn_stats[nrep * B][32][2] (the layout jen1_pack_input
 * accumulates into).  Two launches; the second one is a few microseconds.
 */
int jen1_pack_input_parts(const float* x, const float* ctx, void* y, float* parts, int B, int C, int Cc, int T, int ld, int nrep, int dtype,
                          void* stream);
int jen1_gn_stats_from_parts(const float* parts, float* gn_stats, int B, int T, int ld, int nrep, void* stream);

/* channel-last [B][T][ld] -> [B][C][T] float32 (the non-CFG exit of UNetCFG1d.forward). */
int jen1_unpack_output(const void* y, float* out, int B, int C, int T, int ld, int dtype, void* stream);

/* (sum, sumsq) over the first C columns of each row of x[rows][ld]  (LayerNorm statistics of the
 * text context, blocks.py:401,427). */
int jen1_row_stats(const void* x, float* stats, int rows, int C, int ld, int dtype, void* stream);

/* Fine-group GroupNorm statistics of x[B][L][ld] in a FIXED summation order: stats[b][fg] = (sum, sumsq) over the L rows and the
 * ld / 32 channels of fine group fg -- what the conv epilogues accumulate with float atomics (out_gn_stats), written instead of
 * accumulated.  The deterministic mode of the engine (Plan(deterministic=True)) runs it behind every producer of a normalised
 * tensor (blocks.py:141: GroupNorm statistics of the block input). */
int jen1_gn_stats(const void* x, float* stats, int B, int L, int ld, int dtype, void* stream);

/*
 * Time features + MLP in float32: LearnedPositionalEmbedding -> Linear -> GELU
 * (utils/module.py:58-79, model.py:84-89 / :286-291).  t: [n] int64 raw timesteps,
 * freq: [half], w: [out][2*half+1], out: [n][out].
 */
int jen1_time_features(const int64_t* t, const float* freq, const float* w, const float* bias, float* out,
                       int n, int half, int out_features, void* stream);
/* the same for float32 times (VDM's continuous t in [0, 1], vdm/vdm.py:44, :98; the reference model promotes any dtype with
 * the same arithmetic, utils/module.py:66-70) */
int jen1_time_features_f32(const float* t, const float* freq, const float* w, const float* bias, float* out,
                           int n, int half, int out_features, void* stream);

/* y[n][out] = act(x[n][in] @ w[out][in]^T + bias), float32 (to_mapping, model.py:75-80). */
int jen1_linear_f32(const float* x, const float* w, const float* bias, float* y, int n, int in_features,
                    int out_features, int act, void* stream);

/*
 * CFG combine + std-rescale (model.py:362-369) fused with model_predictions and the
 * DDIM update (gdm.py:128-131, :212-222).
 *   net: channel-last [nrep*B][T][ld] denoiser output (nrep = 2: cond rows then uncond rows)
 *   x:   [B][C][T] float32 current latents, noise: same shape or NULL
 *   coef: device float[8] = {sqrt_recip, sqrt_recipm1, sqrt_alpha_next, c, sigma, last_step, -, -}
 *   x_out: [B][C][T] float32 next latents;  eps_out / x0_out optional [B][C][T] float32.
 * With nrep = 1 the CFG part is skipped (embedding_scale == 1).
 * coef[5] selects the row kind: 0 DDIM, 1 DDIM's last step (x_next = x0), 2 DDPM posterior row (gdm.py:144-163:
 * {.., coef1, coef2, sd, 2}), 3 VDM row {alpha_t, sigma_t, alpha_next, sigma_next, -, 3} (vdm/vdm.py:52-55: v-prediction,
 * x_pred = alpha x - sigma v, noise_pred = sigma x + alpha v, x_next = alpha' x_pred + sigma' noise_pred; no clamp).
 */
int jen1_cfg_ddim_step(const void* net, const float* x, const float* noise, const float* coef, float* x_out,
                       float* eps_out, float* x0_out, const int32_t* step_idx, int B, int C, int T, int ld, int nrep,
                       float embedding_scale, int scale_cfg, float scale_phi, int objective, int clip_x0, int dtype,
                       void* stream);
/* step_idx (optional device scalar s): coef and noise are tables, row s is used (coef + 8 s, noise + s B C T).
 * jen1_step_advance increments the scalar; it is the last node of a captured sampler step, so replaying the
 * graph S times walks the schedule with no host-side update in between. */
int jen1_step_advance(int32_t* step_idx, void* stream);
/* jen1_cfg_ddim_step with the advance inside: the block that takes the last of the grid's tickets increments step_idx[0] (every thread
 * has read it by then) and leaves *ticket at zero -- one launch fewer per sampler step.  ticket: one uint32, zero on first use. */
int jen1_cfg_ddim_step_adv(const void* net, const float* x, const float* noise, const float* coef, float* x_out, float* eps_out,
                           float* x0_out, int32_t* step_idx, uint32_t* ticket, int B, int C, int T, int ld, int nrep,
                           float embedding_scale, int scale_cfg, float scale_phi, int objective, int clip_x0, int dtype, void* stream);
/* jen1_cfg_ddim_step_adv that also writes the NEXT step's network input, so that a replayed sampler step needs no jen1_pack_input
 * launch at its head (the loop of gdm.py:202-222 feeds x_{t-1} straight back into model.py:240, :332-349): the new latents go, in the
 * compute dtype, into channels [0, C) of rows [nrep * B][T][ld_rows] (the concat-context channels behind them are written once, by
 * jen1_pack_input_parts, and do not change between steps) and their per-channel (sum, sumsq) over each block of 32 time steps into
 * parts [B][ceil(T / 32)][ld_rows][2] -- the layout and summation order of jen1_pack_input_parts, so jen1_gn_stats_from_parts after
 * this call gives bit-identical statistics.  Needs C % 8 == 0 and 16-byte aligned rows; deterministic rows (coef[4] == 0) read no noise. */
int jen1_cfg_ddim_step_pack(const void* net, const float* x, const float* noise, const float* coef, float* x_out, int32_t* step_idx,
                            uint32_t* ticket, void* rows, float* parts, int ld_rows, int B, int C, int T, int ld, int nrep,
                            float embedding_scale, int scale_cfg, float scale_phi, int objective, int clip_x0, int dtype, void* stream);
/* jen1_cfg_ddim_step_pack and the head of the NEXT replayed step in one launch: the blocks behind the step's set every tensor of the
 * next step's persistent launches to the all-ones sentinel and zero the statistics arena (jen1_deep_poison_zero's job, include/jen1_deep.h:
 * poison_table = its n_rows {pointer, bytes} rows, sync / zero_ptr / zero_bytes as there).  A replayed sampler step is then the three
 * persistent launches, this one and jen1_gn_stats_from_parts. */
int jen1_step_tail(const void* net, const float* x, const float* noise, const float* coef, float* x_out, int32_t* step_idx,
                   uint32_t* ticket, void* rows, float* parts, int ld_rows, int B, int C, int T, int ld, int nrep,
                   float embedding_scale, int scale_cfg, float scale_phi, int objective, int clip_x0, int dtype,
                   const void* poison_table, int n_rows, uint32_t* sync, void* zero_ptr, int64_t zero_bytes, void* stream);

/* CFG combine + rescale only: writes the guided denoiser output [B][C][T] float32 (model.py:362-369). */
int jen1_cfg_combine(const void* net, float* out, int B, int C, int T, int ld, float embedding_scale, int scale_cfg,
                     float scale_phi, int dtype, void* stream);

/* zero `bytes` bytes at p on the stream (statistics arena reset; capturable). */
int jen1_memset_zero(void* p, int64_t bytes, void* stream);

/*
 * Optimiser step of the trainer (trainer.py:144-149, train.py:56-60) on flat float32 buffers:
 *   jen1_grad_sqnorm: out[0] += sum(g^2)                       (first half of nn.utils.clip_grad_norm_; zero `out` first)
 *   jen1_adamw_step : g' = g * min(1, max_norm / (sqrt(gnorm_sq[0]) + 1e-6))  (second half; gnorm_sq NULL or max_norm <= 0: no clip)
 *                     then torch.optim.AdamW.step() number `step` (1-based) with decoupled weight decay, in place.
 *                     skip_nonfinite: a non-finite norm leaves p, m, v untouched (GradScaler.step semantics).
 * gnorm_sq is read on the device, so clip + update needs no host synchronisation.
 */
int jen1_grad_sqnorm(const float* g, int64_t n, float* out, void* stream);
/* the same with caller-owned scratch (jen1_grad_sqnorm_scratch_bytes() bytes, 16-byte aligned, zeroed ONCE by the caller; the kernel
 * leaves it ready for the next call): one scratch per optimiser, so optimisers on different streams never share block partials or
 * the arrival ticket.  jen1_grad_sqnorm itself uses one scratch per device: one call at a time. */
int64_t jen1_grad_sqnorm_scratch_bytes(void);
int jen1_grad_sqnorm_ws(const float* g, int64_t n, float* out, void* scratch, void* stream);
int jen1_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                    float weight_decay, int step, const float* gnorm_sq, float max_norm, int skip_nonfinite, void* stream);
/* the same with the step number kept on the device: step_counter[0] = steps TAKEN so far (zeroed by the caller once); the bias
 * corrections use step_counter[0] + 1 and a one-thread node behind the update increments it -- unless skip_nonfinite dropped the
 * step, so a skipped step does not advance the bias correction (torch AdamW under GradScaler.step, trainer.py:146).  No host
 * synchronisation, capturable. */
int jen1_adamw_step_counted(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                            float weight_decay, int32_t* step_counter, const float* gnorm_sq, float max_norm, int skip_nonfinite,
                            void* stream);

/*
 * Large-M matrix-core GEMM (csrc/big_gemm.hip): the cross-attention ``to_kv`` projection over the text context, reference
 * jen1/model/blocks.py:402-407 (``to_kv`` = Linear(1024 -> 2C), bias-free) as Attention.forward applies it at :427-434 --
 * 45.9 % of the as-written FLOPs of a denoiser step.
 *
 *   C_g[orow(m)][n - n0_g] (+)= (alpha * sum_k A[m][k] B[n][k] + bias_g[n - n0_g]) * row_scale[orow(m)]      n0_g <= n < n0_g + N_g
 *
 * A [M][lda] and B [Ntot][ldb] are K-contiguous in the compute dtype (bf16: v_mfma_f32_32x32x16_bf16; f32: exact
 * v_mfma_f32_16x16x4_f32); the rows of B are stacked column GROUPS, each with its own output tensor, bias and pitch (the 13
 * cross-attention layers of a sampling plan: ONE launch projects the text tokens for all of them; LayerNorm's gamma / beta are
 * folded into B and bias at pack time, the standardisation is shared: jen1_standardize_rows).  ``row_scale`` is the padding
 * mask the reference multiplies into k and v (blocks.py:431-434).  orow(m) = (m / rows_in) * rows_out + m % rows_in maps the
 * [B * 128] text rows into the [B][129] rows of a K/V cache (rows_in = 0: orow = m).  K must be a multiple of 64 (bf16) / 32
 * (f32), every n0_g and N_g a multiple of 128 when there is more than one group.  ``groups`` is a DEVICE table; NULL = one
 * group described by the inline fields c / bias / ldc (nothing to copy to the device: usable while a graph is being recorded).
 */
typedef struct jen1_bgemm_group {
  void* c;                 /* output of the group: compute dtype, or float32 when c_f32 */
  const float* bias;       /* [N] or NULL */
  int32_t n0, N, ldc, reserved;
} jen1_bgemm_group;

typedef struct jen1_bgemm_args {
  const void* a;
  const void* b;
  const jen1_bgemm_group* groups;   /* device pointer, n_groups entries, n0 ascending */
  const float* row_scale;           /* indexed by the OUTPUT row orow(m), or NULL */
  int32_t M, Ntot, K, lda, ldb, n_groups;
  int32_t rows_in, rows_out;
  int32_t c_f32, accumulate, dtype, ldc;
  float alpha;
  int32_t group_align;              /* the caller's promise about the group table: every n0 and N is a multiple of it (0 = 128, the minimum);
                                       256 and up lets large products run on the 256 x 256 tile form */
  void* c;                          /* groups == NULL: ONE group over all Ntot columns, given inline (c, bias, ldc) */
  const float* bias;
} jen1_bgemm_args;
int jen1_big_gemm(const jen1_bgemm_args* args, void* stream);
/* the weight gradient of the same projection (autograd of blocks.py:428 ``to_kv``): C[n][k] += alpha * sum_m A[m][n] B[m][k] with
 * A = dY [M][lda >= N], B = the layer's input [M][ldb >= K] (bf16, as they lie in memory: the reduction index is the row), C float32
 * [N][ldc] accumulated with float atomics (the reduction is split over workgroups): ``param.grad`` of the reference layout. */
int jen1_big_gemm_tn(const void* a, const void* b, float* c, int M, int N, int K, int lda, int ldb, int ldc, float alpha, void* stream);
/* the same product WRITTEN to C (C = alpha * A^T B): the reduction is not split, every workgroup stores its own tile -- no atomics and no
 * zero-fill in front (the stacked weight gradient of the text-context projections, 17408 x 1024 floats: csrc/train_kvbank.hip) */
int jen1_big_gemm_tn_store(const void* a, const void* b, float* c, int M, int N, int K, int lda, int ldb, int ldc, float alpha, void* stream);
/* the convolution form of jen1_big_gemm (bf16): y[b T_out + t][n] = sum_tap sum_c x[b T_in + t * stride + tap - pad][c] * W_tap[n][c] (+ bias[n])
 * (+ residual[row][n]); rows outside [0, T_in) count as zeros.  W_tap = w + tap * w_tap_stride (tap_rev: taps - 1 - tap), [co][ld_w] with the
 * ci input channels contiguous.  The forward and (stride 1, pad' = taps - 1 - pad, tap_rev) data-gradient passes of `_Conv1d`
 * (blocks.py:34-53) over many rows: the long levels of the training pass.  ci must be a multiple of 8 (the last 64-channel K step of a tap
 * may be ragged: its missing chunks are not read). */
int jen1_big_gemm_conv(const void* x, const void* w, const float* bias, const void* residual, void* y, int B, int T_in, int T_out, int ci, int co,
                       int taps, int stride, int pad, int tap_rev, int ld_x, int ld_w, int w_tap_stride, int ld_y, const int32_t* shift_b /* [B] added to
                       the row shift tap - pad per batch element (causal and centred clips in one pass, blocks.py:45-50), or NULL */,
                       int div /* > 1: the mapped position t * stride + tap - pad must be a multiple of div and is divided by it, other taps read
                       zeros: the forward pass of nn.ConvTranspose1d (blocks.py:80-88) with stride = 1, tap_rev, pad' = taps - 1 - padding */,
                       void* stream);
/* the weight (and bias) gradient of a Conv1d / Linear over many rows (autograd of blocks.py:34-53 `_Conv1d`, the long levels of the
 * pass: B * T_out = 6 000 .. 24 000 reduction rows against 128 .. 512 channels): gw[co][ci][tap] += alpha * sum_{b,t} dy[b T_out + t][co] *
 * x[b T_in + t * stride + tap - pad][ci] (rows outside [0, T_in) count as zeros), gb[co] += alpha * sum_{b,t} dy[..][co] when gb is not
 * NULL.  dy [B T_out][ld_dy], x [B T_in][ld_x] bf16 as they lie in memory, gw float32 in the reference layout (`param.grad`),
 * accumulated with float atomics (runs of consecutive floats: a column tile of the kernel holds an 8-channel chunk of every tap).  taps <= 16. */
int jen1_big_gemm_tn_conv(const void* dy, const void* x, float* gw, float* gb, int B, int T_out, int T_in, int co, int ci, int taps, int stride,
                          int pad, int ld_dy, int ld_x, float alpha, const int32_t* shift_b /* as in jen1_big_gemm_conv */, void* stream);

/* y[r][0..C) = (x[r] - mean_r) / sqrt(var_r + eps): LayerNorm's standardisation (blocks.py:400-401 ``norm_context`` without its
 * affine, which the packed weights carry) of float32 rows, written in the compute dtype; statistics over the ROUNDED values */
int jen1_standardize_rows(const float* x, void* y, int rows, int C, int ldx, int ldy, float eps, int dtype, void* stream);

/* the unconditional K/V slots of all cross-attention layers in one launch: out_l[b][n][:] = fixed_l[n][:] * mask[b][n]
 * (model.py:337: the learned fixed embedding is masked with the TEXT mask).  table: device array of n_layers entries
 * {const void* fixed; void* out; int32 C2; int32 0; int64 0}. */
int jen1_kv_fixed_fill(const void* table_dev, int n_layers, const float* mask, int B, int rows, int dtype, void* stream);

const char* jen1_last_error(void);
/* "gfx950" build tag + ABI version, for the loader's sanity check */
const char* jen1_build_info(void);
int jen1_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* JEN1_HIP_H */
