/*
 * jen1_train.h -- C ABI of the training (forward-with-saves + backward) kernels in libjen1_hip.so.
 *
 * The reference trains through torch.autograd (trainer.py:141 `scaler.scale(loss / n).backward()`), i.e. it has
 * no interface of its own for the backward pass: the seam is the autograd formula of every operator on the path
 * ("for training ops -- registered with an autograd formula").  Each entry point below is
 * the forward or the backward of one such operator; jen1_amd/train.py binds them as torch.autograd.Function's.
 *
 * Layout: activations are channel-last rows, x[row][c] with a row pitch `ld` (elements) that is a multiple of 8
 * and zero padding columns; row = b * L + t.  Statistics, gradients of parameters and every reduction are
 * float32.  dtype: JEN1_F32 (parity mode) or JEN1_BF16 (bf16 storage, float32 accumulation).
 * All functions return 0 on success (jen1_last_error() otherwise) and only enqueue work on `stream`.
 */
#ifndef JEN1_TRAIN_H
#define JEN1_TRAIN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/*
 * One operand of jen1_train_gemm.  Element (row r, tap, inner index k) of batch z lives at
 *   p + (z / zdiv) * zs0 + (z % zdiv) * zs1 + tap * tap_stride + R * ld_r + K * ld_k      (element units)
 * where (R, K) = (r, k) except along `map_axis` (1: rows, 2: inner index), whose index i = b * map_L + t is sent to
 *   s = t * map_mul + tap * map_tapmul + map_shift;  if map_div > 1: s must be divisible, s /= map_div;
 *   valid iff 0 <= s < map_Lsrc;   mapped index = b * map_Lsrc + s;   invalid elements read as 0.
 * This one rule is the zero padding of _Conv1d (blocks.py:45-50: mul = stride, tapmul = 1, shift = -pad_left),
 * its data gradient and ConvTranspose1d (blocks.py:80-88: mul = 1, tapmul = -1, shift = +pad, div = stride).
 * 16-byte loads are used when the contiguous axis (ld_k == 1 or ld_r == 1) is the unmapped one and aligned.
 */
typedef struct jen1_gemm_operand {
  const void* p;
  int64_t zs0, zs1;
  int64_t ld_r, ld_k, tap_stride;
  int32_t zdiv;
  int32_t map_axis;
  int32_t map_L, map_Lsrc, map_mul, map_tapmul, map_shift, map_div;
  int32_t map_reflect;     /* 1: out-of-range s is mirrored (s < 0 -> -s, s >= Lsrc -> 2 (Lsrc - 1) - s) instead of reading 0:
                              F.pad(mode="reflect") of the SEANet convolutions (encodec 0.1.1 modules/conv.py pad1d) */
  int32_t reserved;        /* 0, or 1: the shift of batch element b of the mapped axis is jen1_gemm_args.map_shift_b[b] instead of map_shift */
} jen1_gemm_operand;

/*
 * C[z][tap?][m][n] (+)= alpha * sum_{tap?} sum_k A(m, tap, k) * B(n, tap, k)  (+ bias[n])
 *   taps_in_z = 0: taps are summed (forward conv, data gradient);  1: one output matrix per tap (weight gradient)
 *   C element address: c + (z / c_zdiv) * c_zs0 + (z % c_zdiv) * c_zs1 + tap * c_tap_stride + m * ldc_m + n * ldc_n
 *   c_f32: C is float32 whatever dtype is;  atomic: C += through float atomics (needs c_f32; required by splitk > 1)
 *   accumulate: C += without atomics (exclusive tiles);  splitk: the K range is cut into `splitk` slices.
 * Replaces: F.conv1d / F.conv_transpose1d / F.linear / einsum forward and their autograd formulas on the path
 * (blocks.py:34-53, 69-95, 337-380, 440-446; model.py:75-89).
 */
typedef struct jen1_gemm_args {
  jen1_gemm_operand a, b;
  void* c;
  const void* bias;        /* float32 [N] or NULL */
  int64_t c_zs0, c_zs1, ldc_m, ldc_n, c_tap_stride;
  int32_t c_zdiv;
  int32_t M, N, K, taps, batches;
  int32_t taps_in_z, splitk, atomic, accumulate, c_f32, dtype;
  float alpha;
  int32_t reserved;        /* 0, or 1 = prefer the SKINNY form when both operands are K-contiguous and aligned: 32 x 16 output tiles whose
                              four waves split K (few rows against a big weight: N / 16 x M / 32 workgroups instead of a split-K launch
                              with float atomics + a conversion launch); ignored when the operands do not allow it */
  float* rowsum;           /* taps_in_z only: rowsum[m] += alpha * sum_k A(m, tap 0, k)  (the bias gradient, blocks.py:52
                              nn.Conv1d bias, riding on the weight gradient); float32 [M] or NULL */
  const void* residual;    /* NULL, or a tensor of C's dtype and indexing that is added in the epilogue (the residual of a
                              ResnetBlock1d / transformer sub-block, blocks.py:231, :486-488); not with the atomic epilogue */
  const int32_t* map_shift_b;  /* NULL, or one map_shift per batch element of the mapped axis (for the operands with reserved == 1): a
                              pass that holds causal and non-causal clips side by side -- _Conv1d pads k - 1 on the left for the
                              former and (k - 1) / 2 on both sides for the latter (blocks.py:45-50), which is the only place the
                              flag enters a convolution.  One entry more than there are batch elements (the K walk of a weight
                              gradient steps one element past the end before its loads are masked). */
} jen1_gemm_args;

int jen1_train_gemm(const jen1_gemm_args* args, void* stream);
/* Two INDEPENDENT products of one dtype in one launch: one grid holds the workgroups jen1_train_gemm(first) and
 * jen1_train_gemm(second) would launch (the second's are dispatched first); results are those of the two calls.  For the weight gradient (first) and the data gradient
 * (second) of one layer -- loss.backward() of trainer.py:141 issues them back to back, both read dY, neither reads the other -- the
 * launch lasts as long as the longer of the two instead of their sum.  first must be of the 64 x 64 form (reserved == 0 or operands
 * that do not allow the skinny form); second may be either.  Nothing may order the two (no common output). */
int jen1_train_gemm_pair(const jen1_gemm_args* first, const jen1_gemm_args* second, void* stream);

/* --- GroupNorm (+FiLM) (+SiLU): ConvBlock1d's prologue, blocks.py:137-143; Transformer1d's GroupNorm, :509 ---
 * sums[B][G][2] float32 = (sum x, sum x^2) over the group (zeroed by the call).  film: [B][film_ld] float32 or bf16
 * (same dtype as x) holding scale at [c] and shift at [C + c], or NULL.  flags bit0: SiLU. */
int jen1_gn_sums(const void* x, float* sums, int B, int L, int C, int ld, int groups, int dtype, void* stream);
int jen1_gn_apply(const void* x, const float* sums, const float* gamma, const float* beta, const void* film, int film_ld,
                  void* y, int B, int L, int C, int ld, int groups, float eps, int flags, int dtype, void* stream);
/* jen1_gn_sums + jen1_gn_apply as ONE launch where the shape allows (rows without padding, groups of 8 x 2^k channels, at least 32
 * (batch element, group) pairs of at most 2^17 elements: a workgroup per pair computes the statistics and normalises, no scratch
 * reset, no atomics); other shapes run the two calls.  sums[B][G][2] is written as by jen1_gn_sums (the backward pass reads it). */
int jen1_gn_forward(const void* x, float* sums, const float* gamma, const float* beta, const void* film, int film_ld, void* y, int B,
                    int L, int C, int ld, int groups, float eps, int flags, int dtype, void* stream);
/* backward: P[B][C][4] and Gm[B][G][2] are float32 scratch (P is zeroed by the call); dgamma/dbeta are ACCUMULATED
 * (float32, the parameter's .grad); dfilm [B][2C] float32 is written (NULL when film is NULL).  flags bit1: dfilm MIRRORS film
 * instead -- x's dtype, rows film_ld apart (scale gradient at [c], shift gradient at [C + c]): the slice of a buffer that holds the
 * FiLM gradients of every block side by side, the operand of ONE data-gradient GEMM for all their projections. */
int jen1_gn_backward(const void* dy, const void* x, const float* sums, const float* gamma, const float* beta, const void* film,
                     int film_ld, void* dx, float* dgamma, float* dbeta, void* dfilm, float* P, float* Gm, int B, int L, int C,
                     int ld, int groups, float eps, int flags, int dtype, void* stream);

/* the same with dx_add (x's dtype and layout, may be NULL): dx = the GroupNorm gradient + dx_add.  A tensor that feeds a norm AND a
 * residual branch (ResnetBlock1d's input, blocks.py:219-231) gets two gradients; adding the second one here replaces autograd's
 * accumulation launch. */
int jen1_gn_backward_add(const void* dy, const void* x, const float* sums, const float* gamma, const float* beta, const void* film,
                         int film_ld, void* dx, const void* dx_add, float* dgamma, float* dbeta, void* dfilm, float* P, float* Gm, int B,
                         int L, int C, int ld, int groups, float eps, int flags, int dtype, void* stream);
/* ... and with a second one, dx_add2 (may be NULL): a ResnetBlock1d input that is also a skip connection of the U-Net (blocks.py:641-643
 * collects every block output for the up path, :732-734 consumes it) receives three gradients. */
int jen1_gn_backward_add2(const void* dy, const void* x, const float* sums, const float* gamma, const float* beta, const void* film,
                          int film_ld, void* dx, const void* dx_add, const void* dx_add2, float* dgamma, float* dbeta, void* dfilm, float* P,
                          float* Gm, int B, int L, int C, int ld, int groups, float eps, int flags, int dtype, void* stream);

/* --- LayerNorm over the last axis (blocks.py:400-401): stats[rows][2] = (mean, rstd) --- */
int jen1_ln_forward(const void* x, const float* gamma, const float* beta, void* y, float* stats, int rows, int C, int ld,
                    float eps, int dtype, void* stream);
int jen1_ln_backward(const void* dy, const void* x, const float* stats, const float* gamma, void* dx, float* dgamma,
                     float* dbeta, int rows, int C, int ld, int dtype, void* stream);
/* the same with dx_add (x's dtype, rows ld apart, may be NULL): dx = the LayerNorm gradient + dx_add (the residual branch of a
 * transformer sub-block, blocks.py:486-488).  dx may be NULL when the input needs no gradient (norm_context over the text
 * embedding, blocks.py:426): only dgamma / dbeta are accumulated. */
int jen1_ln_backward_add(const void* dy, const void* x, const float* stats, const float* gamma, void* dx, const void* dx_add,
                         float* dgamma, float* dbeta, int rows, int C, int ld, int dtype, void* stream);
/* Two LayerNorms of ONE input in one launch (self-attention: ``norm`` for to_q and ``norm_context`` for to_kv over the same x,
 * blocks.py:427-429 with context = x): y1 = LN(x; gamma1, beta1), y2 = LN(x; gamma2, beta2), one (mean, rstd) pair per row.  Backward:
 * dx = LN'(dy1 gamma1 + dy2 gamma2) (+ dx_add), the four parameter gradients accumulate.  Rows of a multiple of 8 channels, C <= 1024,
 * 16-byte aligned tensors (callers fall back to two jen1_ln_forward / jen1_ln_backward_add launches otherwise). */
int jen1_ln2_forward(const void* x, const float* gamma1, const float* beta1, const float* gamma2, const float* beta2, void* y1, void* y2,
                     float* stats, int rows, int C, int ld, float eps, int dtype, void* stream);
int jen1_ln2_backward_add(const void* dy1, const void* dy2, const void* x, const float* stats, const float* gamma1, const float* gamma2,
                          void* dx, const void* dx_add, float* dgamma1, float* dbeta1, float* dgamma2, float* dbeta2, int rows, int C,
                          int ld, int dtype, void* stream);

/* --- pointwise activations: mode 0 GELU(erf) (blocks.py:443, model.py:77-89), 1 SiLU (blocks.py:158),
 *     2 ELU(alpha = 1) (the SEANet decoder behind generation.py:130) --- */
int jen1_act_forward(const void* x, void* y, int64_t n, int mode, int dtype, void* stream);
int jen1_act_backward(const void* dy, const void* x, void* dx, int64_t n, int mode, int dtype, void* stream);

/* --- softmax over keys (blocks.py:367-371): s float32 [Z][Nq][ld_s] (already scaled) -> p (dtype) [Z][Nq][ld_p];
 * causal keeps j <= i + (Nk - Nq) (blocks.py:315-319); padding columns Nk..ld_p-1 of p are written as 0.
 * backward: ds = p * (dp - sum_j dp*p), dp float32 [Z][Nq][ld_s], ds (dtype) [Z][Nq][ld_p]. */
int jen1_softmax_forward(const float* s, void* p, int rows, int Nq, int Nk, int ld_s, int ld_p, int causal, int dtype, void* stream);
int jen1_softmax_backward(const void* p, const float* dp, void* ds, int rows, int Nk, int ld_s, int ld_p, int dtype, void* stream);

/* --- the whole attention core (AttentionBase.forward, math path, blocks.py:355-380) of SHORT sequences in one launch each way: one
 * workgroup per (batch element, head) keeps Q, K, V (and dO) in LDS -- the transformer blocks of JEN-1 sit where a 1500-frame clip
 * is 1 .. 24 positions and the text context 130 tokens.  q [B][Nq][ldq], k / v [B][Nk][ldk / ldv], o / d_o [B][Nq][ldo] hold head h
 * in columns [h d, (h + 1) d); p [B H][Nq][ldp] (dtype) is the softmax output rounded to the dtype (P V uses the rounded values;
 * columns Nk .. ldp - 1 are written as 0), saved for the backward pass; causal keeps j <= i + (Nk - Nq) (blocks.py:315-319);
 * causal_b (int32 [B], may be NULL) gives the flag per batch element instead (a pass that mixes causal and non-causal clips).
 * kv_mask (float32 [B][Nk], may be NULL) multiplies the rows of K and V on the way in -- the padding mask of the text context, which
 * the reference multiplies into k and v (blocks.py:431-434) -- and the rows of dk / dv on the way out.
 * backward writes dq [B][Nq][lddq], dk [B][Nk][lddk], dv [B][Nk][lddv] (head h in the same columns).
 * jen1_attn_small_fits: whether (Nq, Nk, d) fit one workgroup's LDS (callers fall back to jen1_train_gemm + softmax otherwise). */
int jen1_attn_small_fits(int Nq, int Nk, int d, int dtype);
int jen1_attn_small_forward(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo,
                            void* p, int64_t ldp, int B, int H, int Nq, int Nk, int d, float scale, int causal, const int32_t* causal_b,
                            const float* kv_mask, int dtype, void* stream);
int jen1_attn_small_backward(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* p,
                             int64_t ldp, const void* d_o, int64_t ldo, void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv,
                             int64_t lddv, int B, int H, int Nq, int Nk, int d, float scale, const float* kv_mask, int dtype,
                             void* stream);
/* the same with a row map for K / V: batch element b READS k / v of batch row kv_row[b] (NULL: b).  The unconditional half of the CFG
 * pair attends to the learned fixed embedding whatever the batch element (model.py:333): its context rows are projected ONCE and
 * shared; dk / dv are still written per batch element (jen1_sum_rows_inplace adds the sharers' rows up afterwards). */
int jen1_attn_small_forward_rows(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, void* o, int64_t ldo,
                                 void* p, int64_t ldp, int B, int H, int Nq, int Nk, int d, float scale, int causal, const int32_t* causal_b,
                                 const float* kv_mask, const int32_t* kv_row, int dtype, void* stream);
int jen1_attn_small_backward_rows(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv, const void* p,
                                  int64_t ldp, const void* d_o, int64_t ldo, void* dq, int64_t lddq, void* dk, int64_t lddk, void* dv,
                                  int64_t lddv, int B, int H, int Nq, int Nk, int d, float scale, const float* kv_mask, const int32_t* kv_row,
                                  int dtype, void* stream);
/* p[0 .. n) = sum over r < rows of p[r * n .. r * n + n)  (in place, float32 accumulation): the gradients of context rows that several
 * batch elements shared */
int jen1_sum_rows_inplace(void* p, int rows, int64_t n, int dtype, void* stream);

/* dst[i] = (dtype) src[i]; src[i] = 0  for i < n (n a multiple of 4): hands the float32 accumulator of a split-K GEMM
 * over in the compute dtype and leaves it zeroed for the next launch of the stream. */
int jen1_convert_clear(float* src, void* dst, int64_t n, int dtype, void* stream);
/* the same with a residual: dst[i] = (dtype)(src[i] + res[i]) (res in `dtype`) */
int jen1_convert_clear_add(float* src, void* dst, const void* res, int64_t n, int dtype, void* stream);

/* --- Encodec pieces either side of the sampler (generation.py:113, :130, :145-150) ---
 * jen1_rvq_decode: ResidualVectorQuantizer.decode -- out[b][d][t] = sum_q tables[q][codes[q][b][t]][d]
 *   codes int64 [n_q][B][T], tables float32 [n_q][bins][D], out float32 [B][D][T] (the reference's latent layout).
 * jen1_lstm_layer: one nn.LSTM layer (gate order i, f, g, o) over T steps for B sequences, as SLSTM uses it:
 *   gin float32 [B][T][4H] = x W_ih^T + b_ih + b_hh (a jen1_train_gemm), whh_t [H][4H] = W_hh^T in `dtype`,
 *   y (dtype) [B][T][ld_y] = h_t (+ skip[b][t][j] when skip != NULL: SLSTM's residual), zero initial state. */
int jen1_rvq_decode(const int64_t* codes, const float* tables, float* out, int n_q, int B, int T, int bins, int D, void* stream);
int jen1_lstm_layer(const float* gin, const void* whh_t, const void* skip, void* y, int B, int T, int H, int ld_y, int dtype,
                    void* stream);
/* The same layer spread over H / 32 workgroups per group of 8 sequences, the slice of W_hh of every workgroup resident in
 * registers and one grid barrier per step (needs all workgroups co-resident: nothing else may occupy the GPU's CUs).
 * (bf16, H = 512, B > 1: the recurrent product runs on the matrix cores, 16 sequences per group, h split into bf16 high + low.)
 * whh [4H][H] (NOT transposed) in `dtype`; hbuf: float32 scratch [ceil(B / 8)][2][16][H]; counters: uint32 [ceil(B / 8)][32], ZERO on
 * entry; counters[g * 32 + 1] != 0 afterwards reports a barrier time-out (output invalid). */
int jen1_lstm_layer_multi(const float* gin, const void* whh, const void* skip, void* y, float* hbuf, uint32_t* counters, int B, int T,
                          int H, int ld_y, int dtype, void* stream);

/* --- the two ends of a training pass (csrc/train_glue.hip): what the reference writes as ATen elementwise / cat / reduce ops around
 * the network, one launch each ---
 * jen1_train_pack_input: x_t = ca[b] x0 + cb[b] noise (q_sample, gdm.py:232-243), torch.cat([x_t, input_concat_cond], 1)
 *   (model.py:240), the CFG pair's torch.cat([x, x]) (model.py:332; nrep = 2) and the change to channel-last rows
 *   y[(r B + b) T + t][0..ld) in `dtype` (zero padding columns); with tgt != NULL also the loss target of the objective
 *   (gdm.py:260-266) tgt[(b T + t) C + c] = ta[b] noise + tb[b] x0 as float32 rows.  x0 / noise [B][C][T], ctx [B][Cc][T] float32.
 * jen1_train_context: the cross-attention context rows out[r][n][0..F) of the pass in `dtype`: r < B: the text embedding
 *   emb[r][n] (n < NL) followed by the time token tok[r] (model.py:315-316), or the learned fixed embedding for rows with
 *   drop[r] != 0 (CFG dropout, model.py:323-328); B <= r < nrep B: the fixed embedding (the pair's unconditional half, :333).
 *   nrep = 0: B + 1 rows -- the unconditional half as ONE shared row set (it is the same for every batch element; the attention
 *   kernels read it through a row map, jen1_attn_small_forward_rows).
 *   _backward: d_fixed[n] += sum of d over the rows that read the fixed embedding; d_tok[r] = d[r][N - 1] (0 for dropped rows).
 * jen1_time_features_fwd / _bwd: f[b] = [t, sin(2 pi t w), cos(2 pi t w), 0 ..] (LearnedPositionalEmbedding, utils/module.py:58-72;
 *   phases in float32, evaluated left to right like the reference); dw[k] += its gradient.  t: int64 or float32 [B].
 * jen1_cfg_loss_forward / _backward: on the network's output rows net[(r B + b) T + t][0..C) (`dtype`): the CFG combine
 *   out_masked + (out - out_masked) s and the unbiased-std rescale (model.py:362-369), the elementwise l2 / l1 against tgt and the
 *   mean over (C, T) (gdm.py:268-272) -> loss_ps[b] (float32, zeroed by the call); backward: dnet = d loss / d net for
 *   upstream gradients gps[b] of the per-sample losses, both halves of the pair, padding columns zeroed. */
int jen1_train_pack_input(const float* x0, const float* noise, const float* ca, const float* cb, const float* ctx, void* y, int B, int C, int Cc,
                          int T, int ld, int nrep, const float* ta, const float* tb, float* tgt, int dtype, void* stream);
int jen1_train_context(const float* emb, const float* tok, const float* fixed, const uint8_t* drop, void* out, int B, int NL, int N, int F,
                       int nrep, int dtype, void* stream);
int jen1_train_context_backward(const void* d, const uint8_t* drop, float* d_fixed, float* d_tok, int B, int NL, int N, int F, int nrep, int dtype,
                                void* stream);
int jen1_time_features_fwd(const void* t, int t_is_float, const float* w, float* f, int B, int half, int ld, void* stream);
int jen1_time_features_bwd(const void* t, int t_is_float, const float* w, const float* df, float* dw, int B, int half, int ld, void* stream);
int jen1_cfg_loss_forward(const void* net, const float* tgt, float* loss_ps, int B, int C, int T, int ld, int nrep, float embedding_scale,
                          int scale_cfg, float scale_phi, int l1, int dtype, void* stream);
int jen1_cfg_loss_backward(const void* net, const float* tgt, const float* gps, void* dnet, int B, int C, int T, int ld, int nrep,
                           float embedding_scale, int scale_cfg, float scale_phi, int l1, int dtype, void* stream);

/* --- the skip concat of the up path (blocks.py:732-734: torch.cat([x, skip * 2^-1/2], dim=channels)) on channel-last rows ---
 * jen1_concat2: out[row] = [a[row][0..Ca) | scale_b * b[row][0..Cb)];  jen1_split2 (its backward): da[row] = d[row][0..Ca),
 * db[row] = scale_b * d[row][Ca..Ca+Cb).  Rows are dense (pitch = channel count), channel counts multiples of 8. */
int jen1_concat2(const void* a, const void* b, void* out, int64_t rows, int Ca, int Cb, float scale_b, int dtype, void* stream);
int jen1_split2(const void* d, void* da, void* db, int64_t rows, int Ca, int Cb, float scale_b, int dtype, void* stream);

/* --- compute copies of the parameters, all in one launch (refreshed after every optimiser step) ---
 * entry: dst[i0][i1][i2] (dims d0 x d1 x d2, row pitch ld >= d2, element type `dtype`; padding columns are not touched)
 *        = (dtype) src[i0 s0 + i1 s1 + i2 s2]   (float32 parameter in the reference layout: _Conv1d [Co][Ci][k] blocks.py:41-52,
 *        ConvTranspose1d [Ci][Co][k] :80-88, Linear [Co][Ci]); tile0 = first 32 x 32 tile of the entry in the launch
 *        (prefix sum of d0 ceil(d1 / 32) ceil(d2 / 32)), entries sorted by tile0.  entries_dev: device copy of n entries. */
typedef struct jen1_repack_entry {
  const float* src;
  void* dst;
  int32_t d0, d1, d2, ld;
  int64_t s0, s1, s2;
  int32_t tile0, ld2;
  void* dst2;              /* NULL, or a second copy written from the same read: dst2[i0][i2][i1] with rows ld2 apart (the
                              data-gradient transpose [k][C_in][C_out] of the same weight) */
} jen1_repack_entry;
int jen1_repack(const jen1_repack_entry* entries_dev, int n, int total_tiles, int dtype, void* stream);

/* --- the text-context K / V projections of all cross-attention layers as one product (csrc/train_kvbank.hip; reference
 * jen1/model/blocks.py:400-407, :427-434: k, v = chunk(to_kv(norm_context(context))) on the same context in every layer).
 * to_kv_l(norm_context_l(x)) = xhat Wf_l^T + bias_l with Wf_l = W_l diag(gamma_l), bias_l = W_l beta_l and xhat the standardised rows
 * (shared).  One table entry per layer, n0 ascending; every n0 and N a multiple of 32. */
typedef struct jen1_kv_layer {
  const float* w;          /* to_kv.weight [N][K], the float32 parameter */
  const float* gamma;      /* norm_context.weight [K] */
  const float* beta;       /* norm_context.bias [K] */
  float* gw;               /* their gradients (``param.grad``), accumulated by jen1_kv_fold_backward */
  float* ggamma;
  float* gbeta;
  int32_t n0, N;           /* the layer's rows in the stacked operands */
} jen1_kv_layer;
/* wf [Ntot][K] bf16 = W diag(gamma), wft [K][ld_wft >= Ntot] bf16 = its transpose, bias [Ntot] float32 = W beta */
int jen1_kv_fold(const jen1_kv_layer* table_dev, int n_layers, int Ntot, int K, void* wf, void* wft, int ld_wft, float* bias, void* stream);
/* gw += dwf diag(gamma) + dbias beta^T, ggamma += colsum(dwf * W), gbeta += W^T dbias; dwf [Ntot][K], dbias [Ntot] float32 */
int jen1_kv_fold_backward(const jen1_kv_layer* table_dev, int n_layers, int Ntot, int K, const float* dwf, const float* dbias, void* stream);
/* block 0 += blocks 1 .. nblk - 1 in place; a block = rows_per_block rows of row_elems elements (a multiple of 8), rows ld apart, blocks
 * rows_per_block * ld apart: dK | dV of the batch elements that share one set of context rows, inside the stacked gradient matrix */
int jen1_sum_rows_strided(void* p, int nblk, int rows_per_block, int row_elems, int64_t ld, int dtype, void* stream);

/* out[c] += sum_rows x[row][c]  (bias gradients), float32 accumulate */
int jen1_colsum(const void* x, float* out, int rows, int C, int ld, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* JEN1_TRAIN_H */
