/*
 * dldkd_hip.h - C ABI of libdldkd_hip.so: the MI355X (gfx950) kernels under the DL-DKD scoring +
 * distillation hot path.
 *
 * The upstream reference (HuiGuanLab/DL-DKD) has no FFI: its boundary for this path is the Python
 * class method.model.DLDKD and the module functions of method/eval.py.  This
 * header is the boundary one level down: what a maintainer of the reference would bind (ctypes, see
 * INTEGRATION.md) to run those functions on an MI355X.  Every entry point names the reference code it
 * replaces.
 *
 * Conventions
 *   - plain C: pointers are DEVICE pointers unless the name says host_; sizes are ints; `stream` is a
 *     hipStream_t passed as void* (NULL = the default stream).  No torch / C++ types.
 *   - every function only ENQUEUES work on `stream` (no allocation, no free, no host sync, no process-global device
 *     state: safe to capture in a hipGraph and to call from several streams) and returns DLDKD_OK or a negative
 *     DLDKD_E* code; dldkd_last_error() gives the text.  Scratch memory is always the caller's (`workspace` arguments,
 *     sized by the matching *_bytes() function).
 *   - float tensors are row-major fp32 unless stated; "bf16" buffers are opaque device blobs whose size
 *     comes from the matching *_bytes() function.
 */
#ifndef DLDKD_HIP_H
#define DLDKD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DLDKD_OK 0
#define DLDKD_EINVAL (-1)   /* bad shape / null pointer / unsupported size */
#define DLDKD_ELAUNCH (-2)  /* HIP launch error */
#define DLDKD_ECOMM (-3)    /* RCCL error (dldkd_comm_*), or librccl.so.1 not loadable */

#define DLDKD_HIDDEN 384    /* config.py:70-71 hidden size the kernels are specialised for */
#define DLDKD_MAX_CLIPS 128 /* config.py:60 max_ctx_l */

int dldkd_abi_version(void);
const char* dldkd_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Scoring: query x video x clip similarity with key-clip max-pool.
 * Replaces DLDKD.get_sim_scores (method/model.py:307-329) / get_unnormalized_sim_scores (:331-350) as
 * they are used by compute_query2ctx_info (method/eval.py:200-208) and eval_epoch's 0.7/0.3 fusion
 * (method/eval.py:254).
 * ------------------------------------------------------------------------------------------- */

/* Bytes of the packed bf16 query blob for nq queries of dimension DLDKD_HIDDEN. */
size_t dldkd_packed_queries_bytes(int nq);
/* Bytes of the packed bf16 gallery blob for nv videos padded to L clips (L <= DLDKD_MAX_CLIPS). */
size_t dldkd_packed_gallery_bytes(int nv, int L);
/* Bytes of scratch the eval scorer needs (per-branch transposed partial scores). */
size_t dldkd_simpool_eval_workspace_bytes(int nq, int nv, int n_branches);

/* q (nq, 384) fp32 -> packed bf16 MFMA-fragment order; normalize != 0 applies F.normalize (eps 1e-12,
 * model.py:318) in fp32 before rounding.  bad_flags (nq rounded up to a multiple of 32 floats, zeroed by the caller, or NULL): set to 1 for every query
 * vector with a NaN / Inf component.  The scorer's max-pool drops NaN products, so such a query would tie every video
 * (and rank its ground truth FIRST); hand the flags to dldkd_simpool_finish_range / dldkd_simpool_rank_partials and the
 * query's scores become NaN, which ranks last (dldkd_rank_gt's NaN policy): a diverged model scores R@K = 0, not 100. */
int dldkd_pack_queries_bf16(const float* q, int nq, int normalize, void* q_packed, float* bad_flags, void* stream);

/* lens[i] = number of entries > 0 of row i of a (n, L) 0/1 mask (model.py:192 counts a video's clips this way; masks are prefix
 * masks, data_provider.py:81-84): one kernel in place of (mask > 0).sum(1).to(int32). */
int dldkd_mask_lens_f32(const float* mask, int n, int L, int32_t* lens, void* stream);
/* g (nv, L, 384) fp32 + mask (nv, L) fp32 0/1 prefix masks (NULL = all valid) -> bf16 gallery blob,
 * lens[nv] int32 (number of valid clips, data_provider.py:81-84).  normalize as above (model.py:319).
 * Blob layout: row-major bf16 [nv][round_up(L,32)][384]; rows past a video's length inside its last 16-row tile replicate
 * its last valid clip (so the scorer needs no padding mask), rows beyond that tile are zero. */
int dldkd_pack_gallery_bf16(const float* g, const float* mask, int nv, int L, int normalize,
                            void* g_packed, int32_t* lens, void* stream);

/* Streaming form of dldkd_pack_gallery_bf16 for the eval driver (compute_context_info, eval.py:114-175, encodes the
 * gallery in batches of eval_context_bsz and zero-pads to the global max length, eval.py:139-155): packs one encoded
 * batch (nv_chunk, L_chunk, 384) + its mask into videos [v0, v0 + nv_chunk) of a blob sized for (nv_total, L_total);
 * clips l >= L_chunk are zero rows.  lens points at the lens array of the WHOLE gallery. */
int dldkd_pack_gallery_chunk_bf16(const float* g, const float* mask, int nv_chunk, int L_chunk, int normalize,
                                  void* g_packed, int32_t* lens, int v0, int nv_total, int L_total, void* stream);

/* The scorer's visiting order: order[pos] = video by DESCENDING lens (equal lengths in index order), inv_order[video] = pos.
 * Any permutation is valid for dldkd_simpool_eval_bf16; this one keeps the four videos of a workgroup alike and the grid's tail
 * light.  lens in [0, DLDKD_MAX_CLIPS] (clamped). */
int dldkd_order_by_len_desc(const int32_t* lens, int nv, int32_t* order, int32_t* inv_order, void* stream);

/* All-pairs pooled scores, stage 1 (the dominant kernel).  For each branch b < n_branches (1 or 2):
 *     part_b[pos(v), q] = max_{l < lens[v]} < q_packed[b][q], g_packed[b][v, l] >   (model.py:321-327)
 * written to `workspace` as [n_branches][nv][round_up(nq,32)] fp32, videos in `order` (order[pos] = v;
 * any permutation is valid; descending-length order balances the workgroups).  bf16 operands, fp32
 * accumulation; the (nq, L, nv) clip tensor of the reference is never formed.
 *
 * q_split: the queries are cut into that many contiguous ranges of whole 32-query tiles and the launch grid becomes
 * [range][branch][4 videos]: a small gallery (one rank's shard when the gallery is sharded over 8 GPUs, the loop of
 * eval.py:188-212) still fills the 256 CUs.  0 = choose (dldkd_simpool_eval_plan with min_split 1 tells what is chosen).
 * Results do not depend on q_split, bit for bit.
 * done: NULL, or int32 counters [number of ranges], zeroed by the caller on `stream` before this call.  Ranges are
 * dispatched in order; every workgroup of range s adds 1 to done[s] after releasing its scores at agent scope, so
 * done[s] == ceil(nv/4) * n_branches means "the scores of the queries of range s are in memory" while later ranges are
 * still being computed (dldkd_stream_wait_counter parks a second stream on exactly that). */
int dldkd_simpool_eval_bf16(const void* const* q_packed, const void* const* g_packed, const int32_t* lens,
                            const int32_t* order, int nq, int nv, int L, int n_branches, int q_split, int32_t* done,
                            void* workspace, void* stream);

/* The same scores from PAIR WAVES: wave w of the launch scores the videos at sorted positions pairs[2 w] and pairs[2 w + 1]
 * (-1 = none) in its 128 register-resident rows - video A in rows [0, round_up(len A, 4)), video B behind it; the caller pairs
 * them so that 112 < round_up(len A, 4) + len B <= 128 (scoring.pair_waves; a pair that does not fit is scored as A alone).  A
 * ragged gallery then needs fewer, full waves (TVR lengths U{24..128}: ~13.7 k waves instead of 21.8 k, 5 % fewer 16-row MFMA
 * tiles).  Every position of `order` appears in exactly one wave.  The workspace layout and contents are those of
 * dldkd_simpool_eval_bf16, bit for bit (same finish / rank entry points); done[s] counts ceil(n_waves/4) * n_branches
 * workgroups. */
int dldkd_simpool_eval_pairs_bf16(const void* const* q_packed, const void* const* g_packed, const int32_t* lens,
                                  const int32_t* order, const int32_t* pairs, int n_waves, int nq, int nv, int L, int n_branches,
                                  int q_split, int32_t* done, void* workspace, void* stream);

/* HOST-side planning of the query split (no GPU work): the number of ranges (>= min_split when nq allows it) that
 * minimises the modelled time of dldkd_simpool_eval_bf16 on 256 CUs, and the queries per range (a multiple of 32; the
 * last range is shorter).  Pass *n_ranges as q_split. */
int dldkd_simpool_eval_plan(int nq, int nv, int n_branches, int min_split, int* n_ranges, int* queries_per_range);

/* Stage 2: (nq, nv) row-major outputs from the workspace.  inv_order[v] = pos(v).
 *     s_b[q, v]   = part_b[pos(v), q]                  what get_sim_scores returns, model.py:327-329
 *     fused[q, v] = w0 * s_0 + w1 * s_1                eval.py:254   (= s_0 when n_branches == 1)
 * fused / s0 / s1 may each be NULL. */
int dldkd_simpool_finish(const void* workspace, const int32_t* inv_order, int nq, int nv, int n_branches,
                         float w0, float w1, float* fused, float* s0, float* s1, void* stream);
/* The same for the queries [q_lo, q_hi) only (q_lo a multiple of 4): outputs are (q_hi - q_lo, nv) blocks.  q_bad: NULL or the
 * bad_flags of dldkd_pack_queries_bf16 (nq floats): flagged queries get NaN scores. */
int dldkd_simpool_finish_range(const void* workspace, const int32_t* inv_order, int nq, int nv, int n_branches,
                               float w0, float w1, int q_lo, int q_hi, const float* q_bad, float* fused, float* s0,
                               float* s1, void* stream);

/* Enqueue on `stream` a wait until *counter >= at_least (hipStreamWaitValue32, no host involvement): everything
 * enqueued on `stream` afterwards runs once the producer kernel - typically still running on another stream - has
 * pushed the device counter that far.  Used with the `done` counters of dldkd_simpool_eval_bf16 to overlap the
 * all-gather of one query range (the exchange BASELINE.json's north_star adds to eval.py:188-212) with the scoring of
 * the next. */
int dldkd_stream_wait_counter(void* stream, int32_t* counter, int32_t at_least);

/* ---------------------------------------------------------------------------------------------
 * Encoder towers, fp32 parity-grade forward (fp32-input MFMA: exact fp32 products and sums).
 * Together these replace DLDKD.encode_input / encode_context / encode_query (method/model.py:199-258).
 * ------------------------------------------------------------------------------------------- */

/* C[M,N] = act(sum_k A(m,k) * B(n,k) + bias[n]);  a_kmajor / b_kmajor = 1 when the contraction index is
 * the row index of that operand in memory.  (0,0): nn.Linear forward Y = X W^T + b
 * (model_components.py:302,388-390,442; model.py:39);  (0,1): dX = dY W;  (1,1): dW = dY^T X.
 * relu != 0 applies max(.,0) (LinearLayer, model_components.py:310-311).  bias may be NULL.
 * float4 loads are used when an operand is 16-byte aligned with a leading dimension divisible by 4.
 *
 * workspace / workspace_bytes: device scratch for split-K.  The backward layouts with a small output and a long
 * contraction (weight gradients dW = dY^T X: 384 x 3072 outputs over 16k-80k rows) are cut along K into partial planes
 * that one reduce pass sums in a fixed order (no atomics: bitwise reproducible).  dldkd_gemm_workspace_bytes() tells how
 * many bytes a shape wants (0 = this shape never splits).  NULL or too small a workspace is legal: the product is then
 * computed unsplit (same result up to fp32 summation order, slower for those shapes).  The library itself never
 * allocates: the workspace must stay valid until the work enqueued by this call has run, and two calls that may overlap
 * (different streams) need different workspaces. */
#define DLDKD_GEMM_F32 0    /* dldkd_gemm_f32   */
#define DLDKD_GEMM_F32X3 1  /* dldkd_gemm_f32x3 */
#define DLDKD_GEMM_BF16 2   /* dldkd_gemm_bf16  */
#define DLDKD_GEMM_F32X2 3  /* dldkd_gemm_f32x2: forward layout and the pooled simpool product only */
size_t dldkd_gemm_workspace_bytes(int precision, int M, int N, int K, int a_kmajor, int b_kmajor);
int dldkd_gemm_f32(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int lda,
                   int ldb, int ldc, int a_kmajor, int b_kmajor, int relu, void* workspace, size_t workspace_bytes,
                   void* stream);

/* out[row] = LayerNorm(x[row] + add[..]) * gamma + beta over the last dim D (nn.LayerNorm, eps inside the
 * sqrt).  add == NULL: plain LayerNorm (LinearLayer.LayerNorm, model_components.py:308); add_mod == L > 0:
 * add[row % L] = position rows (TrainablePositionalEncoding.forward :277-284); add_mod == 0: add[row] =
 * residual (BertSelfOutput.forward :446-450). */
int dldkd_layernorm_f32(const float* x, const float* add, int add_mod, const float* gamma, const float* beta,
                        float* out, long M, int D, float eps, void* stream);
/* LayerNorm followed by inverted dropout in ONE pass (LinearLayer: LayerNorm -> Dropout -> Linear, model_components.py:305-312;
 * TrainablePositionalEncoding: LayerNorm(x + pos) -> Dropout, :277-284): out = keep ? LN(x + add) / (1-p) : 0, keep (M x D
 * bytes, 4-byte aligned) = the mask for the backward pass.  The masks are those dldkd_dropout_fwd_f32 draws for a tensor of
 * the output's shape at the same (seed, offset) / device state. */
int dldkd_layernorm_dropout_f32(const float* x, const float* add, int add_mod, const float* gamma, const float* beta,
                                float* out, unsigned char* keep, long M, int D, float eps, float p_drop,
                                unsigned long long seed, unsigned long long offset, const unsigned long long* state,
                                void* stream);
/* dldkd_layernorm_f32 / _dropout_f32 with a row-group filter: group_flags (M / 32 bytes, M % 32 == 0, or NULL) as written by
 * dldkd_layernorm_dropout_bf16 for the padded batch this tensor's rows belong to - rows of a group flagged 0 (clips past a video's
 * length) are not read, their output row (and keep bytes) are zeros.  keep may be NULL when p_drop == 0. */
int dldkd_layernorm_groups_f32(const float* x, const float* add, int add_mod, const float* gamma, const float* beta, float* out,
                               unsigned char* keep, long M, int D, float eps, float p_drop, unsigned long long seed,
                               unsigned long long offset, const unsigned long long* state, const unsigned char* group_flags, void* stream);
/* The general form of the LayerNorm launches in this section (LayerNorm(x [+ add]) [-> inverted dropout]): any of the outputs -
 * fp32 rows, bf16 rows, TWO bf16 planes [2][M][D] (h = bf16(y), m = bf16(y - h): the operands of dldkd_gemm_bf16_nt16_planes -
 * "mixed" training feeds its forward GEMM from them and keeps plane 0 for the backward pass; any subset may be given), keep bytes (NULL with p_drop > 0: the mask is applied, not written), statistics [2][M] - and either filter: row_mask (M
 * floats; group_flags_out then receives the 32-row group flags) or group_flags_in (flags an earlier launch wrote).  Semantics of each
 * as documented at dldkd_layernorm_dropout_f32 / _groups_f32 / _dropout_bf16. */
int dldkd_layernorm_ex_f32(const float* x, const float* add, int add_mod, const float* gamma, const float* beta, float* out_f32,
                           void* out_bf16, void* out_planes, unsigned char* keep, float* stats, long M, int D, float eps, float p_drop,
                           unsigned long long seed, unsigned long long offset, const unsigned long long* state, const float* row_mask,
                           unsigned char* group_flags_out, const unsigned char* group_flags_in, void* stream);
/* The fp32-row twin of dldkd_layernorm_dropout_bf16 (parity mode): same row mask / group flags / statistics, fp32 output. */
int dldkd_layernorm_dropout_rows_f32(const float* x, const float* gamma, const float* beta, float* out, unsigned char* keep, float* stats,
                                     long M, int D, float eps, float p_drop, unsigned long long seed, unsigned long long offset,
                                     const unsigned long long* state, const float* row_mask, unsigned char* group_flags, void* stream);

/* The same LayerNorm (+ inverted dropout when p_drop > 0: same masks) writing the row as bf16 (round to nearest even) - the
 * operand form of the bf16 GEMMs that consume it (dldkd_gemm_bf16_mixed) - and, when stats != NULL, the row statistics
 * (mean -> stats[row], rstd -> stats[M + row]) the backward pass would otherwise recompute.  Training input projection in
 * throughput mode: LinearLayer.forward's LayerNorm -> Dropout (method/model_components.py:305-310).  keep may be NULL: the dropout
 * is applied all the same and no mask is written (the bits are Philox4x32-10 on the flat element index; dldkd_inproj_bwd_bf16 draws
 * the few it needs again from the same p_drop / seed / offset / state); out_bf16 8-byte aligned.  row_mask (M floats, or NULL): rows with row_mask[row] == 0 - the clips past a video's
 * length in a padded batch (collate_train's mask, method/data_provider.py:75-86) - are not read; their output row, keep bytes
 * and statistics are zeros (no loss term depends on them and their gradients are exactly zero).  group_flags (M / 32 bytes, or
 * NULL; needs row_mask, M % 32 == 0 and a padded length that is a multiple of 32): 1 when the 32-row group starts with a valid
 * row - what dldkd_gemm_bf16_mixed(dw = 1, ..., k_flags) skips by. */
int dldkd_layernorm_dropout_bf16(const float* x, const float* gamma, const float* beta, void* out_bf16, unsigned char* keep, float* stats,
                                 long M, int D, float eps, float p_drop, unsigned long long seed, unsigned long long offset,
                                 const unsigned long long* state, const float* row_mask, unsigned char* group_flags, void* stream);

/* BertSelfAttention.forward (model_components.py:398-436) for N sequences of L <= 128 tokens, 4 heads x 96:
 * qkv (N, L, 1152) = [query | key | value] projections, mask (N, L) 0/1 or NULL, out (N, L, 384) context
 * layer.  softmax(QK^T / sqrt(96) + (1 - mask) * -10000) V; the L x L probabilities never leave the CU. */
int dldkd_attention_fwd_f32(const float* qkv, const float* mask, float* out, int N, int L, void* stream);
/* Same contract on bf16 MFMA (Q, K, V and the probabilities rounded to bf16; softmax statistics and output fp32):
 * the throughput-mode attention of the towers.  qkv_is_bf16 != 0: qkv is a (N, L, 1152) bf16 buffer. */
int dldkd_attention_fwd_bf16(const void* qkv, const float* mask, float* out, int N, int L, int qkv_is_bf16, void* stream);

/* get_modularized_queries (model.py:245-258): h (N, L, 384), mask (N, L), w (384) -> out (N, 384);
 * attn (N, L) optional (softmax weights, kept for the backward pass).  L <= 64. */
int dldkd_modpool_fwd_f32(const float* h, const float* mask, const float* w, float* out, float* attn, int N,
                          int L, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K4: 16-bit input projection of the raw features (inference), the HBM-bound stage of the gallery encode.
 * "h16" in the names of this and the K5 section = the 16-bit operand format of the EVAL-path towers: IEEE fp16 (10 mantissa bits),
 * fp32 accumulation - not bf16: same MFMA rate, and with bf16 operands the K = 3072 projection alone moved the fused scores by
 * 1.8e-4, enough to put R@K a coin flip away from the reference's +-0.1 (dl-dkd_amd/csrc/common.hpp, profiles/r05/).  Every
 * operand of these kernels is a LayerNorm output, a weight, a probability or an L2-normalised feature: far inside fp16's range.
 * The SCORER (dldkd_simpool_eval_*) keeps bf16 operands, and so does training.  Buffers called Wf / Wfrag / blob / x_h16 / y_h16
 * hold fp16 values.
 * ------------------------------------------------------------------------------------------- */

/* Fold LayerNorm(K) into the following Linear(K -> N): Wf[n,k] = h16(gamma[k] * W[n,k]),
 * cs[n] = sum_k Wf[n,k], bb[n] = sum_k beta[k] * W[n,k] + bias[n]  (LinearLayer, model_components.py:294-312).
 * For two branches call it twice on the two halves of one (768, K) Wf buffer. */
int dldkd_fold_ln_linear_h16(const float* W, const float* bias, const float* gamma, const float* beta, int N, int K,
                              void* Wf, float* cs, float* bb, void* stream);

/* y_b[m, :] = act( LayerNorm(x[m, :]) . W_b^T + b_b ) for b < N/384 branches, from the folded weights:
 * rstd[m] * (h16(x[m,:]) . Wf[n,:] - mean[m] * cs[n]) + bb[n];  x (M, K) fp32 is read ONCE for all branches,
 * mean / rstd are accumulated from the same tiles.  N = 384 (y1 unused) or 768; K a multiple of 32.
 * Replaces LinearLayer.forward (model_components.py:305-312) on the inference path. */
int dldkd_in_proj_h16(const float* x, const void* Wf, const float* cs, const float* bb, float* y0, float* y1, long M,
                       int N, int K, float eps, int relu, void* stream);

/* Full-row variant for two branches (N = 768): Wfrag holds the folded weights in MFMA B-fragment order
 * [k-tile of 32][32-column tile (24)][kk (2)][lane (64)][8 h16]; fold each branch with n_offset = 0 / 384 into
 * the same Wfrag / cs / bb buffers.  One workgroup computes all 768 columns of its 128 rows, so x is converted to
 * h16 and its LayerNorm sums are taken once.  Same result as dldkd_in_proj_h16 with N = 768. */
int dldkd_fold_ln_linear_h16_frag(const float* W, const float* bias, const float* gamma, const float* beta, int N, int K,
                                   int n_offset, void* Wfrag, float* cs, float* bb, void* stream);
int dldkd_in_proj_h16_full(const float* x, const void* Wfrag, const float* cs, const float* bb, float* y0, float* y1,
                            long M, int K, float eps, int relu, void* stream);
/* Second-generation kernel for the same contract (same Wfrag / cs / bb from dldkd_fold_ln_linear_h16_frag with n_total = 768;
 * same result up to fp32 summation order: the k-tiles are summed in a rotated order).  One 128-row x 768-column tile per
 * workgroup of 4 waves (128 rows x 192 columns each, 384 accumulator registers), x and W' streamed by LDS-DMA through rings that
 * fill all 160 KiB of LDS, hand-counted waits, no branch in the k-loop (in_proj_rows128.hip).  Needs K % 64 == 0, K >= 128:
 * dldkd_in_proj_h16_rows128_ok(K) tells; otherwise use dldkd_in_proj_h16_full.  Wfrag must be 16-byte aligned. */
int dldkd_in_proj_h16_rows128(const float* x, const void* Wfrag, const float* cs, const float* bb, float* y0, float* y1,
                               long M, int K, float eps, int relu, void* stream);
int dldkd_in_proj_h16_rows128_ok(int K);
/* The same kernel over a ROW-GROUP TABLE: x and y0 / y1 are (M, K) / (M, 384) as above, but only the rows of the listed groups
 * are read and written.  groups[g] = first row of a group of 32 consecutive rows (g < n_groups, n_groups a multiple of 4: pad by
 * repeating a group; groups[g] + 32 <= M).  For a padded (n, L, K) batch with L a multiple of 32 the host lists the groups that
 * hold valid clips (video v, clips 32 t ..: row v L + 32 t for t < ceil(len_v / 32)): the rows of the padding are neither
 * projected nor written - 29 % fewer rows on a TVR-like length mix. */
int dldkd_in_proj_h16_rows128_groups(const float* x, const void* Wfrag, const float* cs, const float* bb, float* y0, float* y1,
                                      long M, int K, float eps, int relu, const int32_t* groups, long n_groups, void* stream);
/* K4b: the same projection on rows in their RESIDENT form - h16 features (M, K) plus the rows' fp32 LayerNorm statistics
 * mean[M], rstd[M] - written once by dldkd_rows_to_h16_stats when a dataset's raw features become device-resident (the
 * reference re-reads, re-pads and re-uploads the fp32 features of the validation videos in every epoch: method/eval.py:114-175
 * through method/data_provider.py:111-136).  Same arithmetic as dldkd_in_proj_h16_rows128 (which rounds x to h16 before the
 * MFMA and takes the statistics from the fp32 values), same Wfrag / cs / bb, same row-group table (groups == NULL: all rows,
 * tile t = rows 128 t ..; the table rows index x, mean, rstd and y alike) - without the fp32 fragment reads, conversions and
 * LayerNorm sums in the k-loop and with half the bytes per row (in_proj_rows128b.hip).  Needs K % 64 == 0, K >= 256
 * (dldkd_in_proj_h16_rows128b_ok); x_h16 and Wfrag 16-byte aligned. */
int dldkd_in_proj_h16_rows128b(const void* x_h16, const float* mean, const float* rstd, const void* Wfrag, const float* cs,
                                const float* bb, float* y0, float* y1, long M, int K, int relu, const int32_t* groups, long n_groups,
                                void* stream);
int dldkd_in_proj_h16_rows128b_ok(int K);
/* The same with h16 output rows (round to nearest even, row stride 384 h16): what the fused tower reads through
 * dldkd_tower_seq_h16_rows16 - half the bytes written here and read there (2 x 1.28 GB instead of 2 x 2.56 GB per branch pair at
 * TVR's 1.67 M clips), and the tower's prologue becomes one round of 16-byte loads straight into its operand registers. */
int dldkd_in_proj_h16_rows128b_out16(const void* x_h16, const float* mean, const float* rstd, const void* Wfrag, const float* cs,
                                      const float* bb, void* y0_h16, void* y1_h16, long M, int K, int relu, const int32_t* groups,
                                      long n_groups, void* stream);
/* PARITY-grade two-branch input projection (in_proj_rows128x3.hip): y = ReLU(LayerNorm(x) W^T + b) with fp32-grade products
 * (three bf16 planes per operand, six MFMAs per product: the scheme of dldkd_gemm_f32x3), both branches in one pass.
 *   dldkd_row_meanrstd_f32: mean[M], rstd[M] of the rows exactly as dldkd_layernorm_f32 computes them (D % 4 == 0, D <= 4096).
 *   dldkd_fold_ln_linear_planes: W' = gamma (.) W split into three bf16 planes in the kernel's fragment order
 *       (Wplanes: 3 * 768 * K * 2 bytes for the two branches, 16-byte aligned; call once per branch with n_offset 0 / 384),
 *       bb[n_offset + n] = W[n].beta + bias[n].
 *   dldkd_in_proj_f32x3_rows128: out = relu(((x - mean) * rstd) . W'^T + bb); K % 32 == 0, 64 <= K <= 4096
 *       (dldkd_in_proj_f32x3_rows128_ok).  Replaces reference LinearLayer.forward (method/model_components.py:305-312) on the
 *       inference path of parity mode. */
int dldkd_row_meanrstd_f32(const float* x, float* mean, float* rstd, long M, int D, float eps, void* stream);
/* Training backward of the input projection on RAW features (LinearLayer: LayerNorm -> Dropout -> Linear -> ReLU,
 * model_components.py:294-312; the features need no gradient): LayerNorm's parameter gradients straight from the accumulators of
 * dz' = dy W (the dX layout of dldkd_gemm_bf16 / dldkd_gemm_f32x3; precision = DLDKD_GEMM_BF16 or DLDKD_GEMM_F32X3) - dz' is never written and no LayerNorm
 * backward pass runs over x:
 *     dz = dz' (.) keep * keep_scale,  dgamma[k] += sum_m dz[m,k] (x[m,k] - mean[m]) rstd[m],  dbeta[k] += sum_m dz[m,k].
 * dy (M, N) gradient behind the ReLU, W (N, K), x (M, K), keep (M, K) bytes of dldkd_layernorm_dropout_f32 or NULL, mean / rstd
 * from dldkd_row_meanrstd_f32; workspace >= 2 * ceil(M / 128) * K floats; dgamma / dbeta (K) zero-initialised by the caller.  *   row_flags (M / 32 bytes from dldkd_layernorm_dropout_bf16, or NULL; bf16 precision, M % 128 == 0): 32-row groups flagged 0 are
 *   rows of the padding (dy and the statistics are zero there) - neither loaded nor multiplied. */
int dldkd_linear_lngrad(int precision, const float* dy, const float* W, const float* x, const unsigned char* keep, float keep_scale,
                        const float* mean, const float* rstd, float* workspace, size_t workspace_bytes, float* dgamma,
                        float* dbeta, long M, int N, int K, const unsigned char* row_flags, void* stream);
int dldkd_fold_ln_linear_planes(const float* W, const float* bias, const float* gamma, const float* beta, int N, int K, int n_offset,
                                void* Wplanes, float* bb, void* stream);
int dldkd_in_proj_f32x3_rows128(const float* x, const float* mean, const float* rstd, const void* Wplanes, const float* bb, float* y0,
                                float* y1, long M, int K, int relu, void* stream);
int dldkd_in_proj_f32x3_rows128_ok(int K);
/* The same kernel as a plain fp32-grade linear layer over full rows (the 384 / 768-wide linears of the towers, parity mode,
 * inference): y = act(x W^T + b), N = 384 (y0) or 768 (columns [0, 384) -> y0, [384, 768) -> y1), output row stride ldy floats
 * (so q | k | v can land side by side in one buffer), mean / rstd NULL (or both given: rows are normalised first).
 * dldkd_pack_linear_planes: the weights of one nn.Linear (N % 32 == 0 columns at n_offset of n_total = 384 | 768) as three bf16
 * planes in the kernel's fragment order (3 * n_total * K * 2 bytes), bb = bias (gamma, beta NULL) or the LayerNorm fold. */
int dldkd_pack_linear_planes(const float* W, const float* bias, const float* gamma, const float* beta, int N, int K, int n_offset,
                             int n_total, void* Wplanes, float* bb, void* stream);
int dldkd_linear_f32x3_rows(const float* x, const float* mean, const float* rstd, const void* Wplanes, const float* bb, float* y0,
                            float* y1, long M, int N, int K, int ldy, int relu, void* stream);
/* Diagnostics (tools/k4_timeline.py): the same launch, plus 12 u64 per workgroup in `stamps` (size it for ceil(M / 128)
 * workgroups; the kernel is persistent and uses fewer): s_memtime / s_memrealtime at kernel start, first k-loop start / end, first
 * epilogue end, kernel end; the XCC id; the number of tiles the workgroup did. */
int dldkd_debug_in_proj_rows128_timeline(const float* x, const void* Wfrag, const float* cs, const float* bb, float* y0, float* y1,
                                         long M, int K, float eps, int relu, unsigned long long* stamps, void* stream);

/* Plain y = act(x W^T + b) on the same full-row 16-bit MFMA kernel (h16 operands, no LayerNorm fold) for the 384-wide linears of the
 * towers in throughput mode (model_components.py:388-390 query/key/value, :442 dense; model.py:39 out_mapping_linear).
 * dldkd_pack_linear_h16_frag writes rows [n_offset, n_offset + N) of a weight block of n_total (384 or 768) output
 * columns in MFMA fragment order (h16) and its bias into bb[n_offset ..]; call it once per source matrix.
 * dldkd_linear_rows_h16: x (M, K) fp32 contiguous; output columns [0, 384) go to y0 and [384, 768) to y1, both with
 * row stride ldy elements (so a (M, 1152) q|k|v buffer is filled by one N = 768 and one N = 384 launch).  out_bf16 != 0:
 * y0 / y1 are bf16 buffers (ldy a multiple of 8) - what dldkd_attention_fwd_bf16 consumes with qkv_is_bf16 != 0. */
int dldkd_pack_linear_h16_frag(const float* W, const float* bias, int N, int K, int n_offset, int n_total, void* Wfrag, float* bb,
                                void* stream);
int dldkd_linear_rows_h16(const float* x, const void* Wfrag, const float* bb, void* y0, void* y1, int ldy, long M, int N, int K,
                           int relu, int out_bf16, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training path, fp32 (DLDKD.forward + backward, method/model.py:100-197,353-387;
 * method/model_components.py:106-234).  Heavy contractions = dldkd_gemm_f32 / _f32x3 / _bf16; the rest is
 * row-wise.  "d*" pointers are gradients; functions documented "+=" accumulate into zero-initialised or
 * partially filled buffers.
 * ------------------------------------------------------------------------------------------- */

/* Throughput-mode twin of dldkd_gemm_f32: same arguments and operand layouts, fp32
 * operands in memory converted to bf16 on the way to LDS, bf16 MFMA with fp32 accumulation.  Used by the training
 * step when the precision is set to "bf16" (BASELINE.json configs[2]); the fp32 entry points remain the parity path. */
int dldkd_gemm_bf16(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int lda, int ldb,
                    int ldc, int a_kmajor, int b_kmajor, int relu, void* workspace, size_t workspace_bytes, void* stream);
/* The two GEMMs of the training input projection whose activation operand is stored as bf16 (dldkd_layernorm_dropout_bf16),
 * bf16 MFMA with fp32 accumulation as dldkd_gemm_bf16 (which would round the same values to bf16 itself: same result):
 *   dw == 0  forward  C[M, N] = act(A16[M, K] . B[N, K]^T + bias)   A16 bf16 row-major (lda elements, lda % 4 == 0), B fp32 (N, K)
 *   dw != 0  dW       C[M, N] = sum_k A[k, m] B16[k, n]             A fp32 (K, M) = dy, B16 bf16 (K, N) = the saved rows (ldb, N even);
 *                     no bias / ReLU; split-K like dldkd_gemm_bf16 with a workspace of
 *                     dldkd_gemm_workspace_bytes(DLDKD_GEMM_BF16, M, N, K, 1, 1) bytes (NULL: no split); k_flags (one byte per 32
 *                     consecutive k, or NULL): tiles flagged 0 hold zero rows (padding) and are skipped;
 *   dw == 2           the dW layout with B fp32 (K, N) as well (dldkd_gemm_bf16's) plus k_flags;
 *   dw == 3           the dW layout with A bf16 (K, M) too (lda, M even): both operands are saved bf16 rows (fused training towers);
 *   dw == 0           k_flags, if given, are per 32 ROWS of A16 / C (M % 128 == 0): the groups flagged 0 are not multiplied. */
int dldkd_gemm_bf16_mixed(int dw, const void* A, const void* B, const float* bias, float* C, int M, int N, int K, int lda, int ldb,
                          int ldc, int relu, void* workspace, size_t workspace_bytes, const unsigned char* k_flags, void* stream);
/* dldkd_layernorm_dropout_bf16 for BOTH branches' input projections over the same raw rows in one pass (the inheritance and the
 * exploration video tower normalise the same student features: method/model.py:229-243, model_components.py:305-310): x is read
 * once, mean / rstd taken once (stats: shared by the two backward passes), the row written twice - out0 = LN(x; gamma0, beta0)
 * with the dropout bits of (seed, offset0), out1 likewise with (gamma1, beta1, offset1) - exactly what two calls of
 * dldkd_layernorm_dropout_bf16 with keep == NULL write.  row_mask / group_flags as there.  planes != 0: out0 / out1 are [2][M][D], the
 * second bf16 plane (m = bf16(y - bf16(y))) behind the first - the two-plane GEMM operands of the "mixed" training precision. */
int dldkd_layernorm_dropout_bf16_dual(const float* x, const float* gamma0, const float* beta0, const float* gamma1, const float* beta1,
                                      void* out0_bf16, void* out1_bf16, float* stats, long M, int D, float eps, float p_drop,
                                      unsigned long long seed, unsigned long long offset0, unsigned long long offset1,
                                      const unsigned long long* state, const float* row_mask, unsigned char* group_flags, int planes,
                                      void* stream);
/* dldkd_gemm_bf16_nt with BOTH operands bf16 in memory (A (M, K) = the rows dldkd_layernorm_dropout_bf16 writes, B (N, K) = the
 * weight cast by dldkd_cast_bf16): k-tiles of 64, no conversion on the way to the MFMA, half the dependent tile round trips.  The
 * forward GEMM of the training input projection (LinearLayer.forward, model_components.py:305-312).  K % 64 == 0, lda / ldb % 8 == 0,
 * 16-byte aligned operands (dldkd_gemm_bf16_nt16_ok); row_flags as dldkd_gemm_bf16_nt. */
int dldkd_cast_bf16(const float* x, void* y, long n, void* stream);
int dldkd_gemm_bf16_nt16_ok(int M, int N, int K, int lda, int ldb);
int dldkd_gemm_bf16_nt16(const void* A, const void* B, const float* bias, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                         int relu, const unsigned char* row_flags, void* stream);
/* Two-plane operands of the "mixed" training precision's forward GEMMs.  dldkd_split2_bf16_jobs: up to 12 jobs in one launch, job j
 * splitting host_n[j] fp32 values (a multiple of 4) into h = bf16(x) -> host_dst_h[j] and m = bf16(x - h) -> host_dst_m[j] (kind 0), or
 * copying them as fp32 to host_dst_h[j] (kind 1: e.g. the three attention biases into one vector) - a tower's weights once per step.
 * dldkd_gemm_bf16_nt16_planes: C = act(A B^T + bias) from A_planes [2][M][lda] and B_planes [2][N][ldb] (plane strides in elements,
 * multiples of 8): a_h b_h + a_h b_m + a_m b_h, smallest term first, as ONE pass of dldkd_gemm_bf16_nt16's LDS-DMA kernel over three
 * K-long segments - the two-plane product of dldkd_gemm_f32x2 without the split on the way to LDS (2.5x its rate at the tower shapes).
 * Constraints of dldkd_gemm_bf16_nt16. */
int dldkd_split2_bf16_jobs(const float* const* host_src, void* const* host_dst_h, void* const* host_dst_m, const long* host_n,
                           const int* host_kind, int njobs, void* stream);
int dldkd_gemm_bf16_nt16_planes(const void* A_planes, const void* B_planes, const float* bias, float* C, int M, int N, int K, int lda, int ldb,
                                int ldc, int relu, const unsigned char* row_flags, long a_plane_stride, long b_plane_stride, void* stream);
/* dldkd_gemm_bf16_mixed(dw = 1 or 3) with the bias gradient on the side: a_colsum[m] += sum_k A[k, m] over the k-tiles that are not
 * skipped (fp32 atomics into a buffer zeroed by the caller) - in nn.Linear's backward pass A is dY, so this is the bias gradient, taken
 * from the A tiles on their way to LDS instead of by a second pass over dY (ldc = N, no bias / ReLU). */
int dldkd_gemm_bf16_dw_bias(int dw, const void* A, const void* B, float* C, int M, int N, int K, int lda, int ldb, void* workspace,
                            size_t workspace_bytes, const unsigned char* k_flags, float* a_colsum, void* stream);

/* Backward pass of the training input projection in bf16 mode as ONE weight-gradient GEMM with two accumulator sets (replaces
 * dldkd_gemm_bf16_dw_bias + dldkd_linear_lngrad for LinearLayer on raw features, method/model_components.py:294-312):
 *     dW[n, k]  = sum_m dy[m, n] z[m, k]                      z = the saved bf16 LayerNorm-dropout rows (dldkd_layernorm_dropout_bf16)
 *     H[n, k]   = sum_m dy[m, n] [z[m, k] != 0]
 *     dbeta[k]  = keep_scale sum_n W[n, k] H[n, k]            dgamma[k] = (sum_n W[n, k] dW[n, k] - beta[k] dbeta[k]) / gamma[k]
 * - the (M, K) product dy W of dldkd_linear_lngrad reassociated into the M-long contraction of the weight gradient.  Columns with
 * |gamma[k]| < 0.05 (z holds no trace of xhat there) are recomputed exactly from x / the dropout bits / mean / rstd (rstd = 0 marks
 * padding rows).  The dropout bits: `keep` (M, K) bytes as dldkd_layernorm_dropout_bf16 wrote them, or keep == NULL and (p_drop, seed,
 * offset, state) = the arguments that launch was given - the bits are then drawn again (Philox4x32-10 on the flat element index),
 * and the forward pass need not write a keep byte per element.
 * dy (M, N) fp32, N <= 384; z (M, K) bf16; W (N, K); dW (N, K); dbias (N) zeroed by the caller or NULL; dgamma, dbeta (K) ZEROED by
 * the caller (used as accumulators); k_flags as in dldkd_gemm_bf16_mixed; workspace: dldkd_inproj_bwd_workspace_bytes.
 * dy_bf16: dy rounded to bf16 (M, N) as dldkd_tower_train_b1 leaves it beside the fp32 rows, or NULL (the rows are then cast into the
 * workspace first).  The GEMM (gemm_bf16_tn.hip: LDS-DMA row tiles read back transposed) contracts the bf16 rows; dbias is the column
 * sum of those bf16 rows.
 * (Three launches: the GEMM, the plane reduce + dot products, the finalisation.  Merging the last two behind an arrival ticket was
 * measured: the __threadfence of every workgroup made the finish 2-4 x slower than the two kernels.) */
size_t dldkd_inproj_bwd_workspace_bytes(int N, int K, long M);
int dldkd_inproj_bwd_bf16(const float* dy, const void* z_bf16, const float* W, const float* gamma, const float* beta,
                          float keep_scale, const float* x, const unsigned char* keep, float p_drop, unsigned long long seed,
                          unsigned long long offset, const unsigned long long* state, const float* mean, const float* rstd,
                          float* dW, float* dbias, float* dgamma, float* dbeta, long M, int N, int K, void* workspace,
                          size_t workspace_bytes, const unsigned char* k_flags, const void* dy_bf16, void* stream);
/* The forward layout of dldkd_gemm_bf16 - C[M, N] = act(A[M, K] . B[N, K]^T + bias), both operands fp32 and k-minor (a Linear's
 * forward pass; its input gradient once the weight is transposed) - with the operand tiles staged HBM -> LDS by LDS-DMA instead
 * of through registers (gemm_bf16_dma.hip): same products in the same order, bit-identical results, about half the time on
 * the towers' 16,384-row x 384 / 1152-wide layers, which are latency-bound in the register-staged kernel.  Needs K % 32 == 0,
 * lda / ldb % 4 == 0, 16-byte aligned operands (dldkd_gemm_bf16_nt_ok); no split-K.  row_flags (M / 32 bytes, or NULL; M % 128 == 0):
 * 32-row groups of A flagged 0 are rows of the padding - not loaded, not multiplied; their C rows are act(bias). */
int dldkd_gemm_bf16_nt(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                       int relu, const unsigned char* row_flags, void* stream);
int dldkd_gemm_bf16_nt_ok(int M, int N, int K, int lda, int ldb);

/* fp32-GRADE GEMM on the bf16 matrix cores: each fp32 operand is split into three bf16 planes (h + m + l = 24 mantissa
 * bits) on the way to LDS and every product is rebuilt from the six plane products of order <= 2 with fp32 accumulation
 * (relative error ~2^-24 per product, like a true fp32 multiply).  Same arguments as dldkd_gemm_f32.  6 bf16
 * MFMAs replace 8 fp32-input MFMAs that each run 2x slower: ~3x the throughput of dldkd_gemm_f32 at parity-grade
 * accuracy; the host mirror uses it for precision "fp32" and keeps the true fp32-input MFMA as "fp32_exact". */
int dldkd_gemm_f32x3(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int lda, int ldb,
                     int ldc, int a_kmajor, int b_kmajor, int relu, void* workspace, size_t workspace_bytes, void* stream);
/* The two-plane form of dldkd_gemm_f32x3 for the FORWARD layout (C = act(A B^T + bias), A (M, K), B (N, K) row-major): every fp32
 * operand split into two bf16 planes, three MFMAs per product instead of six, ~2^-16 relative error per product instead of 2^-24.
 * The forward pass of the "mixed" training precision (ops.set_gemm_precision("mixed")): measured on the seven losses of the C3 / C5
 * steps it stays inside north_star's 1e-4 with a wide margin (tests/test_train_mode_gpu.py), at two thirds of the three-plane time.
 * row_flags as dldkd_gemm_f32x3_flags (forward layout). */
int dldkd_gemm_f32x2(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                     int relu, const unsigned char* row_flags, void* stream);
/* dldkd_gemm_f32x3 with a filter for the rows of the padding: flags = one byte per 32 rows of the ACTIVATION operand - A's rows
 * when A is k-minor (forward / dX; M % 128 == 0: groups flagged 0 are not multiplied, their C rows come out as act(bias)), the
 * contraction index when both operands are k-major (dW: the k-tiles of groups flagged 0 - zero rows of dy - are skipped). */
int dldkd_gemm_f32x3_flags(const float* A, const float* B, const float* bias, float* C, int M, int N, int K, int lda, int ldb,
                           int ldc, int a_kmajor, int b_kmajor, int relu, void* workspace, size_t workspace_bytes,
                           const unsigned char* flags, void* stream);


/* Backward of dldkd_layernorm_f32: dx (may be NULL) written, dgamma / dbeta += (zero-initialised by the caller).
 * keep / keep_scale: NULL / any, or the byte mask and 1/(1-p) of dldkd_layernorm_dropout_f32: dy is masked and scaled on
 * load (the backward of the fused dropout), so no separate dldkd_mask_scale_f32 pass over dy is needed. */
int dldkd_layernorm_bwd_f32(const float* x, const float* add, int add_mod, const float* gamma, const float* dy,
                            float* dx, float* dgamma, float* dbeta, long M, int D, float eps, const unsigned char* keep,
                            float keep_scale, void* stream);
/* ... and its backward pass: rows of a group flagged 0 contribute nothing to dgamma / dbeta and get a zero dx row (their dy is
 * zero: no loss term depends on a padded clip); neither x nor dy is read there. */
int dldkd_layernorm_bwd_groups_f32(const float* x, const float* add, int add_mod, const float* gamma, const float* dy, float* dx,
                                   float* dgamma, float* dbeta, long M, int D, float eps, const unsigned char* keep, float keep_scale,
                                   const unsigned char* group_flags, void* stream);

/* out[c] += sum_r x[r, c]  (bias gradients; position-table gradient with x viewed as (batch, L*D)). */
int dldkd_colsum_f32(const float* x, float* out, long M, long N, void* stream);
/* dy[i] = 0 where y[i] <= 0 (ReLU backward, in place). */
int dldkd_relu_bwd_f32(float* dy, const float* y, long n, void* stream);
/* a += alpha * b. */
int dldkd_axpy_f32(float* a, const float* b, float alpha, long n, void* stream);
/* out = a * m * scale  (dropout masks, forward and backward). */
int dldkd_mul_f32(const float* a, const float* m, float scale, float* out, long n, void* stream);

/* Inverted dropout in one pass (reference: nn.Dropout in model_components.py / model.py input_drop, drop).
 * keep[i] = Philox4x32-10(key = seed, counter = offset + i/4)[i%4] >= p * 2^32;  out = keep ? x / (1-p) : 0.
 * The caller advances `offset` by ceil(n/4) per call (the host side draws seed/offset from torch's CUDA generator,
 * so torch.manual_seed makes a run reproducible).  state: NULL, or two device uint64 {seed, base offset}: the kernel
 * then uses state[0] as the seed and state[1] + offset as the offset - a hipGraph-captured training step bakes `offset`
 * (the call's position inside the step) into the graph and refreshes `state` before every replay, so replays draw new
 * masks and a captured run equals the eager run bit for bit.  dldkd_mask_scale_f32 is the backward:
 * out = keep ? a*scale : 0 (out may alias a).  Buffers 16-byte aligned, keep 4-byte aligned. */
int dldkd_dropout_fwd_f32(const float* x, float* out, unsigned char* keep, long n, float p, unsigned long long seed,
                          unsigned long long offset, const unsigned long long* state, void* stream);
int dldkd_mask_scale_f32(const float* a, const unsigned char* keep, float scale, float* out, long n, void* stream);

/* F.normalize(x, dim=-1) (eps 1e-12, model.py:318-319): y, inv (1/norm per row); and its backward. */
int dldkd_normalize_rows_fwd_f32(const float* x, float* y, float* inv, long M, int D, void* stream);
int dldkd_normalize_rows_bwd_f32(const float* y, const float* inv, const float* dy, float* dx, long M, int D,
                                 void* stream);

/* mask_logits + max over clips (model.py:325-327,347-349) on S (nq, nv, L): entries l >= lens[v] are set to
 * exactly -1e10 in place; pooled (nq, nv) = max_l, arg = its first index.  Backward: dS[q,v,arg] += dpooled. */
int dldkd_clip_pool_fwd_f32(float* S, const int32_t* lens, float* pooled, int32_t* arg, int nq, int nv, int L,
                            void* stream);
int dldkd_clip_pool_bwd_f32(const float* dpooled, const int32_t* arg, const int32_t* lens, float* dS, int nq,
                            int nv, int L, void* stream);

/* Backward of dldkd_modpool_fwd_f32: dh (N, L, 384) written, dw (384) += . */
int dldkd_modpool_bwd_f32(const float* h, const float* mask, const float* w, const float* attn, const float* dout,
                          float* dh, float* dw, int N, int L, void* stream);

/* compute_kl_loss(mode='frame_score') (model.py:183-197): out[q] (may be NULL) = KL(softmax(St/temp) ||
 * softmax(Sp/temp)) over the first lens[label_q] clips of Sp/St[q, label_q, :] ((nq, nv, L) clip scores; nv == 0: the
 * compact form, Sp / St are (nq, L) = the positive column of every query as dldkd_simpool_train_fwd_f32 produces it);
 * dSp (may be NULL) += *g * d out[q] / d Sp.  g (here and in the two losses below): the upstream gradient as a DEVICE
 * scalar - autograd hands it over on the device; reading it back would put one host synchronisation per loss into
 * every training step and make the step impossible to capture into a hipGraph.  Ignored (may be NULL) when no gradient
 * output is requested. */
int dldkd_kl_frame_f32(const float* Sp, const float* St, const int32_t* labels, const int32_t* lens, float temp,
                       int nq, int nv, int L, float* out, float* dSp, const float* g, void* stream);

/* Symmetric InfoNCE on raw pooled scores S (nq, nv) (clip_nce_soft / clip_nce, model_components.py:126-234).
 * T = soft-label source scores or NULL (hard labels); rows q >= hardQ / columns v >= hardV use
 * (1-beta) * softmax(T) + beta * onehot; cq[nq], cv[nv] = per-row / per-column weights (alpha, 1/count
 * folded in by the host); eps = 1e-12 (soft) or 0 (clip_nce).  terms (nq + nv, may be NULL) = weighted
 * row and column terms (their sum is the loss); dS (may be NULL) = g * dloss/dS (written);
 * dT (may be NULL) = g * dloss/dT through the soft targets (written). */
int dldkd_nce_f32(const float* S, const float* T, const int32_t* labels, const float* cq, const float* cv,
                  int hardQ, int hardV, float beta, float eps, int nq, int nv, float* terms, float* dS, float* dT,
                  const float* g, void* stream);

/* get_clip_triplet_loss (model.py:353-387) on pooled cosine scores C (nq, nv): r_t2v[q] in [1, nv) = the
 * reference's torch.randint draw (rank of the sampled negative in the row sorted descending with the
 * positive first); r_v2t[v] = 0-based rank among the other queries' scores (ignored when hard != 0: the
 * hardest negative).  terms (nq + nv) = per-query / per-video hinge terms already divided by nq / nv;
 * dC (may be NULL) += g * dloss/dC. */
int dldkd_triplet_f32(const float* C, const int32_t* labels, const int32_t* r_t2v, const int32_t* r_v2t, int hard,
                      float margin, int nq, int nv, float* terms, float* dC, const float* g, void* stream);

/* Training form of BertSelfAttention.forward (model_components.py:398-436), fused: forward with dropout on the attention
 * probabilities, and its backward, for N sequences of L <= 128 tokens, 4 heads x 96, exact fp32 products (fp32-input MFMA).
 *   fwd: qkv (N, L, 1152), mask (N, L) 0/1 or NULL -> probs (N, 4, L, L) = softmax(QK^T / sqrt(96) + (1 - mask) * -10000)
 *        BEFORE dropout (saved for the backward pass), out (N, L, 384) = dropout(probs) V.
 *   bwd: dout (N, L, 384) -> dqkv (N, L, 1152) (written in full); dS (N, 4, L, L) is scratch; probs is OVERWRITTEN with the
 *        dropped probabilities (a saved tensor of a graph that is walked once).
 * Dropout: keep iff Philox4x32-10(seed, offset + idx / 4)[idx % 4] >= p * 2^32 on the flat index of probs - the masks
 * dldkd_dropout_fwd_f32 would draw for that tensor with the same (seed, offset); p_drop = 0 disables it.  state: NULL or
 * device {seed, base offset} as in dldkd_dropout_fwd_f32 (hipGraph-captured step).  probs may be NULL in the forward call (not written: a caller whose backward pass
 * recomputes them - the "mixed" training precision). */
int dldkd_attention_train_fwd_f32(const float* qkv, const float* mask, float* probs, float* out, int N, int L, float p_drop,
                                  unsigned long long seed, unsigned long long offset, const unsigned long long* state,
                                  void* stream);
int dldkd_attention_train_bwd_f32(const float* qkv, const float* dout, float* probs, float* dS, float* dqkv, int N, int L,
                                  float p_drop, unsigned long long seed, unsigned long long offset,
                                  const unsigned long long* state, void* stream);
/* The same two passes with every product on the bf16 matrix cores (throughput mode, attention_train_bf16.hip): q, k, v, dO
 * and the probabilities are rounded to bf16 as MFMA operands, accumulation and softmax stay fp32.  Nothing is saved between the
 * passes: the backward pass recomputes the probabilities from qkv and `mask` (same Philox keep bits as the forward pass) and
 * writes dqkv in ONE kernel; L <= 32 runs four (sequence, head) pairs per workgroup. */
int dldkd_attention_train_fwd_bf16(const float* qkv, const float* mask, float* out, int N, int L, float p_drop,
                                   unsigned long long seed, unsigned long long offset, const unsigned long long* state,
                                   void* stream);
int dldkd_attention_train_bwd_bf16(const float* qkv, const float* mask, const float* dout, float* dqkv, int N, int L, float p_drop,
                                   unsigned long long seed, unsigned long long offset, const unsigned long long* state,
                                   void* stream);
/* The same two kernels with qkv / out / dout / dqkv stored as bf16 (the fused training towers below keep every activation in
 * bf16) and the sequences' valid lengths: rows >= lens[n] of a sequence are never read (they enter as zero rows: padded keys are
 * masked by `mask` as in the reference, model_components.py:422) and the 32-row query / key tiles past lens[n] are neither
 * computed nor written.  lens == NULL: every sequence has L rows.  Pointers 16-byte aligned. */
int dldkd_attention_train_fwd_bf16io(const void* qkv, const float* mask, const int* lens, void* out, int N, int L, float p_drop,
                                     unsigned long long seed, unsigned long long offset, const unsigned long long* state,
                                     void* stream);
int dldkd_attention_train_bwd_bf16io(const void* qkv, const float* mask, const int* lens, const void* dout, void* dqkv, int N, int L,
                                     float p_drop, unsigned long long seed, unsigned long long offset,
                                     const unsigned long long* state, void* stream);

/* Fused training towers, throughput mode (tower_train.hip): everything of an encoder tower behind the input projection -
 * TrainablePositionalEncoding.forward (method/model_components.py:277-284), the q / k / v projections of BertSelfAttention
 * (:398-410), BertSelfOutput (:446-450) and out_mapping_linear (method/model.py:219) - as two row kernels forward (f1, f3) and two
 * backward (b3, b1) around the attention kernels above; activations cross HBM once, as bf16.  M = N L rows of the padded batch,
 * row = n L + l; flags (one byte per 32 consecutive rows, or NULL = all): groups flagged 0 hold no valid clip and are neither read
 * nor written (b1 writes zero rows for them).  Dropout: p_drop with the Philox (seed, offset, state) convention of
 * dldkd_dropout_fwd_f32 on the flat index of the (N, L, 384) tensor; the backward kernels recompute the masks from the same
 * arguments.  Weights reach the kernels in MFMA fragment order (dldkd_tower_train_pack; 288 KiB per 384 x 384 matrix):
 *   mode 0 / 1: W as the A operand of Y^T = W X^T, k natural / in the permuted order of an accumulator tile used as operand
 *   mode 2 / 3: W^T likewise (input gradients); up to three (384, 384) sources are concatenated along the OUTPUT axis (q | k | v).
 *   mode 4:     plain cast of host_src[3 j] to bf16, host_nsrc[j] = number of 8-element groups (the input projection's weight for
 *               dldkd_gemm_bf16_nt16, so that one launch per tower and step prepares every weight operand)
 * host_src (3 per job), host_nsrc, host_mode, host_out are HOST arrays of njobs <= 7 entries (the pointers in them are device
 * pointers); the call enqueues one kernel.
 *   f1: y0 (M, 384) fp32 (the input projection's output) + pos (L, 384) -> LayerNorm(gamma, beta) -> dropout -> h1d (M, 384) bf16,
 *       stats [2][M] (mean, rstd), qkv (M, 1152) bf16 = h1d Wqkv^T + (bq | bk | bv)          wqkv_pack: mode 0, three sources;
 *       xh1 (M, 384) bf16: the normalised rows; relu_bits (M, 48) bytes, 8-byte aligned: bit c % 8 of byte c / 8 of a row = [y0 > 0]
 *       (the input projection's ReLU mask in a plane of its own since round 6 - it used to take bit 0 of every xh1 element) - together
 *       all b1 needs of y0 and pos
 *   f3: ctx (M, 384) bf16 -> dense (wd_pack: mode 0) + bd -> dropout -> + h1d -> LayerNorm -> xh2 (normalised rows, bf16),
 *       rstd2 [M], and h2 = xh2 gamma + beta as bf16 (h2_bf16) followed by the out mapping g = h2 Wo^T + bo (fp32; wo_pack:
 *       mode 1) when wo_pack != NULL, as fp32 rows (h2_f32) otherwise (query towers: get_modularized_queries reads them)
 *   b3: dg (M, 384) fp32 = gradient of g (wot_pack: mode 2) or of h2 (wot_pack NULL) -> LayerNorm backward (dgamma, dbeta: [384],
 *       ADDED with atomics, zeroed by the caller) -> ddo = gradient of the dense output (bf16), dres = gradient reaching h1d through
 *       the residual (bf16), dctx = ddo Wd (bf16; wdt_pack: mode 3)
 *   b1: dqkv (M, 1152) bf16 -> dh1d = dqkv Wqkv + dres (wqkvt_pack: mode 2, three sources) -> dropout mask -> LayerNorm backward
 *       from xh1 and stats (dgamma, dbeta added) -> dx1 (M, 384) fp32 (gradient of y0 + pos; NULL: not written) and dy0 = dx1 (.) [y0 > 0] when
 *       relu_mask (the ReLU of LinearLayer, model_components.py:311; the mask = f1's relu_bits), else dy0 = dx1.
 * Weight gradients are dldkd_gemm_bf16_mixed(dw = 3 / 1) over the saved bf16 rows, bias gradients dldkd_colsum_bf16 /
 * dldkd_colsum_f32. */
/* "mixed" training precision (fp32-grade forward, bf16 backward): after a tower's forward pass on the fp32-grade kernels, ONE launch
 * writes from its fp32 intermediates the bf16 rows the fused backward kernels above read (what f1 / f3 save in throughput mode):
 * xh1 = ((y0 + pos) - mean1) rstd1, relu_bits = [y0 > 0] (f1's bit plane), h1d = bf16(h1), qkv16, ctx16, xh2 = ((dd + h1) - mean2) rstd2, rstd2,
 * h2_16 (video towers; h2 = h2_16 = NULL otherwise).  y0, h1 (the position LayerNorm's output behind its dropout), ctx, dd (the
 * dense layer's output behind its dropout), h2: (M, 384) fp32; qkv (M, 1152); stats1 / stats2 [2][M] = (mean, rstd) of the two
 * LayerNorms (dldkd_layernorm_ex_f32); pos (>= L, 384); flags: the tower's 32-row group flags or NULL (rows of groups flagged 0
 * are not written); h1d / ctx16 / h2_16 may be NULL (the caller already holds those rows: plane 0 of its two-plane GEMM operands).
 * Reference: model_components.py:277-284, 398-450. */
int dldkd_tower_train_emit(const float* y0, const float* pos, int L, const float* stats1, const float* h1, const float* qkv,
                           const float* ctx, const float* dd, const float* stats2, const float* h2, const unsigned char* flags, long M,
                           void* xh1, void* relu_bits, void* h1d, void* qkv16, void* ctx16, void* xh2, float* rstd2, void* h2_16, void* stream);
size_t dldkd_tower_train_pack_bytes(int n_mats);
int dldkd_tower_train_pack(const float* const* host_src, const int* host_nsrc, const int* host_mode, void* const* host_out, int njobs,
                           void* stream);
/* dldkd_tower_train_pack plus the sequence lengths of the tower's batch in the same launch: lens[n] = number of mask[n, :L]
 * entries > 0 (mask (n_seq, L) fp32, a prefix mask: data_provider.py:81-84), as dldkd_mask_lens_f32. */
int dldkd_tower_train_prepare(const float* const* host_src, const int* host_nsrc, const int* host_mode, void* const* host_out, int njobs,
                              const float* mask, int n_seq, int L, int32_t* lens, void* stream);
int dldkd_tower_train_f1(const float* y0, const float* pos, int L, const float* gamma, const float* beta, float eps, float p_drop,
                         unsigned long long seed, unsigned long long offset, const unsigned long long* state, const void* wqkv_pack,
                         const float* bq, const float* bk, const float* bv, const unsigned char* flags, long M, void* h1d, void* xh1,
                         float* stats, void* qkv, void* relu_bits, void* stream);
int dldkd_tower_train_f3(const void* ctx, const void* h1d, const void* wd_pack, const float* bd, float p_drop, unsigned long long seed,
                         unsigned long long offset, const unsigned long long* state, const float* gamma, const float* beta, float eps,
                         const void* wo_pack, const float* bo, const unsigned char* flags, long M, void* xh2, float* rstd2, void* h2_bf16,
                         float* h2_f32, float* g, void* stream);
int dldkd_tower_train_b3(const float* dg, const void* wot_pack, const void* xh2, const float* rstd2, const float* gamma, float p_drop,
                         unsigned long long seed, unsigned long long offset, const unsigned long long* state, const void* wdt_pack,
                         const unsigned char* flags, long M, void* ddo, void* dctx, void* dres, float* dgamma, float* dbeta, void* dg_bf16,
                         void* dh2_bf16, void* stream);
int dldkd_tower_train_b1(const void* dqkv, const void* dres, const void* wqkvt_pack, const void* xh1, const void* relu_bits, const float* stats,
                         const float* gamma, float p_drop, unsigned long long seed, unsigned long long offset,
                         const unsigned long long* state, const unsigned char* flags, long M, int relu_mask, float* dy0, float* dx1,
                         float* dgamma, float* dbeta, void* dz_bf16, void* dy_bf16, void* stream);
/* The LayerNorm parameter gradients of the two backward kernels: with dgamma / dbeta (zeroed accumulators) the kernels sum them
 * themselves (cross-lane reductions + atomics: a third of their time at the TVR batch); with dgamma = dbeta = NULL they leave the
 * gradient of that LayerNorm's output as bf16 rows instead - dh2_bf16 (M, 384; b3 under an out mapping: without one it is dg itself)
 * and dz_bf16 (M, 384; b1) - for dldkd_tower_train_dw_ln, which sums them beside the split-K reduce.  dg_bf16 (b3, out mapping only) and
 * dy_bf16 (b1): optional bf16 copies of dg / dy0 (M, 384) for the weight-gradient GEMMs, which contract bf16 rows
 * (dldkd_tower_train_dw block 0, dldkd_inproj_bwd_bf16).  Rows of skipped 32-row groups: zeros in dy_bf16 (as in dy0), not written in the other bf16 outputs. */
/* The weight and bias gradients of one fused training tower as ONE split-K product: dW (384 n_blocks, 384) fp32, block b =
 * sum over the batch rows r of A_b[r, acol_b .. acol_b + 384)^T B_b[r, :] (A_b: the gradient of layer b's output, fp32 or bf16
 * (a16_b) rows of lda_b elements; B_b: the layer's saved bf16 input rows (rows, 384)), dbias (384 n_blocks; NULL: not wanted;
 * zeroed by the caller) += the column sums of the A_b ranges.  k_flags as dldkd_gemm_bf16_mixed(dw = 1).  host_* are HOST arrays of
 * n_blocks <= 5 entries; workspace of dldkd_tower_train_dw_workspace_bytes(n_blocks, rows) bytes (NULL: no split-K). */
size_t dldkd_tower_train_dw_workspace_bytes(int n_blocks, long rows);
int dldkd_tower_train_dw(const void* const* host_A, const int* host_lda, const int* host_acol, const int* host_a16,
                         const void* const* host_B, int n_blocks, long rows, float* dW, float* dbias, void* workspace,
                         size_t workspace_bytes, const unsigned char* k_flags, void* stream);
/* The same plus the gradient of the position table (TrainablePositionalEncoding, model_components.py:277-284): dpos[c] += sum over
 * the n_seq sequences of dx1[n, c], c < cols = L * 384 (dx1 from dldkd_tower_train_b1; dpos zeroed by the caller) - inside the launch
 * that reduces the split-K planes (one launch fewer per tower than dldkd_tower_train_dw + dldkd_colsum_f32). */
int dldkd_tower_train_dw_pos(const void* const* host_A, const int* host_lda, const int* host_acol, const int* host_a16,
                             const void* const* host_B, int n_blocks, long rows, float* dW, float* dbias, void* workspace,
                             size_t workspace_bytes, const unsigned char* k_flags, const float* dx1, float* dpos, long n_seq, long cols,
                             void* stream);
/* dldkd_tower_train_dw_pos (dx1 / dpos may be NULL) plus the parameter gradients of the tower's two LayerNorms in the same finishing
 * launch: ln_grads (4, 384) fp32, ZEROED by the caller = [dgamma2 | dbeta2 | dgamma1 | dbeta1] with dgamma[c] += sum_r a[r, c] xh[r, c],
 * dbeta[c] += sum_r a[r, c] over the rows of flagged 32-row groups (k_flags) - LayerNorm 2: a = dh2 (dldkd_tower_train_b3's dh2_bf16, or
 * the fp32 rows dg it was given when there is no out mapping: dh2_is_bf16 = 0), xh = xh2; LayerNorm 1: a = dz1_bf16
 * (dldkd_tower_train_b1), xh = xh1.  All (rows, 384). */
int dldkd_tower_train_dw_ln(const void* const* host_A, const int* host_lda, const int* host_acol, const int* host_a16,
                            const void* const* host_B, int n_blocks, long rows, float* dW, float* dbias, void* workspace,
                            size_t workspace_bytes, const unsigned char* k_flags, const float* dx1, float* dpos, long n_seq, long cols,
                            const void* dz1_bf16, const void* xh1, const void* dh2, int dh2_is_bf16, const void* xh2, float* ln_grads,
                            void* stream);
/* out[c] += sum over the rows r of x[r, c0 + c] (x (M, ld) bf16, c < N) whose 32-row group is flagged (flags NULL: all rows); out is
 * zeroed by the caller.  The bias gradients of the fused training towers (rows of skipped groups are not written: they must not be
 * read). */
int dldkd_colsum_bf16(const void* x, int ld, int c0, int N, long M, const unsigned char* flags, float* out, void* stream);

/* Training-side simpool: for one (query set, gallery) pair of DLDKD.forward (model.py:113-129) everything the losses read
 * of get_sim_scores (model.py:307-329) and get_unnormalized_sim_scores (model.py:331-350), from ONE raw product
 * S[n, v, l] = <q_n, g_vl> that is never written to memory:
 *     pooled_raw[n, v] = max_{l < lens[v]} S             arg_raw = the clip          (model.py:344-349)
 *     pooled_cos[n, v] = max_{l < lens[v]} S rq_n rg_vl   arg_cos = the clip          (model.py:318-327)
 *     clip_pos[n, l]   = S[n, labels[n], l] rq_n rg_vl  (l < lens, else -1e10)  (nq, L): the [i, :, label_i] column that
 *                        compute_kl_loss(frame_score) reads (model.py:183-197); may be NULL
 * rq (nq) / rg (nv * L) = 1 / max(|row|, 1e-12) from dldkd_row_invnorm_f32 (F.normalize's clamp).  q (nq, D), g (nv, L, D)
 * fp32, L <= 128.  precision: DLDKD_GEMM_F32X3 (parity grade) or DLDKD_GEMM_BF16.  A video without valid clips pools to
 * -1e10 / clip 0 like the reference's masked maximum.
 * Backward: d_cos / d_raw (nq, nv) and d_clip (nq, L) (each may be NULL) -> dq (nq, D), dg (nv, L, D) (each may be NULL;
 * written, not accumulated): the max-pool gradient goes to the arg-max clip, the cosine through the normalisation Jacobian
 * rq (ghat - cos qhat); D a multiple of 4, <= 512. */
int dldkd_row_invnorm_f32(const float* x, float* inv, long M, int D, void* stream);
/* The same for the two operands of one scored pair (x0: M0 rows, x1: M1 rows, both D wide) in ONE launch. */
int dldkd_row_invnorm2_f32(const float* x0, float* inv0, long M0, const float* x1, float* inv1, long M1, int D, void* stream);
/* ... which also writes both operands as bf16 rows (round to nearest even; y0 (M0, D), y1 (M1, D)), and the pooled forward product
 * over those: the outputs of dldkd_simpool_train_fwd_f32(DLDKD_GEMM_BF16) from bf16 operands by LDS-DMA tiles (L <= 128, D % 64 ==
 * 0, 16-byte aligned operands; row tiles past a video's length are neither loaded nor multiplied).  The operands are rounded
 * exactly as the fp32-operand kernel rounds them on their way to LDS; the products differ in summation order only. */
int dldkd_row_invnorm2_cast_f32(const float* x0, float* inv0, void* y0_bf16, long M0, const float* x1, float* inv1, void* y1_bf16, long M1,
                                int D, void* stream);
int dldkd_simpool_train_fwd_bf16in(const void* q_bf16, const void* g_bf16, const float* rq, const float* rg, const int32_t* lens,
                                   const int32_t* labels, int nq, int nv, int L, int D, float* pooled_cos, float* pooled_raw,
                                   int32_t* arg_cos, int32_t* arg_raw, float* clip_pos, void* stream);
/* The same from TWO bf16 planes per operand (dldkd_row_invnorm2_planes_f32 writes them beside the norms: q_planes [2][nq][D], g_planes
 * [2][nv L][D]): the two-plane fp32-grade product - the pooled scores of the "mixed" training precision.  D % 64 == 0. */
int dldkd_row_invnorm2_planes_f32(const float* x0, float* inv0, void* y0_planes, long M0, const float* x1, float* inv1, void* y1_planes, long M1,
                                  int D, void* stream);
int dldkd_simpool_train_fwd_planes(const void* q_planes, const void* g_planes, const float* rq, const float* rg, const int32_t* lens,
                                   const int32_t* labels, int nq, int nv, int L, int D, float* pooled_cos, float* pooled_raw,
                                   int32_t* arg_cos, int32_t* arg_raw, float* clip_pos, void* stream);
int dldkd_simpool_train_fwd_f32(int precision, const float* q, const float* g, const float* rq, const float* rg,
                                const int32_t* lens, const int32_t* labels, int nq, int nv, int L, int D, float* pooled_cos,
                                float* pooled_raw, int32_t* arg_cos, int32_t* arg_raw, float* clip_pos, void* stream);
int dldkd_simpool_train_bwd_f32(const float* q, const float* g, const float* rq, const float* rg, const int32_t* lens,
                                const int32_t* labels, const int32_t* arg_cos, const int32_t* arg_raw, const float* pooled_cos,
                                const float* clip_pos, const float* d_cos, const float* d_raw, const float* d_clip, int nq, int nv,
                                int L, int D, float* dq, float* dg, void* stream);

/* One branch's loss terms of DLDKD.forward (method/model.py:137-155: get_clip_triplet_loss + 0.04 clip_nce_soft / clip_nce +
 * kl_intra_weight weight compute_kl_loss) with their gradients in three launches: [triplet t2v | triplet v2t | InfoNCE rows | KL] as
 * one grid, the InfoNCE column pass, the three sums.  Same arithmetic as dldkd_triplet_f32 / dldkd_nce_f32 / dldkd_kl_frame_f32.
 * C / S (nq, nv): pooled cosine / raw scores; T: soft-label scores (nq, nv), NULL with fold_t != 0 (the targets are S itself and
 * their gradient is added to dS: exploration branch) or for hard labels (hardQ = nq, hardV = nv, eps = 0); clip_p / clip_t (nq, L):
 * positive-column clip cosines of student / teacher (NULL: no KL term); cq / cv: the per-row / per-column coefficients of
 * dldkd_nce_f32.  terms: scratch of 2 (nq + nv) + nq floats; dC (nq, nv) and dclip (nq, L) zeroed by the caller, dS (nq, nv) written;
 * out[3] = triplet, w_nce InfoNCE, w_kl KL.  Gradients are those of out[] for an upstream gradient of 1;
 * dldkd_branch_losses_scale_f32 multiplies them by the actual upstream gradients (device scalars) in place.
 * nq_valid: rows [nq_valid, nq) of C / S / T / clip_p / clip_t belong to PADDING queries (a batch whose query axis was padded to a
 * bucket so that batches with different caption counts - method/data_provider.py:34-72, Charades / ActivityNet - share one captured
 * graph): the terms, normalisers and the columns' softmax run over the first nq_valid queries only, the padding rows' dS is written
 * as zero (dC / dclip stay as zeroed), labels / r_t2v need nq_valid valid entries; <= 0: nq.
 * sched: NULL, or 5 device words {hardQ, hardV, bits of beta, bits of w_kl, nq_valid} that REPLACE the by-value arguments of those
 * names when the kernels run - the scalars an epoch's schedule moves (method/train.py:66-113: alpha -> hardQ / hardV and cq / cv,
 * belta, the KD weight) and the batch's query count.  A launch captured into a hipGraph then follows them (the caller rewrites the
 * words and cq / cv in place). */
int dldkd_branch_losses_f32(const float* C, const float* S, const float* T, const float* clip_p, const float* clip_t,
                            const int32_t* labels, const int32_t* lens, const int32_t* r_t2v, const int32_t* r_v2t, const float* cq,
                            const float* cv, int nq, int nv, int L, int hard, int hardQ, int hardV, int fold_t, float margin, float beta,
                            float eps, float temp, float w_nce, float w_kl, float* terms, float* dC, float* dS, float* dclip, float* out,
                            int nq_valid, const int32_t* sched, void* stream);
int dldkd_branch_losses_scale_f32(float* dC, float* dS, long n, float* dclip, long n_clip, const float* g_trip, const float* g_nce,
                                  const float* g_kl, void* stream);
/* out[0] = sum of x[0..n) in a fixed order (single workgroup). */
int dldkd_sum_f32(const float* x, long n, float* out, void* stream);
/* out[0] = (((x0 + x1) + x2) + ...) of n <= 8 device scalars, left to right in fp32: the reference's sum of its loss terms
 * (model.py:157-160) as one launch.  host_ptrs: HOST array of n device pointers. */
int dldkd_sum_scalars_f32(const float* const* host_ptrs, int n, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Ranking (the step after scoring).  Replaces the np.argsort loop of eval_q2m (method/eval.py:69-83) and
 * the Python list walk of t2v_map (method/eval.py:97-111).
 * scores (nq, nv) fp32 similarity (higher = better; the reference ranks -scores ascending, eval.py:250);
 * GT videos of query q are gt_idx[gt_ptr[q] .. gt_ptr[q+1]) (CSR, built from get_gt, eval.py:43-57).
 * rank_best[q]  = 1 + #(scores[q,:] > best GT score)   -> gt_ranks of eval_q2m (min over GT videos);
 * rank_first[q] = 1 + #(scores[q,:] > first GT score)  -> AP = 1/rank for t2v_map (may be NULL).
 * Queries without GT get nv + 1 (eval.py:76).  NaN policy: "above" is evaluated as !(s <= gt), so NaN scores count as
 * above and a NaN ground-truth score ranks nv + 1 - a diverged model scores R@K = 0, not 100. */
int dldkd_rank_gt(const float* scores, int nq, int nv, const int32_t* gt_ptr, const int32_t* gt_idx,
                  int32_t* rank_best, int32_t* rank_first, void* stream);

/* Ranks of the ground-truth videos straight from the scorer's partial planes (the `workspace` of dldkd_simpool_eval_bf16),
 * for eval_epoch (eval.py:237-263), which ranks the inheritance, exploration and fused scores but never needs the (nq, nv)
 * matrices: replaces dldkd_simpool_finish + three dldkd_rank_gt passes by one read of the two planes.
 * counts (int32, [3 kinds][2][nq], zeroed here): kind 0 / 1 / 2 = branch 0 / branch 1 / fused (w0 s0 + w1 s1, the same
 * expression as dldkd_simpool_finish: bit-identical scores); [0] = #(scores above the BEST ground-truth video's score) ->
 * rank_best = 1 + count (eval_q2m, eval.py:69-83), [1] = the same for the FIRST listed ground-truth video (t2v_map,
 * eval.py:97-111).  "above" = !(s <= gt): the NaN policy of dldkd_rank_gt; a query without ground truth counts nv.
 * With n_branches == 1 all three kinds equal branch 0.  q_bad: NULL or dldkd_pack_queries_bf16's flags (flagged queries
 * count nv: they rank last).  thr_scratch: 6 * nq floats. */
int dldkd_simpool_rank_partials(const void* workspace, const int32_t* inv_order, int nq, int nv, int n_branches, float w0,
                                float w1, const int32_t* gt_ptr, const int32_t* gt_idx, const float* q_bad, float* thr_scratch,
                                int32_t* counts, void* stream);

/* The two halves of dldkd_simpool_rank_partials for a gallery SHARDED by video (eval_epoch_sharded: the loop of
 * method/eval.py:188-212 cut by video, ranked as method/eval.py:59-94 ranks).  Every rank holds the partial planes of ITS videos:
 *   _thr:   thresholds over the shard's own ground-truth videos.  gt_ptr / gt_idx: CSR of the LOCAL GT videos (local indices);
 *           first_local[q] != 0 iff the query's first listed GT video is local.  thr[3][2][nq] comes out NaN-free (-inf where
 *           the shard holds no GT video of the query) and nan_flag[3][2][nq] = 1 where the owner's first-GT score is NaN.
 *           The caller all-reduces both with MAX.
 *   _count: counts[3][2][nq] = # of the shard's videos with !(score <= thr) (zeroed here).  The caller all-reduces with SUM;
 *           rank = 1 + count, nv_total + 1 for a flagged query or one without ground truth.
 * No (nq, nv / N) score matrix is written on any rank. */
int dldkd_simpool_rank_partials_thr(const void* workspace, const int32_t* inv_order, int nq, int nv, int n_branches, float w0, float w1,
                                    const int32_t* gt_ptr, const int32_t* gt_idx, const int32_t* first_local, const float* q_bad,
                                    float* thr, float* nan_flag, void* stream);
int dldkd_simpool_rank_partials_count(const void* workspace, int nq, int nv, int n_branches, float w0, float w1, const float* thr,
                                      int32_t* counts, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Optimiser step and sharded-ranking helper.
 * ------------------------------------------------------------------------------------------- */

/* Zero n floats on `stream` the way the captured step's entry points zero their scratch (BertAdam's per-tensor norms): by a kernel
 * (default) or, after dldkd_set_zero_by_memset(1), by hipMemsetAsync = a MEMSET node under capture.  ROCm 7.0.2 replayed such a
 * node with every fourth word of a 296-byte buffer left stale when the stream was idle at launch; the host side captures this call
 * in a one-node graph and replays it over a poisoned buffer to learn what the runtime at hand does (staging.memset_node_defect). */
int dldkd_zero_scratch_f32(float* x, int n, void* stream);
/* 0 (default): small scratch buffers inside captured steps are zeroed by a kernel; 1: by hipMemsetAsync (a memset node).
 * Returns the previous setting.  The training stepper sets it from the probe. */
int dldkd_set_zero_by_memset(int on);

/* One BertAdam step (method/optimization.py:278-343) over ALL tensors of a flat parameter buffer in two
 * launches.  p/g/m/v: flat fp32 buffers; tensor t = [t_start[t], t_start[t]+t_numel[t]), t_start multiples of
 * 256; chunk_tensor[c] = tensor of 256-element chunk c.  Per-tensor clip to max_grad_norm (coef =
 * min(1, max/(norm+1e-6)), optimization.py:311-312), no bias correction, update += wd_t * p, p -= lr_t * update;
 * t_lr[t] = group lr x schedule multiplier (host).  norm2_scratch: n_tensors floats.  t_active (n_tensors floats, or
 * NULL = all): 0 marks a tensor whose gradient is None this step - it is skipped entirely (no moment decay, no weight
 * decay), as `if p.grad is None: continue` does (optimization.py:294-295). */
int dldkd_bert_adam_step_f32(float* p, const float* g, float* m, float* v, const int32_t* chunk_tensor, int n_chunks,
                             const int32_t* t_start, const int32_t* t_numel, int n_tensors, float* norm2_scratch,
                             const float* t_wd, const float* t_lr, const float* t_active, float b1, float b2, float eps,
                             float max_grad_norm, void* stream);
/* The step in two halves, for a caller whose gradients become final tower by tower (the graphed training step, one graph per tower):
 * dldkd_gather_sumsq_f32 copies n <= 32 gradient tensors (host arrays: device pointer, flat offset = t_start of the tensor, element
 * count, tensor index) into the flat gradient buffer and ADDS each one's sum of squares to norm2[tensor] (norm2 zeroed by the caller
 * once per step, dldkd_zero_scratch_f32; NULL: copy only; src may be the tensor's flat range itself) - one launch per tower, on that
 * tower's stream; dldkd_bert_adam_update_f32 is then the update alone, reading the finished norm2 (same arguments and arithmetic as
 * dldkd_bert_adam_step_f32, whose first two launches it leaves out). */
int dldkd_gather_sumsq_f32(const float* const* host_src, const int* host_start, const int* host_numel, const int* host_tensor, int n,
                           float* flat_grad, float* norm2, void* stream);
int dldkd_bert_adam_update_f32(float* p, const float* g, float* m, float* v, const int32_t* chunk_tensor, int n_chunks,
                               const int32_t* t_start, const int32_t* t_numel, int n_tensors, const float* norm2,
                               const float* t_wd, const float* t_lr, const float* t_active, float b1, float b2, float eps,
                               float max_grad_norm, void* stream);

/* counts[q] = #{v < nv : scores[q*ld + v] > thr[q]}: the local half of gather-free sharded ranking (each rank
 * counts the videos of its shard that beat the query's ground-truth score; the counts are all-reduced). */
int dldkd_count_above_f32(const float* scores, const float* thr, int nq, int nv, int ld, int32_t* counts, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K5: the whole tower behind the input projection as ONE kernel per (32-row tile slot, branch), throughput mode (MFMA on h16 =
 * fp16 operands, see the K4 section; the gallery rows it writes for the scorer are bf16):
 *   h1 = LayerNorm(h0 + pos)                          TrainablePositionalEncoding.forward  method/model_components.py:277-284
 *   ctx = softmax(q k^T / sqrt(96) + key mask) v      BertSelfAttention.forward            method/model_components.py:398-436
 *   h2 = LayerNorm(ctx Wd^T + bd + h1)                BertSelfOutput / BertAttention       method/model_components.py:446-450,345-353
 *   y = h2 Wo^T + bo                                  out_mapping_linear                   method/model.py:219
 *   out_mode 1: row = bf16(y / max(|y|, 1e-12))       F.normalize in get_sim_scores        method/model.py:319
 *   out_mode 2 (query towers): modular pooling on top       get_modularized_queries      method/model.py:245-258
 * dldkd_tower_blob_bytes / dldkd_tower_pack_h16: one branch's weights as h16 MFMA fragments in the kernel's consumption
 * order (k permuted to the accumulator layout, q pre-scaled by log2(e) / sqrt(96)) followed by its bias / gamma / beta vectors;
 * all weight matrices (384, 384) row-major, vectors (384).  A video tower passes wo / bo (out_mapping_linear) and mod_w = NULL;
 * a query tower passes wo = bo = NULL and mod_w = modular_vector_mapping.weight (384).
 *   pos (max_pos, 384) = position_embeddings.weight goes into the blob too, re-arranged per 32-position tile (positions past
 *   max_pos read as zeros).
 * dldkd_tower_seq_h16: h0 / blob / out_rows / gallery are HOST arrays of n_branches device pointers.
 *   h0[b] (rows, 384) fp32: the input projection's output; sequence s owns rows row0[s] .. row0[s] + lens[s] - 1
 *     (row0 == NULL: s * seq_rows); at most 128 rows per sequence.
 *   items (n_items, 4) int32 or NULL: the four 32-row slots of workgroup i: (s << 10) | (tile << 8) | lens[s], -1 = idle; the tiles of one
 *     sequence occupy consecutive slots of ONE workgroup in order (short sequences share a workgroup); every scheduled sequence
 *     has lens > 0.  NULL: workgroup i is sequence i (n_items = n_seq; out_mode 2: sequences 4 i .. 4 i + 3, n_items = ceil(n_seq / 4)).
 *   out_mode 0 (video-tower blobs): out_rows[b] (rows, 384) fp32 indexed like h0.  Rows lens[s] .. seq_rows - 1 of a sequence:
 *     without an item table they are computed as the reference computes the clips past a video's length (queries like any other,
 *     only keys are masked: method/model_components.py:422; don't-care values, but the same ones); with an item table zeros.
 *   out_mode 1 (video-tower blobs): gallery[b] = the scorer's bf16 blob [nv_total][Lp][384] (dldkd_pack_gallery_bf16's layout:
 *     rows past the length inside the last 16-row tile replicate the last clip, further rows zero); sequence s is video v0 + s;
 *     lens_out (whole gallery, or NULL) receives lens.  Lp a multiple of 32.
 *   out_mode 2 (query-tower blobs): out_rows[b] (n_seq, 384) fp32 = the modular query vectors; 1 <= lens[s] <= 32 (a sequence
 *     with lens < 1 gives a zero vector).
 *   nonfinite_flag (one device int32 the caller zeroed, or NULL; out_mode 0 / 1 - a non-finite query vector of out_mode 2 is
 *     flagged by dldkd_pack_queries_bf16's bad_flags): set to 1 when the second LayerNorm of a VALID row meets a mean that is not finite.  The operands of this kernel and of K4 / K4b are IEEE fp16 (65,504 max): an activation of an
 *     arbitrary checkpoint that overflows it (h0 = ReLU(W LN(x) + b), q | k | v, the context - none of them LayerNorm outputs)
 *     turns into Inf, then NaN, and every such row passes through this LayerNorm; the host reads the flag with the epoch's results
 *     and re-runs in parity mode (dldkd_amd.eval) instead of ranking NaN scores last in silence.  One compare per 32-row tile. */
size_t dldkd_tower_blob_bytes(int with_out_map);
int dldkd_tower_pack_h16(const float* ln1_g, const float* ln1_b, const float* wq, const float* bq, const float* wk, const float* bk,
                          const float* wv, const float* bv, const float* wd, const float* bd, const float* ln2_g, const float* ln2_b,
                          const float* wo, const float* bo, const float* mod_w, const float* pos, int max_pos, void* blob, void* stream);
int dldkd_tower_seq_h16(const float* const* h0, const void* const* blob, const int32_t* row0,
                         const int32_t* lens, const int32_t* items, int n_items, int n_seq, int n_branches,
                         int out_mode, float* const* out_rows, int seq_rows, void* const* gallery, int v0, int Lp, int32_t* lens_out,
                         int32_t* nonfinite_flag, void* stream);
/* out_mode 1 from h16 h0 rows (dldkd_in_proj_h16_rows128b_out16; ragged: row0 required): the prologue is one round of
 * 16-byte loads straight into the operand registers (no LDS staging); everything else as above.  skip_zero_rows != 0: the rows
 * >= ceil(lens / 16) * 16 of a video - which the scorers never load - are not written: the caller's gallery buffer holds zeros there
 * already (zero-filled once; a dataset's lengths do not change between epochs). */
int dldkd_tower_seq_h16_rows16(const void* const* h0_h16, const void* const* blob, const int32_t* row0,
                             const int32_t* lens, const int32_t* items, int n_items, int n_seq, int n_branches,
                             void* const* gallery, int v0, int Lp, int32_t* lens_out, int32_t* nonfinite_flag, int skip_zero_rows,
                             void* stream);

/* Diagnostics: the out_mode 1 kernel (two branches) with clock stamps at its phase boundaries; stamps = 24 x uint64 per workgroup
 * (h16 != 0: h0 holds h16 rows, the dldkd_tower_seq_h16_rows16 kernel)
 * (8 * ceil(n_items / 4) workgroups): [0] start, [1] prologue, [2 + 2 h] head h projected, [3 + 2 h] head h attended, [10] dense,
 * [11] LayerNorm, [12] out mapping, [13] rows stored, [16..21] inside the prologue (tools/tower_timeline.py). */
int dldkd_debug_tower_seq_timeline(const float* const* h0, const void* const* blob, const int32_t* lens,
                                   const int32_t* items, int n_items, int n_seq, int seq_rows, void* const* gallery, int Lp,
                                   unsigned long long* stamps, int h16, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Feature ingest: temporal down-sampling + L2 normalisation + padding of raw frame features on the GPU.
 * Replaces uniform_feature_sampling / l2_normalize_np_array / cat_videos (method/data_provider.py:52-86).
 * frames (n_frames, D) fp32; output row r (n_rows = batch * Lmax, D): padding when seg_start[r] < 0, else the
 * mean of frames [seg_start[r], seg_end[r]) (the single frame seg_start[r] when the range is empty), divided by
 * (its L2 norm + eps) - eps is ADDED to the norm (data_provider.py:71-73). */
int dldkd_segment_mean_l2norm_f32(const float* frames, const int32_t* seg_start, const int32_t* seg_end, float* out,
                                  long n_rows, int D, float eps, void* stream);

/* Batch assembly from device-resident ragged feature tables (dldkd_amd.data.DeviceTrainSet): replaces the host-side padding of
 * collate_train / collate_frame_val (method/data_provider.py:75-86,111-136) and the H2D copy of the padded batch.
 * src (total_rows, D) fp32; item i owns rows [row_start[i], row_start[i] + lens[i]); out (n_items, Lmax, D):
 * out[b, l] = src row row_start[items[b]] + l for l < lens[items[b]], zeros beyond; mask (n_items, Lmax) = 1 / 0 or NULL. */
int dldkd_gather_pad_rows_f32(const float* src, const long long* row_start, const int32_t* lens, const int32_t* items, int n_items,
                              int Lmax, int D, float* out, float* mask, void* stream);

/* Raw feature rows -> their device-resident form (dldkd_in_proj_h16_rows128b): h16 = fp16 (round to nearest even) + fp32 LayerNorm
 * statistics of the fp32 values (mean, rstd = 1 / sqrt(biased variance + eps): nn.LayerNorm, method/model_components.py:297).
 * src (n_items, L, K) fp32, a padded batch as collate_frame_val builds it (method/data_provider.py:111-136); row l of item b goes
 * to table row dst_row0[b] + l when l < lens[b] (lens == NULL: every row; dst_row0 == NULL: row b L + l): the padding is dropped,
 * the table is ragged.  x_h16 (rows, K) 8-byte aligned, mean / rstd (rows). */
int dldkd_rows_to_h16_stats(const float* src, const int32_t* lens, const long long* dst_row0, int n_items, int L, int K, float eps,
                             void* x_h16, float* mean, float* rstd, void* stream);

/* Upload of a small host-produced int32 table (the slot and row-group tables of dldkd_tower_seq_h16 /
 * dldkd_in_proj_h16_rows128_groups; nothing in the reference) by a kernel: pinned_src is page-locked, device-mapped host
 * memory (hipHostMalloc / torch pin_memory), read over the bus on the compute queue - no copy-engine hand-off.  The caller keeps
 * pinned_src unchanged until the launch has executed (an event, as for hipMemcpyAsync). */
int dldkd_upload_words(const int32_t* pinned_src, int32_t* dst, long n_words, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Collectives: the two exchanges of the one-process-per-GPU form of the path (nothing in the single-GPU reference; they stand
 * where torch.nn.parallel would put them): the all-gather of the (Nq, Nv / world) score blocks of the gallery sharded by video
 * (the loop of method/eval.py:188-212) and the all-reduce of the flat gradient buffer between loss.backward() and
 * optimizer.step() (method/train.py:147-151).  RCCL over xGMI, driven directly (librccl.so.1 is resolved at the first call; the
 * copy already mapped by the host process is used when there is one).
 *   - a communicator is an explicit HOST object: created by dldkd_comm_init on the calling thread's current device (blocking:
 *     RCCL's rendezvous; the 128-byte id comes from rank 0's dldkd_comm_unique_id and travels to the other ranks by whatever
 *     the host has - dldkd_amd.comm uses the TCP store of torch.distributed's env:// rendezvous), destroyed by
 *     dldkd_comm_destroy.  These three and dldkd_comm_abort are the only entry points of the library that do more than enqueue.
 *   - every collective is ONE enqueue on `stream` and returns at once; completion is ordinary stream order.  No helper thread,
 *     no event polling: the calls may be issued between hipGraph replays or inside a capture, as RCCL allows.  All ranks must issue
 *     the same collectives in the same order; buffers are device pointers; in-place (send == recv, or recv + rank * send_count
 *     for the gather) is allowed.
 * ------------------------------------------------------------------------------------------- */
#define DLDKD_COMM_ID_BYTES 128
#define DLDKD_F32 0
#define DLDKD_F64 1
#define DLDKD_I32 2
#define DLDKD_I64 3
#define DLDKD_U8 4
#define DLDKD_SUM 0
#define DLDKD_MAX 1
#define DLDKD_MIN 2
/* RCCL's version code (e.g. 22606), or DLDKD_ECOMM when librccl.so.1 cannot be loaded. */
int dldkd_comm_rccl_version(void);
/* Rank 0: fill host_id_out (DLDKD_COMM_ID_BYTES bytes of host memory) with a fresh rendezvous id. */
int dldkd_comm_unique_id(void* host_id_out);
/* Every rank, with the same id: *host_comm_out = the communicator of `world` ranks on the current device. */
int dldkd_comm_init(void** host_comm_out, int world, int rank, const void* host_id);
int dldkd_comm_info(void* comm, int* host_world_out, int* host_rank_out);
/* Destroy after the streams that carry its collectives have drained (the caller synchronises; this call does not). */
int dldkd_comm_destroy(void* comm);
int dldkd_comm_abort(void* comm);
/* DLDKD_OK, or DLDKD_ECOMM with the text in dldkd_last_error() if a collective failed asynchronously (a peer died). */
int dldkd_comm_async_error(void* comm);
/* recv[i] = op over ranks of send[i], i < count (the gradient mean is SUM followed by the caller's 1 / world). */
int dldkd_comm_all_reduce(void* comm, const void* send, void* recv, size_t count, int dtype, int op, void* stream);
/* recv[r * send_count + i] = rank r's send[i]: the score blocks of one query range, rank-major. */
int dldkd_comm_all_gather(void* comm, const void* send, void* recv, size_t send_count, int dtype, void* stream);
/* buf on every rank = buf of rank `root` (the replicas' initial parameters, method/train.py:186-201 has one replica). */
int dldkd_comm_broadcast(void* comm, void* buf, size_t count, int dtype, int root, void* stream);
/* Several collectives as one RCCL group (one launch for a run of small ones). */
int dldkd_comm_group_begin(void);
int dldkd_comm_group_end(void);

#ifdef __cplusplus
}
#endif
#endif /* DLDKD_HIP_H */
