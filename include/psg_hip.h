/*
 * psg_hip.h - C ABI of libpsg_hip.so: MI355X (gfx950) kernels for OpenPSG's pairwise
 * relation-query + LMM relation-decode hot path.
 *
 * This is the drop-in boundary.  The reference has no native code: every entry
 * point below replaces arithmetic that the reference runs through torch / HF transformers /
 * timm inside `RelationTransformerHeadV4.forward`
 * (V4 = kings_sgg/models/relation_heads/relation_transformer_head_v4.py; HF-IB / HF-LL = the
 * un-vendored transformers InstructBLIP Q-Former / Llama modules it instantiates at V4:78-84, 99-100).
 * The host side that calls this ABI is openpsg_amd/ (ctypes; INTEGRATION.md shows the binding).
 *
 * Conventions
 *  - every function returns 0 on success, a negative psg_status otherwise; psg_last_error()
 *    returns a thread-local message for the last failure on the calling thread;
 *  - tensor arguments are RAW DEVICE POINTERS with explicit sizes; memory is allocated and owned
 *    by the caller (torch); the library never allocates, frees or retains caller memory;
 *  - every launch takes the hipStream_t to enqueue on (void* stream; pass
 *    torch.cuda.current_stream().cuda_stream); no hidden synchronisation, so calls are HIP-graph
 *    capturable;
 *  - `dtype` selects the ACTIVATION storage type (PSG_F32 verification mode / PSG_BF16 / PSG_F16: the matrix-core
 *    kernels exist for both 16-bit types, with the matching MFMA opcodes);
 *    reductions, softmax and normalisation statistics are always fp32; LayerNorm/RMSNorm
 *    parameters, biases, embedding tables and the existence-head weights are fp32;
 *  - a psg_ctx is used by one host thread at a time; different contexts are independent.
 */
#ifndef PSG_HIP_H
#define PSG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct psg_ctx psg_ctx;

enum psg_status {
  PSG_OK = 0,
  PSG_ERR_INVALID = -1,      /* bad argument (null pointer, unsupported size) */
  PSG_ERR_UNSUPPORTED = -2,  /* shape outside what the kernels are specialised for */
  PSG_ERR_HIP = -3,          /* HIP runtime error (message in psg_last_error) */
  PSG_ERR_NO_DEVICE = -4
};

enum psg_dtype { PSG_F32 = 0, PSG_BF16 = 1, PSG_F16 = 2 };

/* cross-attention empty-pair-mask policy: additive finfo.min => uniform softmax */
enum psg_empty_policy { PSG_EMPTY_UNIFORM = 0, PSG_EMPTY_UNMASKED = 1 };

/* cross-attention implementation: matrix-core kernels (default: the LDS-DMA kernel when its LDS image fits, else
 * the first-generation kernel), the scalar fp32 checker kernel, or the first-generation matrix-core kernel */
enum psg_xattn_variant { PSG_XATTN_MFMA = 0, PSG_XATTN_SIMPLE = 1, PSG_XATTN_MFMA_V1 = 2 };

/* ABI version: bumped whenever an entry point's argument list changes.  psg_version() returns the version the
 * library was built with; a binding compares it with the PSG_ABI_VERSION of the header it was written against
 * (openpsg_amd/_lib.py does, at load time) instead of passing shifted arguments silently.
 *   100  round 1        200  round 2 (psg_skinny_gemm / psg_rope_kvwrite gained, psg_qformer_cross_attn lost an argument)
 *   300  round 3 (psg_rmsnorm: resid_dtype; psg_greedy_step: embedding row of the chosen token;
 *        psg_train_* gradient kernels, psg_add_layernorm_res32, psg_gather_pair_rows,
 *        psg_masked_split_mean_pool added)
 *   400  round 4 (psg_skinny_gemm_plan: dtype; psg_skinny_gemm accepts PSG_F32 = the reference's own precision;
 *        psg_train_attn_fwd / _bwd: attention-probability dropout mask; psg_split_f16x3, psg_scale_rows_cols added)
 *   401  round 4 (psg_dense_gemm_tiled, psg_interleave_gate_up added)
 *   500  round 5 (psg_decode_layer*: one persistent launch per decoder layer of the decode step; psg_qformer_cross_attn /
 *        psg_qformer_self_attn(_shared) / psg_prefill_attn accept PSG_F32 on the matrix cores)
 *   501  round 5 (psg_batch_gemm*, psg_qformer_cross_attn_indexed, psg_skinny_gemm_w16, psg_split_f16x2, psg_split_gemm_w16,
 *        psg_rmsnorm_split2, psg_rmsnorm_split / psg_rope_kvwrite_scaled /
 *        psg_silu_mul_split added)
 *   600  round 6 (psg_dense_gemm_split added; psg_split_f16x3 order 2) */
#define PSG_ABI_VERSION 600
int psg_version(void);
const char* psg_last_error(void);
int psg_create(int device, psg_ctx** out);
int psg_destroy(psg_ctx* ctx);
/* number of compute units / arch name of the context's device (diagnostics, roofline) */
int psg_device_info(psg_ctx* ctx, int* num_cu, char* arch, int arch_len);

/* Tunables of a context (kernel variants, split planner; names in openpsg_amd/csrc/psg_common.h `psg_opts`).
 * Defaults are the measured-best settings; psg_create also reads PSG_<NAME> from the environment once per
 * context.  The library keeps no process-global mutable state: every launch consults only its psg_ctx. */
int psg_set_option(psg_ctx* ctx, const char* name, int value);
int psg_get_option(psg_ctx* ctx, const char* name, int* value);

/* Debugging aid: per-wave cycle-counter stamps of the next launches of one kernel family are written to a
 * CALLER-PROVIDED device buffer (the library does not allocate, copy or synchronise; a buffer too small for a
 * launch is simply not written).  PSG_TRACE_SKINNY_GEMM: 8 int64 per wave; PSG_TRACE_CROSS_ATTN: 32 per wave;
 * PSG_TRACE_DECODE_LAYER: 24 per workgroup of psg_decode_layer (100 MHz wall-clock stamps at its phase boundaries). */
enum psg_trace_kind { PSG_TRACE_NONE = 0, PSG_TRACE_SKINNY_GEMM = 1, PSG_TRACE_CROSS_ATTN = 2, PSG_TRACE_DECODE_LAYER = 3 };
int psg_set_trace_buffer(psg_ctx* ctx, int kind, void* device_buffer, int64_t bytes);

/* ---- A4 / K1: patch embedding, V4:410 (timm PatchEmbed = Conv2d(C, Cout, 16, 16) + flatten(2).transpose(1,2)):
 * out[l][o] = bias[o] + sum_{c,dy,dx} feat[c][16 py+dy][16 px+dx] * weight[o][c][dy][dx], l = py*(Wf/16)+px.
 * Exact fp32 (f32 matrix cores), split-K over the chip with a deterministic second-pass reduction.
 * feat [C][Hf][Wf] fp32, weight [Cout][C*256] fp32, out [L][Cout] fp32; the caller provides the
 * workspace (size from psg_patch_embed_workspace). */
int psg_patch_embed_workspace(psg_ctx*, int C, int Hf, int Wf, int Cout, int patch, int64_t* bytes);
int psg_patch_embed(psg_ctx*, const float* feat, int C, int Hf, int Wf, const float* weight,
                    const float* bias, int Cout, int patch, float* out, float* workspace,
                    int64_t workspace_bytes, void* stream);

/* ---- A4 / K2: panoptic id map -> patch grid.  Replaces V4:416-423
 * (F.interpolate nearest -> F.pad zero -> F.interpolate nearest).  grid[gh*gw] float32 ids. */
int psg_mask_grid(psg_ctx*, const int32_t* pan, int H0, int W0, int img_h, int img_w,
                  int pad_h, int pad_w, int gh, int gw, float* grid, void* stream);

/* ---- A4 / K3: per-object patch bitmasks.  Replaces V4:425-433: the N^2 pair masks are never
 * materialised; pair (i,j) uses bits[i] | bits[j] inside the attention kernel.
 * bits[N][words] uint64, bit l of object n = (grid[l] == object_ids[n]); words >= ceil(L/64). */
int psg_object_bitmasks(psg_ctx*, const float* grid, int L, const int32_t* object_ids, int N,
                        uint64_t* bits, int words, void* stream);

/* ---- K4: Q-Former embeddings, HF-IB:728-757 (called from V4:179-185).
 * out rows [0, B*nq): LayerNorm(query_rows[r]) for every pair; rows [B*nq, B*(nq+T)):
 * LayerNorm(word_emb[ids[p][t]] + pos_emb[t]).  ids int32 [B][T]. */
int psg_qformer_embed(psg_ctx*, const int32_t* ids, int B, int T, const float* word_emb,
                      const float* pos_emb, const float* query_rows, int nq, const float* ln_w,
                      const float* ln_b, float eps, int hidden, void* out, int dtype, void* stream);

/* ---- BertSelfOutput / BertOutput tail, HF-IB:519-530, 585-596:
 * out = LayerNorm(x + bias + residual) * gamma + beta; bias / residual may be NULL; out may alias x. */
int psg_add_layernorm(psg_ctx*, const void* x, const void* residual, const float* bias,
                      const float* gamma, const float* beta, float eps, int64_t rows, int hidden,
                      void* out, int dtype, void* stream);
/* residual row of output row r = residual_table[r % table_rows] (layer 0: the embedded query rows are one
 * [33][hidden] block shared by all pairs, HF-IB:728-757 + 519-527). */
int psg_add_layernorm_periodic(psg_ctx*, const void* x, const void* residual_table, int table_rows,
                               const float* bias, const float* gamma, const float* beta, float eps,
                               int64_t rows, int hidden, void* out, int dtype, void* stream);
/* residual row of output row r = residual_table[block_index[r / group] * group + r % group]: rows in groups (the 33
 * query rows of a pair), several groups sharing one block of the table (the rows of the pair's PROMPT: layer 0's
 * self-attention output depends on the prompt only, so it is computed per distinct prompt, HF-IB:519-530). */
int psg_add_layernorm_indexed(psg_ctx*, const void* x, const void* residual_table, const int32_t* block_index, int group,
                              const float* bias, const float* gamma, const float* beta, float eps, int64_t rows,
                              int hidden, void* out, int dtype, void* stream);
/* mixed mode (16-bit GEMM operands, fp32 residual stream): x = 16-bit projection output, residual = fp32 (plain:
 * res_period 0; periodic: res_period = table rows; indexed: res_period = group, res_index = block per group); the result is
 * written as fp32 (out32: the next residual, never rounded to 16 bits) and / or 16-bit (out16: the next operand). */
int psg_add_layernorm_res32(psg_ctx*, const void* x, const float* residual, int res_period, const int32_t* res_index,
                            const float* bias, const float* gamma, const float* beta, float eps, int64_t rows,
                            int hidden, void* out16, float* out32, int dtype, void* stream);

/* ---- BertIntermediate activation, HF-IB:563-577: out = gelu_erf(x + bias); bias may be NULL. */
int psg_bias_gelu(psg_ctx*, const void* x, const float* bias, int64_t rows, int cols, void* out,
                  int dtype, void* stream);

/* ---- K5: Q-Former self-attention, HF-IB:471-515 (eager 176-196).
 * qkv [(B*nq + B*T)][3*hidden] = [Q|K|V] of the query rows then the text rows; text_mask uint8
 * [B][T] (V4:158-159; masked keys get additive finfo.min).  nq + T <= 64, head_dim 64.
 * query_rows_only != 0 computes only the nq query rows of each pair (last layer, V4:185). */
int psg_qformer_self_attn(psg_ctx*, const void* qkv, const uint8_t* text_mask, int B, int T, int nq,
                          int heads, int query_rows_only, void* out, int dtype, void* stream);

/* First-layer variant (bf16): the nq query rows entering layer 0 are the same for every pair (learned query
 * tokens through the embedding LayerNorm, HF-IB:728-757), so their fused Q/K/V projection qkv_query
 * [nq][3*hidden] is computed once; qkv_text [B*T][3*hidden] holds the text rows.  out as above. */
int psg_qformer_self_attn_shared(psg_ctx*, const void* qkv_query, const void* qkv_text,
                                 const uint8_t* text_mask, int B, int T, int nq, int heads, void* out,
                                 int dtype, void* stream);
/* Last layer, selection phase: attention of the cls row (row 0) of every pair over the pair's nq + T keys - all that the
 * existence head (V4:206-209) and therefore the selector (V4:235-237) can observe of the last layer's query rows 0.
 * q_cls [B][hidden]: projected queries of the cls rows; kv [B*(nq+T)][2*hidden]: K | V projections of every row (rows
 * ordered as in psg_qformer_self_attn; the queries of rows 1..32 are never projected); out [B][hidden] COMPACT.
 * Rows 1..32 are then computed for the selected pairs only (psg_qformer_self_attn on the gathered pairs): results
 * identical, 97 % of the last layer's query-row work gone. */
int psg_qformer_self_attn_cls(psg_ctx*, const void* q_cls, const void* kv, const uint8_t* text_mask, int B, int T,
                              int nq, int heads, void* out, int dtype, void* stream);
/* The same cls-row attention with the key / value projections (HF-IB:471-475) folded into the input space, so that no
 * K | V tensor exists: score_j = (W_k,h^T q_h) . x_j / 8 (+ a per-head constant that the softmax drops) and
 * context_h = W_v,h (sum_j p_j x_j) + b_v,h.  x_query [B*nq][hidden]: the layer's INPUT query rows, pair-major;
 * x_text: its text rows in blocks of T - block text_index[p] for pair p, or block p when text_index is NULL; pairs with
 * the same prompt may share one block and one row of text_mask (a text row entering the last layer depends on the prompt
 * only); g [heads][B][hidden] FP32 = W_k,h^T q_h, projected by the caller; xbar [heads][B][hidden] FP32 =
 * sum_j p_j x_j, which the caller projects through W_v,h (+ b_v) - fp32 on both sides because a 768-term product of
 * rounded factors would lose what the 64-term q.k keeps.  Same function as psg_qformer_self_attn_cls up to rounding
 * (fp32: 1e-6); the K | V projection of every row of every pair is not computed at all.
 * hidden 768 = 12 heads only; PSG_ERR_UNSUPPORTED when a pair's rows exceed the LDS (fp32, > 48 rows). */
int psg_qformer_cls_attn_input(psg_ctx*, const void* x_query, const void* x_text, const int32_t* text_index,
                               const void* g, const uint8_t* text_mask, int B, int T, int nq, int heads, int hidden,
                               void* xbar, int dtype, void* stream);

/* ---- K6: relation-query cross-attention (primary kernel), HF-IB:464-466, 487-496 with the
 * V4:168-170 expand removed: K/V [L][hidden] are projected ONCE per image and shared by every
 * pair; the pair mask is bits[i] | bits[j] (pair_index[p] = i*N + j) applied on the fly.
 * q / out [P*nq][hidden]; scores = q.k/sqrt(64) + mask; fp32 softmax; all-masked => uniform.
 */
int psg_qformer_cross_attn(psg_ctx*, const void* q, const void* k, const void* v,
                           const uint64_t* bits, int words, const int32_t* pair_index, int N, int P,
                           int L, int nq, int heads, int empty_policy, int variant, void* out,
                           int dtype, void* stream);
/* The same with the queries stored once per PROMPT (the prompt-deduplicated layer 0): q_u [U][33][hidden], q_index [P] =
 * prompt of each pair, q_cls [P][hidden] = the cls row of every pair.  LDS-DMA kernel only: PSG_ERR_UNSUPPORTED otherwise
 * (the caller expands q and uses psg_qformer_cross_attn). */
int psg_qformer_cross_attn_indexed(psg_ctx*, const void* q_u, const int32_t* q_index, const void* q_cls, const void* k,
                                   const void* v, const uint64_t* bits, int words, const int32_t* pair_index, int N, int P,
                                   int L, int heads, int empty_policy, void* out, int dtype, void* stream);

/* ---- K8: pair-existence scoring head, V4:206-209: logit = w . x[p*nq] + b, prob = sigmoid. */
int psg_exist_head(psg_ctx*, const void* x, const float* w, const float* b, int P, int nq, int hidden,
                   float* logit, float* prob, int dtype, void* stream);

/* ---- K9: selector, V4:235-237: indices of the k largest scores, descending, ties -> lower index. */
int psg_topk(psg_ctx*, const float* score, int n, int k, int32_t* out_idx, float* out_val, void* stream);

/* ---- row gather (pair_feature[selected], embed_tokens[ids]; V4:294-297): dst[r] = src[idx[r]];
 * idx < 0 writes zeros.  src_dtype/dst_dtype allow fp32 tables -> bf16 activations. */
int psg_gather_rows(psg_ctx*, const void* src, int src_dtype, const int32_t* idx, int64_t n, int cols,
                    int64_t src_row_stride, void* dst, int dst_dtype, int64_t dst_row_stride, void* stream);
/* rows of the SELECTED pairs for the second phase of the last Q-Former layer (V4:215, 235-237), in one launch: sel[s] is
 * a global pair id; a pair of this chunk (first <= id < first + count) sits at position id - first + slot_off of the pass,
 * any other slot is computed as the chunk's first pair and flagged mine_out[s] = 0.  out [K*(nq+T)][cols]: the K*nq query
 * rows (xq[pos*nq + q]) then the K*T text rows (block text_index[pos], or pos, of xt); mask_out [K][T] = text_mask[pos];
 * pair_out [K] = pair_index[pos].  Optional outputs may be NULL. */
int psg_gather_pair_rows(psg_ctx*, const void* xq, const void* xt, const int32_t* text_index, const uint8_t* text_mask,
                         const int32_t* pair_index, const int32_t* sel, int K, int first, int count, int slot_off,
                         int nq, int T, int cols, void* out, uint8_t* mask_out, int32_t* pair_out, uint8_t* mine_out,
                         int dtype, void* stream);

/* ---- K12: Llama RMSNorm, HF-LL:53-67, fused with the residual add of HF-LL decoder layer:
 * if delta != NULL: resid += delta (written back); out = w * (resid * rsqrt(mean(resid^2)+eps)).
 * delta_splits > 0: delta is fp32 split-K partials [delta_splits][rows][hidden] of psg_skinny_gemm
 * (summed here, rounded once to the activation dtype); 0: delta is an activation-dtype tensor.
 * resid_dtype: storage type of the residual stream `resid` - `dtype` (what HF keeps for a model cast to 16 bits) or
 * PSG_F32 with a 16-bit `dtype` (mixed mode: 16-bit GEMM operands `out`, residual stream never rounded to 16 bits). */
int psg_rmsnorm(psg_ctx*, void* resid, const void* delta, int delta_splits, const float* w, float eps,
                int64_t rows, int hidden, void* out, int dtype, int resid_dtype, void* stream);

/* ---- K13: rotary embedding (half-split, HF-LL:130-160) + KV-cache write.
 * qkv [rows][3*hidden]; tok_pair / tok_pos int32 [rows] give the cache row (pair) and the
 * position (= cache slot = cumsum(mask)-1, V4 left-padding removed by compaction); tok_pos < 0
 * marks a padding row (skipped).  q_out [rows][hidden]; caches [pairs][heads][ctx][head_dim].
 * rope_cos / rope_sin: fp32 tables [ctx][head_dim/2] = cos/sin(position * inv_freq), HF-LL:115-128.
 * qkv_splits > 0: qkv is fp32 split-K partials [qkv_splits][rows][3*hidden].
 * rope_pos (may be NULL = tok_pos): rotary position per row when it differs from the cache slot - the TRAINING
 * forward numbers positions over the padded sequence (plain HF forward, V4:327-330), the cache stays compact. */
int psg_rope_kvwrite(psg_ctx*, const void* qkv, int qkv_splits, const int32_t* tok_pair, const int32_t* tok_pos,
                     const int32_t* rope_pos, const float* rope_cos, const float* rope_sin, int64_t rows, int heads,
                     int head_dim, int ctx, void* q_out, void* k_cache, void* v_cache, int dtype, void* stream);

/* ---- K14: Llama attention over the KV cache (prefill and decode), HF-LL:191-214: query at
 * (pair, pos) attends cache slots [0, pos]; fp32 softmax; out [rows][hidden]. head_dim 128. */
int psg_llm_attn(psg_ctx*, const void* q, const void* k_cache, const void* v_cache,
                 const int32_t* tok_pair, const int32_t* tok_pos, int64_t rows, int heads,
                 int head_dim, int ctx, void* out, int dtype, void* stream);

/* ---- K14 (prefill) on the matrix cores, bf16: the same attention for a pair-major prompt batch.
 * Pair p owns rows [p*rows_per_pair, (p+1)*rows_per_pair) of q / out and cache rows 0.. of its own
 * (pair, head) slab; tok_pos[row] is the token's position, equal to its row index inside the pair, or
 * -1 for the padding rows at the end of a pair (output rows of zeros).  rows_per_pair <= 64;
 * longer prompts and fp32 go through psg_llm_attn. */
int psg_prefill_attn(psg_ctx*, const void* q, const void* k_cache, const void* v_cache,
                     const int32_t* tok_pos, int pairs, int rows_per_pair, int heads, int head_dim,
                     int ctx, void* out, int dtype, void* stream);

/* ---- K13 + K14 (prefill) fused: psg_rope_kvwrite + psg_prefill_attn in one launch.  qkv is the dense
 * bf16 projection output [pairs*rows_per_pair][3*hidden]; the rotated K and the V rows of the real
 * tokens are written to the cache (positions == row index inside the pair), Q is never stored. */
int psg_prefill_attn_rope(psg_ctx*, const void* qkv, const int32_t* tok_pos, const float* rope_cos,
                          const float* rope_sin, int pairs, int rows_per_pair, int heads, int head_dim,
                          int ctx, void* k_cache, void* v_cache, void* out, int dtype, void* stream);

/* ---- K13 + K14 fused for the decode step (one new token per pair): rotary + KV-cache append +
 * attention over the cache in one launch.  qkv [rows][3*hidden] (activation dtype, or fp32 split-K
 * partials when qkv_splits > 0); out [rows][hidden].  Equivalent to psg_rope_kvwrite followed by
 * psg_llm_attn for rows that each hold the newest token of their pair. */
int psg_decode_attn(psg_ctx*, const void* qkv, int qkv_splits, const int32_t* tok_pair, const int32_t* tok_pos,
                    const float* rope_cos, const float* rope_sin, int rows, int heads, int head_dim, int ctx, void* k_cache,
                    void* v_cache, void* out, int dtype, void* stream);

/* ---- SwiGLU gate, HF-LL:163-177: out = silu(gate_up[:, :inter]) * gate_up[:, inter:].
 * splits > 0: gate_up is fp32 split-K partials [splits][rows][2*inter]. */
int psg_silu_mul(psg_ctx*, const void* gate_up, int splits, int64_t rows, int inter, void* out, int dtype,
                 void* stream);

/* ---- K15: weight-streaming skinny GEMM of the batched decode step (HF-LL q/k/v/o/gate/up/down
 * projections and lm_head, all bias-free Linear layers): y[M][N] = x[M][K] . w[N][K]^T with
 * M <= 32 rows (the selected pairs), 16-bit in (`dtype` = PSG_BF16 or PSG_F16) / fp32 accumulate; every weight byte is read from
 * HBM exactly once per call.  N % 16 == 0, K % 64 == 0.
 * `dtype` = PSG_F32: x and w are fp32 (the precision the reference runs the LLM in, V4:99-100), exact fp32 arithmetic on
 * v_mfma_f32_16x16x4_f32 (+ v_mfma_f32_4x4x1_16b_f32 for rows 16..19), K % 32 == 0 (psg_gemm_f32.hip).
 * K is split `splits` ways across workgroups (psg_skinny_gemm_plan chooses the count); the kernel
 * writes fp32 partials part[splits][M][N] and does NOT reduce them: the consumers below
 * (psg_rmsnorm, psg_rope_kvwrite, psg_silu_mul, psg_greedy_step) take a `*_splits` argument and sum
 * the slices in split order while loading (deterministic, no atomics, no extra launch);
 * psg_reduce_partials materialises y for any other consumer.
 * The planner also fixes the slab height (8, 11 or 12 wavefronts x 16 rows per workgroup, option skinny_wide) so that
 * the slabs divide evenly over the compute units: a launch lasts as long as the workgroups that walk one slab more. */
int psg_skinny_gemm_plan(psg_ctx*, int M, int N, int K, int dtype, int* splits);
int psg_skinny_gemm(psg_ctx*, const void* x, const void* w, float* part, int M, int N, int K,
                    int splits, int dtype, void* stream);
int psg_reduce_partials(psg_ctx*, const float* part, int splits, int64_t n, void* y, int dtype,
                        void* stream);
/* The same weights (fp16 values) under fp32-GRADE arithmetic on the 16-bit matrix cores, the decode-step form of the
 * fp32s mode: psg_split_f16x2 writes the high and low fp16 parts of M <= 32 fp32 rows as two planes [2][M][K] + the rows'
 * inverse power-of-two scales; psg_split_gemm_w16 streams w (fp16 [N][K]) ONCE and writes fp32 slices
 * part[slots][M][N] = (xh . w + xl . w) * inv_scale[m] for the consumers of psg_skinny_gemm (2^-22 relative: x's split
 * residual; the weight has no low part).  Built on psg_batch_gemm's kernel (x planes as two row tiles, summed at the
 * flush); mode 1 / 2 as there, 0 = the library's estimate. */
int psg_split_f16x2(psg_ctx*, const float* x, int64_t rows, int K, int64_t row_stride, void* out2, float* inv_scale,
                    void* stream);
/* psg_rmsnorm on fp32 rows and fp32 split-K slices with the result written straight as those two planes (bit-identical to
 * psg_rmsnorm followed by psg_split_f16x2; HF-LL:53-67) */
int psg_rmsnorm_split2(psg_ctx*, float* resid, const float* delta, int delta_splits, const float* w, float eps, int64_t rows,
                       int hidden, void* out2, float* inv_scale, void* stream);
int psg_split_gemm_w16_plan(psg_ctx*, int M, int N, int K, int mode, int* slots);
int psg_split_gemm_w16(psg_ctx*, const void* x2, const float* inv_scale, const void* w_f16, float* part, int M, int N, int K,
                       int slots, int mode, void* stream);
/* psg_skinny_gemm(PSG_F32) over weights STORED as fp16: x fp32 [M][K], w fp16 [N][K], the same plan (psg_skinny_gemm_plan
 * with PSG_F32), the same f32 matrix instructions on the exactly widened weights in the same order - part is bit-identical
 * to the fp32-weight call on w.float(), at half the weight bytes.  For weights that ARE fp16 values: the reference's LLM is
 * the frozen Llama-2-7b-hf checkpoint (fp16 on disk; configs/psg/baseline_v4_ov.py:61-65) upcast by from_pretrained
 * (V4:99-100).  The caller verifies the round trip (openpsg_amd/llm.py does, per tensor, at load time). */
int psg_skinny_gemm_w16(psg_ctx*, const float* x, const void* w_f16, float* part, int M, int N, int K, int splits,
                        void* stream);

/* ---- The same projections for 33..160 rows: several images' selected pairs decoded together (head.forward_batch; the
 * reference decodes one pair at a time, V4:293-312).  16-bit operands, N % 16 == 0, K % 64 == 0.  The weight is the
 * streamed operand (every byte from HBM once), x rides along from L2; work = (slab of 256 / 128 weight rows, K step)
 * units dealt to the workgroups as contiguous ranges (stream-K), so that a launch is balanced whatever N is.  Writes
 * fp32 slices part[slots][M][N] for the same consumers as psg_skinny_gemm (slices a slab did not need are written as
 * zeros); psg_batch_gemm_plan returns the slice count for a shape and `slots` must be that number.
 * slab_rows 256 / 128 and mode 1 (ranges aligned to the slabs: one segment per workgroup) / 2 (stream-K ranges) pick a
 * variant, 0 leaves the choice to the library's estimate; the engine times the variants and the library GEMM per shape
 * once and keeps the fastest (openpsg_amd/llm.py). */
int psg_batch_gemm_plan(psg_ctx*, int64_t M, int N, int K, int dtype, int slab_rows, int mode, int* slots);
int psg_batch_gemm(psg_ctx*, const void* x, const void* w, float* part, int64_t M, int N, int K, int slots, int dtype,
                   int slab_rows, int mode, void* stream);

/* ---- Decode-step projection with its producer row operation in the SAME launch.  The decode step alternates a
 * weight-streaming projection with a latency-bound row operation on <= 32 rows (RMSNorm HF-LL:53-67 + residual add,
 * rotary + KV append + attention HF-LL:130-214, SwiGLU gate HF-LL:163-177); as separate launches each row operation
 * costs a kernel (5-9 us) plus a boundary during which no weight byte moves.  Here the first workgroups of the
 * projection's own grid run the row operation while every workgroup's weight ring fills, publish the x operand
 * write-through, and the grid waits on ONE arrival counter before it stages x (agent-scope hand-off inside the launch;
 * the grid is sized to be fully resident and the producers never wait, so it cannot deadlock; the poll is bounded).
 *   kind RMSNORM     : in = split-K partials of the previous projection (or NULL), resid += sum(in); x = norm(resid);
 *                      same arithmetic order as psg_rmsnorm: bit-identical.  K = 4096 or 1024.
 *   kinds DECODE_ATTN / SILU_MUL: fields reserved, rejected with PSG_ERR_UNSUPPORTED - measured on MI355X the hand-off
 *   costs as much as the launch it removes (26.6 vs 25.6 us per RMSNorm + q/k/v projection), so the engine keeps the
 *   separate launches by default (llm.fuse_rowops) and the other two prologues were not built.
 * x [M][K] (activation dtype) is written by the prologue and then read as the GEMM operand.
 * sync: two device words the CALLER zeroes before the launch (stream-ordered memset; one pair per launch inside a
 * captured graph): [0] arrival counter, [1] set to a nonzero code if the bounded poll gave up. */
enum psg_prologue_kind { PSG_PRO_NONE = 0, PSG_PRO_RMSNORM = 1, PSG_PRO_DECODE_ATTN = 2, PSG_PRO_SILU_MUL = 3 };
typedef struct psg_prologue {
  int kind;
  int in_splits;              /* > 0: `in` holds fp32 split-K partials; 0: activation dtype */
  const void* in;
  void* resid;                /* RMSNORM: residual stream [M][K], updated in place when `in` != NULL */
  const float* norm_w;        /* RMSNORM: weight [K] */
  float eps;
  int heads;                  /* DECODE_ATTN: K = heads * 128 */
  int ctx;                    /* DECODE_ATTN: cache slots per (pair, head) */
  const int32_t* tok_pair;    /* DECODE_ATTN: as psg_decode_attn */
  const int32_t* tok_pos;
  const float* rope_cos;
  const float* rope_sin;
  void* k_cache;
  void* v_cache;
  uint32_t* sync;
} psg_prologue;
int psg_skinny_gemm_fused(psg_ctx*, const psg_prologue* pro, void* x, const void* w, float* part, int M, int N, int K,
                          int splits, int dtype, void* stream);

/* ---- one Llama decoder layer of the decode step as ONE persistent launch (V4:293-312 -> HF-LL:53-67, 130-214, 243-281):
 * resid += attention block; resid += MLP block, for M <= 32 rows that each hold the newest token of their pair -
 * bit-identical to the chain psg_rmsnorm -> psg_skinny_gemm(q|k|v) -> psg_decode_attn -> psg_skinny_gemm(o) -> psg_rmsnorm
 * -> psg_skinny_gemm(gate|up) -> psg_silu_mul -> psg_skinny_gemm(down), whose first RMSNorm consumes `delta` (the previous
 * layer's down-projection partials, delta_splits slices, or NULL) and whose last projection leaves its 16 split-K slices
 * in down_part [16][M][hidden] for the next layer (or for the final psg_rmsnorm).  256 workgroups stay resident and keep
 * their weight rings filled across the row operations (csrc/psg_decode_layer.hip).  fp32 weights / activations / caches,
 * hidden = 4096 = 32 heads x 128, inter % 128 == 0, 13..24 rows, a 256-CU device: psg_decode_layer_supported() says
 * whether a shape qualifies (else keep the chain).  workspace: psg_decode_layer_workspace() floats, contents
 * irrelevant; counters: that many uint32 words ZEROED by the caller before every launch (one block per launch inside a
 * captured graph); word [255 * 64] != 0 afterwards = a bounded poll gave up.  At most ONE of these launches may run on a
 * device at a time (every workgroup must be resident). */
int psg_decode_layer_workspace(psg_ctx*, int M, int hidden, int inter, int64_t* floats, int64_t* counters);
int psg_decode_layer_supported(psg_ctx*, int M, int hidden, int inter, int heads, int dtype);
int psg_decode_layer(psg_ctx*, void* resid, const void* delta, int delta_splits, const float* ln1, const float* ln2,
                     const void* wqkv, const void* wo, const void* wgu, const void* wdown, const int32_t* tok_pair,
                     const int32_t* tok_pos, const float* rope_cos, const float* rope_sin, int M, int hidden, int inter,
                     int heads, int ctx_len, float eps, void* k_cache, void* v_cache, float* workspace, uint32_t* counters,
                     float* down_part, int dtype, void* stream);
/* The same for n_layers decoder layers chained inside ONE launch (the whole stack of a decode step).  layer_table: device
 * array of n_layers x 8 pointers {ln1, ln2, wqkv, wo, wgu, wdown, k_cache, v_cache}; layer l leaves its down partials in
 * down_parts + (l & 1) * 16 * M * hidden floats (the caller's final psg_rmsnorm reads buffer (n_layers - 1) & 1);
 * counters: n_layers blocks of psg_decode_layer_workspace()'s size, zeroed. */
int psg_decode_layers(psg_ctx*, void* resid, const void* delta, int delta_splits, const void* layer_table, int n_layers,
                      const int32_t* tok_pair, const int32_t* tok_pos, const float* rope_cos, const float* rope_sin, int M,
                      int hidden, int inter, int heads, int ctx_len, float eps, float* workspace, uint32_t* counters,
                      float* down_parts, int dtype, void* stream);

/* ---- fp32-grade products on the 16-bit matrix cores (prompt pass of the reference-precision mode, V4:99-100 with
 * HF-LL:163-177): an fp32 row, scaled by a power of two so that its largest magnitude lies in [2^13, 2^14), is written as
 * three fp16 K segments - order 0 (activations): [hi | hi | lo], order 1 (weights): [hi | lo | hi], hi = fp16(v),
 * lo = fp16(v - hi) - so that ONE fp16 GEMM over K' = 3K computes xh.wh + xh.wl + xl.wh with fp32 accumulation
 * (error ~3 * 2^-22 per product; psg_split.hip).  order 2 (round 6, either operand of psg_dense_gemm_split; K % 32 == 0):
 * out [rows][2 K], per 32 k [hi(32) | lo(32)] - every value once.  inv_scale[row] = the power of two that undoes the row's scaling;
 * psg_scale_rows_cols applies y[m][n] *= row_scale[m] * col_scale[n] in place (exact). */
int psg_split_f16x3(psg_ctx*, const float* x, int64_t rows, int K, int64_t row_stride, int order, void* out,
                    float* inv_scale, void* stream);
int psg_scale_rows_cols(psg_ctx*, float* y, int64_t rows, int N, const float* row_scale, const float* col_scale,
                        void* stream);

/* Round 5: the same split / un-scaling inside the row kernels around the prompt pass's projections (fp32; bit-identical to
 * the separate kernels).  A raw fp16-GEMM result y travels with its scale vectors and its reader applies
 * y * (row_scale[m] * col_scale[n]) while loading:
 *   psg_rmsnorm_split       resid += delta * scales (delta may be NULL); RMSNorm(resid) * w -> out3 [rows][3 hidden] fp16
 *                           ([hi | hi | lo]) + inv_scale [rows]                              (HF-LL:53-67)
 *                           delta_slices > 1: delta is [slices][rows][hidden] - the K segments of a split product run
 *                           as ONE batched library GEMM (three times the tiles of the narrow o / down products) - summed
 *                           in slice order before the scales
 *   psg_rope_kvwrite_scaled psg_rope_kvwrite on a raw q|k|v result                            (HF-LL:130-160)
 *   psg_silu_mul_split      silu(gate) * up of a raw gate|up result -> out3 [rows][3 inter] + inv_scale   (HF-LL:163-177) */
int psg_rmsnorm_split(psg_ctx*, float* resid, const float* delta, const float* delta_row_scale, const float* delta_col_scale,
                      int delta_slices, const float* w, float eps, int64_t rows, int hidden, void* out, float* inv_scale,
                      int planes, void* stream);
int psg_rope_kvwrite_scaled(psg_ctx*, const float* qkv, const float* row_scale, const float* col_scale,
                            const int32_t* tok_pair, const int32_t* tok_pos, const float* rope_cos, const float* rope_sin,
                            int slices, int64_t rows, int heads, int head_dim, int ctx, float* q_out, float* k_cache,
                            float* v_cache, void* stream);
int psg_silu_mul_split(psg_ctx*, const float* gate_up, const float* row_scale, const float* col_scale, int slices, int64_t rows,
                       int inter, void* out, float* inv_scale, int planes, void* stream);
/* `slices` > 1: the raw product arrives as [slices][rows][N] and is summed in slice order before the scales (K segments of a
 * batched library product; the two planes of a two-plane operand).  `planes` = 3: out = [rows][3 K] = [hi | hi | lo] for
 * a weight with a low part; 2: out = [2][rows][K] (high plane, low plane) for a weight that is an fp16 value - the product
 * is then ONE library GEMM over 2 x rows rows and K, two thirds of the three-segment form's flops. */

/* ---- Q-Former dense projections with fused epilogue (HF-IB:563-596 intermediate(_query): Linear + exact-erf GELU):
 * out[M][N] = epilogue(x[M][K] . w[N][K]^T + bias[N]); x / w / out bf16 or fp16 row-major, bias fp32 (may be NULL),
 * fp32 accumulate.  N % 16 == 0, K % 64 == 0.  One pass instead of a library GEMM + psg_bias_gelu. */
enum psg_epilogue { PSG_EPI_NONE = 0, PSG_EPI_GELU = 1, PSG_EPI_SWIGLU = 2 };
int psg_dense_gemm(psg_ctx*, const void* x, const void* w, const float* bias, int epilogue, void* out, int64_t M,
                   int N, int K, int dtype, void* stream);
/* The same kernel with an fp32 output (out_dtype = PSG_F32) and per-row / per-column scales applied to the accumulator
 * before the bias: out[m][n] = epilogue(acc[m][n] * row_scale[m] * col_scale[n] + bias[n]) - the second half of the
 * split-fp16 product of the fp32s mode (psg_split_f16x3 operands, K' = 3K).  Every output element is one k-ordered
 * accumulation over the whole K (no split-K): a row's result does not depend on M, so a pair shard
 * reproduces the full pass bit for bit.  out_dtype = dtype: as psg_dense_gemm (scales must be NULL). */
int psg_dense_gemm_ex(psg_ctx*, const void* x, const void* w, const float* bias, int epilogue, void* out, int64_t M,
                      int N, int K, int dtype, int out_dtype, const float* row_scale, const float* col_scale,
                      void* stream);
/* The same kernel with the output tile chosen by the caller - the Llama prompt pass (HF-LL:163-177; ~980 rows = 4 row
 * blocks of 256) needs tiles that fill the 256 CUs in whole rounds: PSG_TILE_AUTO picks the geometry with the fewest
 * rounds x tile area for [M, N].  Geometries other than 256 x 256 take the plain and the SwiGLU epilogue with 16-bit
 * output.  PSG_EPI_SWIGLU (Llama MLP): w is the gate / up weight with its rows interleaved in groups of 8 by
 * psg_interleave_gate_up ([2 inter][K] -> w[16 p + r] = gate[8 p + r], w[16 p + 8 + r] = up[8 p + r], r < 8), N = 2 inter,
 * and out is [M][inter] = silu(gate) * up with the roundings of the separate kernels (GEMM output, act_fn(gate),
 * product: psg_silu_mul); bias must be NULL.  Results do not depend on the tile (one k-ordered accumulation each). */
enum psg_tile { PSG_TILE_AUTO = 0, PSG_TILE_256x256 = 1, PSG_TILE_256x192 = 2, PSG_TILE_256x128 = 3, PSG_TILE_256x64 = 4,
                PSG_TILE_128x128 = 5 };
int psg_dense_gemm_tiled(psg_ctx*, const void* x, const void* w, const float* bias, int epilogue, void* out, int64_t M,
                         int N, int K, int dtype, int out_dtype, const float* row_scale, const float* col_scale, int tile,
                         void* stream);
int psg_interleave_gate_up(psg_ctx*, const void* gate_up, void* out, int inter, int K, void* stream);
/* Round 6: the split-fp16 product with every operand value staged ONCE.  x2 [M][K2], w2 [N][K2] fp16 are
 * psg_split_f16x3(order = 2) images of fp32 matrices - per 32 k [hi(32) | lo(32)], K2 = 2 K - and
 * out[m][n] = epilogue((xh.wh + xh.wl + xl.wh)[m][n] * row_scale[m] * col_scale[n] + bias[n]) in fp32: the three products
 * of psg_dense_gemm_ex's K' = 3K form (HF-IB / HF-LL nn.Linear at the reference's fp32, V4:78-84, 99-100) from one
 * staging of each part, 2/3 of the operand bytes per flop.  One k-ordered accumulation per element (row-count and tile
 * invariant).  epilogue: PSG_EPI_NONE | PSG_EPI_GELU; tile: PSG_TILE_AUTO, 256x256, 256x128, 256x64, 128x128. */
int psg_dense_gemm_split(psg_ctx*, const void* x2, const void* w2, const float* bias, int epilogue, float* out, int64_t M,
                         int N, int K2, const float* row_scale, const float* col_scale, int tile, void* stream);

/* ---- K16: greedy step (HF generate, num_beams=1, do_sample=False; V4:305-312).
 * logits [K][vocab] (dtype); token = argmax (first maximal index); suppress_token >= 0 is
 * excluded.  For each pair k not yet done: tokens[k][step] = token, done[k] |= (token == eos),
 * next_ids[k] = token, tok_pos[k] += 1.  Finished pairs write -1 and keep decoding harmlessly.
 * splits > 0: logits is fp32 split-K partials [splits][K][vocab] (summed before the argmax).
 * x_out (may be NULL) [K][hidden] (x_dtype): receives embed[token] (V4:296 / HF embed_tokens of the next step's input) -
 * the embedding gather of the next decode step folded into this launch. */
int psg_greedy_step(psg_ctx*, const void* logits, int splits, int K, int vocab, int step, int max_new,
                    int eos, int suppress_token, int32_t* tokens, int32_t* done, int32_t* next_ids, int32_t* tok_pos,
                    const void* embed, int embed_dtype, int hidden, void* x_out, int x_dtype, int dtype, void* stream);

/* ---- masked-mean object pooling of the v1-v3 detectors
 * (kings_sgg/models/detectors/openseed_relation.py:453-468):
 * out[n][c] = sum feat[c][y][x] * m_n[y][x] / (sum m_n + 1e-8), m_n = mask of object_ids[n] in the id
 * map `pan`, resampled nearest(ori->img) -> zero pad -> nearest(pad->feature resolution) as
 * :456-461 do (padding belongs to no object).  For disjoint (panoptic) masks every feature element
 * is read exactly once; deterministic.  workspace: int32, size from psg_masked_mean_pool_workspace. */
int psg_masked_mean_pool_workspace(psg_ctx*, int C, int Hf, int Wf, int N, int64_t* bytes);
int psg_masked_mean_pool(psg_ctx*, const float* feat, int C, int Hf, int Wf, const int32_t* pan, int H0,
                         int W0, int img_h, int img_w, int pad_h, int pad_w, const int32_t* object_ids,
                         int N, float* out, int32_t* workspace, int64_t workspace_bytes, void* stream);
/* split-mean variant, `_mask_pooling(output_size > 1)` (openseed_relation.py:175-200): the object's pixels in row-major
 * order cut into `output_size` contiguous chunks (the first count mod output_size one longer), one mean per chunk; fewer
 * pixels than chunks: the pixel list repeats; no pixels: zeros.  out [N][output_size][C]; same workspace. */
int psg_masked_split_mean_pool(psg_ctx*, const float* feat, int C, int Hf, int Wf, const int32_t* pan, int H0,
                               int W0, int img_h, int img_w, int pad_h, int pad_w, const int32_t* object_ids, int N,
                               int output_size, float* out, int32_t* workspace, int64_t workspace_bytes, void* stream);

/* ---- 8f rank 3: training branch of the head (forward arithmetic of the losses).
 * psg_train_object_bitmasks: V4:371-399 prepare_train.  thing_masks uint8 [n_thing][H][W] (padded ground-truth
 *   masks) resampled to the gh x gw patch grid by bilinear interpolation (align_corners=False) > 0.5; stuff objects:
 *   nearest-resampled semantic map sem int32 [H][W] == category.  Object n is thing #thing_index[n] when
 *   is_thing[n].  bits [N][words] uint64 as psg_object_bitmasks.
 * psg_bce_with_logits: V4:463-482 (binary case): out[0] = weight * mean BCE-with-logits.
 * psg_cross_entropy_rows: V4:337-341: loss[row] = logsumexp(logits[row]) - logits[row][label]; label < 0 (the
 *   reference's ignore_index -100) gives 0; the caller averages over the rows that count. */
int psg_train_object_bitmasks(psg_ctx*, const uint8_t* thing_masks, int n_thing, const int32_t* sem, int H, int W,
                              const int32_t* is_thing, const int32_t* category, const int32_t* thing_index, int N,
                              int gh, int gw, uint64_t* bits, int words, void* stream);
int psg_bce_with_logits(psg_ctx*, const float* logit, const float* label, int n, float weight, float* out,
                        void* stream);
int psg_cross_entropy_rows(psg_ctx*, const void* logits, int64_t rows, int vocab, const int32_t* labels, float* loss,
                           int dtype, void* stream);

/* ---- SURVEY 8f rank 3, gradient path (V4:327-351, 463-482; the reference back-propagates both losses, LLM frozen,
 * CFG:65): fp32 row / attention kernels of the training branch with their exact adjoints.  Batches are tiny (<= 32
 * sampled pairs, <= 4 LLM pairs); dense projections and weight gradients go through the library GEMM.
 * attention: q [B][Sq][H*D], k / v [Bk][Sk][H*D] (Bk == B, or 1 = shared by every sequence: the image's cross-attention
 * K/V), keep uint8 [B][Mq][Sk] (Mq == Sq, or 1 = one key mask for all query rows), p [B][H][Sq][Sk] saved by the forward;
 * an all-masked row is a uniform softmax (additive finfo.min).  The backward ACCUMULATES into dk / dv / dgamma / dbeta
 * (caller zeroes them). */
int psg_train_layernorm_fwd(psg_ctx*, const float* x, const float* gamma, const float* beta, float eps, int64_t rows,
                            int hidden, float* y, float* mean, float* rstd, void* stream);
int psg_train_layernorm_bwd(psg_ctx*, const float* x, const float* dy, const float* gamma, const float* mean,
                            const float* rstd, int64_t rows, int hidden, float* dx, float* dgamma, float* dbeta,
                            void* stream);
int psg_train_rmsnorm_fwd(psg_ctx*, const float* x, const float* w, float eps, int64_t rows, int hidden, float* y,
                          float* rstd, void* stream);
int psg_train_rmsnorm_bwd(psg_ctx*, const float* x, const float* dy, const float* w, const float* rstd, int64_t rows,
                          int hidden, float* dx, void* stream);
/* drop (may be NULL): uint8 keep mask [B][H][Sq][Sk] of the attention-probability dropout the reference trains with
 * (HF-IB:176-196, InstructBlipQFormerConfig.attention_probs_dropout_prob = 0.1), drop_scale = 1 / (1 - p_drop). */
int psg_train_attn_fwd(psg_ctx*, const float* q, const float* k, const float* v, const uint8_t* keep, int B, int Bk,
                       int H, int Sq, int Sk, int D, int Mq, float scale, const uint8_t* drop, float drop_scale, float* p,
                       float* out, void* stream);
int psg_train_attn_bwd(psg_ctx*, const float* q, const float* k, const float* v, const float* p, const float* dout,
                       int B, int Bk, int H, int Sq, int Sk, int D, float scale, const uint8_t* drop, float drop_scale,
                       float* dq, float* dk, float* dv, void* stream);
int psg_train_gelu_fwd(psg_ctx*, const float* x, int64_t n, float* y, void* stream);
int psg_train_gelu_bwd(psg_ctx*, const float* x, const float* dy, int64_t n, float* dx, void* stream);
int psg_train_silu_mul_fwd(psg_ctx*, const float* gate_up, int64_t rows, int inter, float* y, void* stream);
int psg_train_silu_mul_bwd(psg_ctx*, const float* gate_up, const float* dy, int64_t rows, int inter, float* dgate_up,
                           void* stream);
/* half-split rotary on [rows][heads*head_dim] with per-row table index `pos`; sign = -1 is the adjoint */
int psg_train_rope(psg_ctx*, const float* x, const int32_t* pos, const float* rope_cos, const float* rope_sin,
                   int table_rows, int64_t rows, int heads, int head_dim, float sign, float* y, void* stream);
int psg_train_ce_bwd(psg_ctx*, const float* logits, int64_t rows, int vocab, const int32_t* labels, const float* dloss,
                     float* dlogits, void* stream);
int psg_train_bce_bwd(psg_ctx*, const float* logit, const float* label, int n, float weight, const float* dloss,
                      float* dlogit, void* stream);

/* ---- 8f: bilinear relation scorer of the closed-set heads (relation_transformer_head_v2.py:204-209):
 * pred[b][r][s][o] = sum_c sub[b][s][r*C + c] * obj[b][o][r*C + c], i.e. einsum('nrsc,nroc->nrso') on the
 * outputs of the two Linear layers in their natural [B][N][R*C] layout.  Exact fp32 (f32 matrix cores). */
int psg_bilinear_scores(psg_ctx*, const float* sub, const float* obj, int B, int N, int R, int C, float* pred,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PSG_HIP_H */
