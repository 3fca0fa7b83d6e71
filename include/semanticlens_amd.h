/*
 * semanticlens_amd.h — C ABI of libsemanticlens_hip.so (gfx950 / MI355X).
 *
 * The reference (jim-berend/semanticlens v0.2.1) is pure Python and has no FFI;
 * its drop-in boundary is the Python plugin API (Lens / ComponentVisualizer /
 * foundation_models).  This header is the native boundary the
 * build places *behind* that API: every entry point replaces the stock-torch
 * arithmetic of one reference call site, cited per function as
 * `semanticlens/<file>:<line>` (paths relative to the reference root).
 *
 * Conventions
 *  - Plain pointers and sizes only.  Pointers named d_* are DEVICE pointers
 *    (HBM, e.g. torch.Tensor.data_ptr()); h_* are host pointers read during
 *    the call.  Nothing is retained after a call returns.
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All
 *    work is enqueued asynchronously on it; no entry point synchronises except
 *    sl_prof_read().
 *  - Return value: 0 (or a non-negative code where documented) on success,
 *    negative SL_E_* on failure; sl_last_error() returns a thread-local message.
 *  - No entry point allocates device memory; scratch is passed in by the caller.  The only object that
 *    outlives a call is the RCCL communicator of sl_comm_init_from_unique_id / sl_comm_destroy.
 *  - Strides are in ELEMENTS.
 */
#ifndef SEMANTICLENS_AMD_H
#define SEMANTICLENS_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SL_ABI_VERSION 1

/* error codes */
#define SL_E_INVALID (-1)   /* bad argument (message says which) */
#define SL_E_HIP (-2)       /* a HIP runtime call failed */
#define SL_E_UNSUPPORTED (-3)

/* element types of activation tensors */
#define SL_F32 0
#define SL_F16 1
#define SL_BF16 2

/* aggregation over H*W of a (B,C,H,W) activation — component_visualization/aggregators.py:38-87 */
#define SL_CONV_MAX 0  /* aggregate_conv_max  (:64-87)  */
#define SL_CONV_MEAN 1 /* aggregate_conv_mean (:38-61)  */
#define SL_CONV_SUM 2  /* plain sum over H*W: zennit-crp's max_target="sum" used by the relevance visualizer
                          (component_visualization/relevance_based.py:111,140-145) */

/* aggregation over tokens of a (B,T,F) activation — aggregators.py:90-244 */
#define SL_TOK_MEAN 0    /* aggregate_transformer_mean    (:90-114)  */
#define SL_TOK_ABSMEAN 1 /* aggregate_transformer_absmean (:117-141) */
#define SL_TOK_MAX 2     /* aggregate_transformer_max     (:144-168) */
#define SL_TOK_ABSMAX 3  /* aggregate_transformer_absmax  (:171-195) */
#define SL_TOK_TOKEN 4   /* get_aggregate_transformer_special_token(pos) (:198-244) */

/* tie order of the streaming top-k */
#define SL_TIES_TOTAL 0 /* value desc (NaN first, -0 == +0), then sample id asc — batch/shard invariant */
#define SL_TIES_ATEN 1  /* torch.topk's CPU order on cat([state, batch]) — bit-identical to the reference */

#define SL_MAX_SLOTS 16

const char* sl_last_error(void);
int sl_abi_version(void);
/* number of HIP devices visible, or SL_E_HIP */
/* Variant switches — which of several BIT-IDENTICAL kernel variants a dispatcher picks (0 = its own rule).  The parity tests walk the
 * variants with these; production code never needs them.  Names: "g3_tile" (split-bf16 GEMM tile: 128, 256, 8, 160, 64, 1280),
 * "f32_tile" (fp32-MFMA GEMM: 128, 8), "g3_strip_off" (1: no column-strip split), "colreduce_nw" (K2 waves per task: 4, 8, 16).
 * The environment variable SL_OPTIONS="name=value,..." presets them for a process; an explicit call wins.  No reference counterpart. */
int sl_set_option(const char* name, int64_t value);
int64_t sl_get_option(const char* name); /* -1 for an unknown name */
int sl_device_count(void);

/* ---- K1: spatial reduce of a conv activation -------------------------------------------
 * Replaces `tensor.clone().flatten(2).amax(-1)` / `.mean(-1)` (aggregators.py:61, :87) and
 * the bf16 cast `acts.T.to(bfloat16)` (activation_caching.py:133).
 * d_act: (B,C,S) with element strides (sb,sc,ss); S = H*W flattened (NCHW: sc=S, ss=1;
 * channels_last: sc=1, ss=C).
 * d_cand_bf16 (B,C) u16, may be NULL: aggregated value rounded like the reference does
 *   (to the activation dtype, then to bf16 RNE; NaN -> 0x7FC0).
 * d_out_f32 (B,C), may be NULL: aggregated value before the bf16 cast (what the Python
 *   aggregator returns).  At least one output must be non-NULL. */
int sl_reduce_conv(const void* d_act, int dtype, int64_t B, int64_t C, int64_t S, int64_t sb, int64_t sc,
                   int64_t ss, int agg, uint16_t* d_cand_bf16, float* d_out_f32, void* stream);

/* Cache policy of K1's row streams, process-wide.  A COLD input streams best with the read-once (nt) policy; an input the
 * previous kernel on the stream wrote microseconds ago is partly still in the 256 MiB Infinity Cache and reads faster
 * with the default policy.  Inputs smaller than nt_min_bytes are read with the default policy; of larger ones the last
 * tail_bytes likewise, the rest with nt.  Defaults (also: environment SL_NT_MIN_BYTES, SL_REDUCE_TAIL_MB in MiB):
 * 256 MiB / 240 MiB, i.e. "the input was just produced" — what a forward hook sees.  (0, 0) = everything nt, for
 * inputs known to be cold; negative values restore the defaults. */
int sl_set_reduce_policy(int64_t nt_min_bytes, int64_t tail_bytes);

/* Relevance visualizer (relevance_based.py:112, abs_norm=True -> zennit-crp ChannelConcept.reference_sampling):
 * d_x (B,C) fp32 in place, x[b][:] /= (sum_c |x[b][c]| + eps). */
int sl_abs_norm_rows(float* d_x, int64_t B, int64_t C, float eps, void* stream);

/* ---- K2: token reduce of a transformer activation ---------------------------------------
 * Replaces aggregators.py:114,141,168,195,242.  d_act: (B,T,F), strides (sb,st,sf).
 * `pos` is used by SL_TOK_TOKEN only (python-style negative index allowed). */
int sl_reduce_tokens(const void* d_act, int dtype, int64_t B, int64_t T, int64_t F, int64_t sb, int64_t st,
                     int64_t sf, int agg, int64_t pos, uint16_t* d_cand_bf16, float* d_out_f32, void* stream);

/* The same reductions over L activations of ONE shape (the hooked outputs of L identical blocks, activation_caching.py:388-418
 * fires once per hooked layer and batch) in one launch where the component axis is contiguous, tensor by tensor otherwise.
 * h_d_acts: host array of L device pointers; d_cand_bf16: (L, B, C) contiguous. */
int sl_reduce_conv_multi(const void* const* h_d_acts, int L, int dtype, int64_t B, int64_t C, int64_t S, int64_t sb, int64_t sc,
                         int64_t ss, int agg, uint16_t* d_cand_bf16, void* stream);
int sl_reduce_tokens_multi(const void* const* h_d_acts, int L, int dtype, int64_t B, int64_t T, int64_t F, int64_t sb, int64_t st,
                           int64_t sf, int agg, int64_t pos, uint16_t* d_cand_bf16, void* stream);

/* ---- K3: streaming top-k state (ActMax) --------------------------------------------------
 * State = d_vals (C,k) bf16 bit patterns + d_ids (C,k) int64, sorted best-first per row.
 * sl_actmax_init: activation_caching.py:101-110 (values -0.0, ids -1). */
int sl_actmax_init(uint16_t* d_vals, int64_t* d_ids, int64_t C, int64_t k, void* stream);

/* Merge `nslots` candidate batches into the state under SL_TIES_TOTAL
 * (ActMax.update, activation_caching.py:112-141, for several batches at once).
 * Slot s holds h_slot_rows[s] samples as a (rows,C) bf16 matrix at
 * d_cand + s*slot_stride (elements); sample b of slot s has id
 * h_slot_id_base[s] + b (the per-layer counter of activation_caching.py:410-413). */
int sl_actmax_merge(uint16_t* d_vals, int64_t* d_ids, int64_t C, int64_t k, const uint16_t* d_cand,
                    int64_t slot_stride, const int64_t* h_slot_id_base, const int64_t* h_slot_rows, int nslots,
                    void* stream);

/* One ActMax.update (activation_caching.py:112-141): d_cand (B,C) bf16; sample b has id
 * d_sample_ids[b] (device int64 array) or, when d_sample_ids is NULL, id_base + b.
 * ties = SL_TIES_TOTAL or SL_TIES_ATEN.  SL_TIES_ATEN needs d_ws of
 * sl_actmax_aten_ws_bytes(C,k,B) bytes and k + B <= 16384. */
int sl_actmax_update(uint16_t* d_vals, int64_t* d_ids, int64_t C, int64_t k, const uint16_t* d_cand,
                     const int64_t* d_sample_ids, int64_t id_base, int64_t B, int ties, void* d_ws, size_t ws_bytes,
                     void* stream);
size_t sl_actmax_aten_ws_bytes(int64_t C, int64_t k, int64_t B);
/* SL_TIES_ATEN update of L states (their own component counts h_Cs[l], one k) from L candidate matrices (B, C_l) in ONE launch
 * — every hooked layer of a forward pass merged once per batch instead of once per layer (activation_caching.py:388-418 fires per
 * layer).  h_id_bases: the sample id of each layer's first candidate row (the reference's per-layer counter, :410-413).
 * `_supported`: 1 when (k + B) rows fit the one-wavefront-per-row kernel, else update layer by layer. */
int sl_actmax_update_multi(uint16_t* const* h_d_vals, int64_t* const* h_d_ids, const int64_t* h_id_bases, const int64_t* h_Cs,
                           const uint16_t* const* h_d_cands, int L, int64_t k, int64_t B, void* stream);
int sl_actmax_update_multi_supported(int64_t C, int64_t k, int64_t B);
/* HOST-only (no device, no stream): the n-element row h_vals_bf16 in, the k positions torch.topk's CPU kernel would select
 * (activation_caching.py:140: `torch.topk(all_acts, k, dim=1)`; ATen TopKImpl.h -> libstdc++ partial_sort / nth_element + sort,
 * comparator (isnan(a) && !isnan(b)) || a > b) out, best first.  The restatement SL_TIES_ATEN's kernels evaluate; the host side runs
 * it against the installed torch.topk once per process (a torch / libstdc++ pair with another tie order is reported, not followed). */
int sl_aten_topk_order_host(const uint16_t* h_vals_bf16, int64_t n, int64_t k, int32_t* h_positions);

/* ---- K4: merge R other states (e.g. all-gathered per-rank states) into this one ----------
 * No reference counterpart (the reference is single-process); semantics = SL_TIES_TOTAL
 * top-k of the union.  d_other_vals/ids are (R,C,k). */
int sl_actmax_merge_states(uint16_t* d_vals, int64_t* d_ids, int64_t C, int64_t k, const uint16_t* d_other_vals,
                           const int64_t* d_other_ids, int64_t R, void* stream);

/* ---- K4 with its exchange step: RCCL behind this ABI --------------
 * No reference counterpart: the reference is single-process (no torch.distributed / NCCL call anywhere).  One
 * process per GPU; rank r collects the sample range [r*ceil(N/R), ...) with global ids into its own (C,k) states.
 * The communicator is the ONE object of this library that outlives a call: rank 0 obtains an id
 * (sl_comm_unique_id, 128 bytes, host), hands it to the other processes by any means (the Python host uses the
 * torch.distributed store it was launched with), and every rank calls sl_comm_init_from_unique_id on its current HIP
 * device (ncclCommInitRank: collective).  The library links librccl. */
#define SL_COMM_ID_BYTES 128
int sl_comm_unique_id(uint8_t* h_id);
int sl_comm_init_from_unique_id(const uint8_t* h_id, int world, int rank, void** comm);
int sl_comm_destroy(void* comm);
int sl_comm_info(void* comm, int* world, int* rank, int* device);
/* ncclAllGather of nbytes per rank: d_recv (world * nbytes) receives rank r's block at r * nbytes. */
int sl_comm_allgather(void* comm, const void* d_send, void* d_recv, int64_t nbytes, void* stream);
/* in-place ncclAllReduce of n elements (the sharded K5 gather is assembled by a SUM of zero-filled blocks; the
 * bench's max-over-ranks time is a MAX of one double). */
#define SL_COMM_F32 0
#define SL_COMM_F64 1
#define SL_COMM_I64 2
#define SL_COMM_SUM 0
#define SL_COMM_MAX 1
#define SL_COMM_MIN 2
int sl_comm_allreduce(void* comm, void* d_buf, int64_t n, int dtype, int op, void* stream);
/* The cross-rank merge in one call: pack the (C_l,k) states of n_layers layers (h_vals[l] bf16 bits, h_ids[l] int64:
 * host arrays of DEVICE pointers) into d_ws -> ONE ncclAllGather over xGMI -> K4 (SL_TIES_TOTAL top-k of the union) of
 * every layer against the other ranks' blocks, read in place from the gathered buffer.  Afterwards every rank holds the
 * same global states.  d_ws: sl_actmax_allgather_merge_ws_bytes(...) bytes, 16-byte aligned, owned by the caller. */
/* The two local halves on their own (what a caller that brings its own transport uses; the gloo test path does):
 * sl_actmax_pack writes one rank's block — [ids of layer 0..L-1 | values of layer 0..L-1 | zero pad to 16 bytes],
 * sl_actmax_packed_bytes(...) bytes — and sl_actmax_merge_packed folds R such blocks (d_gathered, block r at
 * r * packed_bytes, 8-byte aligned) into the states, leaving block skip_rank (-1: none) out. */
size_t sl_actmax_packed_bytes(int n_layers, const int64_t* h_C, int64_t k);
int sl_actmax_pack(int n_layers, uint16_t* const* h_vals, int64_t* const* h_ids, const int64_t* h_C, int64_t k, void* d_out,
                   void* stream);
int sl_actmax_merge_packed(int n_layers, uint16_t* const* h_vals, int64_t* const* h_ids, const int64_t* h_C, int64_t k,
                           const void* d_gathered, int64_t R, int64_t skip_rank, void* stream);
size_t sl_actmax_allgather_merge_ws_bytes(int n_layers, const int64_t* h_C, int64_t k, int world);
int sl_actmax_allgather_merge(void* comm, int n_layers, uint16_t* const* h_vals, int64_t* const* h_ids, const int64_t* h_C,
                              int64_t k, void* d_ws, size_t ws_bytes, void* stream);

/* ---- K5: concept_db[layer] = embeds[sample_ids] (activation_based.py:387-390) ------------
 * d_emb (N,D) f32, d_ids (n_ids) int64; negative ids wrap (id -1 -> row N-1).  An id
 * outside [-N, N) sets *d_err_flag (int32, may be NULL) to 1 and reads row 0. */
int sl_gather_rows(const float* d_emb, int64_t N, int64_t D, const int64_t* d_ids, int64_t n_ids, float* d_out,
                   int32_t* d_err_flag, void* stream);

/* Sharded form (multi-GPU): d_emb_local holds rows [row_offset, row_offset +
 * n_local) of a table of n_total rows.  Ids index the whole table (negative ids wrap by n_total);
 * rows held elsewhere are written as ZEROS, so a sum (all-reduce) over the shards equals
 * sl_gather_rows on the whole table. */
int sl_gather_rows_shard(const float* d_emb_local, int64_t n_local, int64_t D, const int64_t* d_ids, int64_t n_ids,
                         int64_t row_offset, int64_t n_total, float* d_out, int32_t* d_err_flag, void* stream);

/* ---- K6: similarity_score (scores.py:84-128) ---------------------------------------------
 * Returns the branch taken: 0 row-wise cosine (shapes equal; out (xr,)), 1 normalize(x) @
 * normalize(y) (xc == yr; out (xr,yc)), 2 normalize(x) @ normalize(y)^T (xc == yc; out
 * (xr,yr)); SL_E_INVALID for incompatible shapes (the reference raises ValueError).
 * d_ws: scratch of sl_similarity_ws_bytes(...) bytes.
 * Arithmetic of the plain branch: split-bf16 x3 on the bf16 matrix cores (|error| ~1e-6 on cosines) when
 * K >= 64, else fp32-input MFMA; SL_GEMM_MODE=f32 in the environment forces the latter. */
int sl_similarity(const float* d_x, int64_t xr, int64_t xc, const float* d_y, int64_t yr, int64_t yc, float* d_out,
                  void* d_ws, size_t ws_bytes, void* stream);
size_t sl_similarity_ws_bytes(int64_t xr, int64_t xc, int64_t yr, int64_t yc);

/* Arithmetic of the cosine GEMMs of sl_similarity(_multi) / sl_redundancy: 1 = split-bf16 x3 (default), 0 = fp32-input
 * MFMA, -1 = as the SL_GEMM_MODE environment variable says.  Process-wide. */
int sl_set_gemm_mode(int mode);

/* One query matrix against L concept matrices — the per-layer loop of `_probe` (lens.py:206-214).  All
 * layers must take similarity_score's plain branch (Q != C_l and K != C_l; otherwise use sl_similarity).
 * h_d_ys / h_d_outs: host arrays of L device pointers ((C_l,K) inputs, (Q,C_l) outputs); h_Cs: host array. */
int sl_similarity_multi(const float* d_x, int64_t Q, int64_t K, const float* const* h_d_ys, const int64_t* h_Cs, int L,
                        float* const* h_d_outs, void* d_ws, size_t ws_bytes, void* stream);
size_t sl_similarity_multi_ws_bytes(int64_t Q, int64_t K, const int64_t* h_Cs, int L);

/* ---- K7: clarity_score (scores.py:18-47): V (C,n,D) -> out (C) ---------------------------- */
int sl_clarity(const float* d_V, int64_t C, int64_t n, int64_t D, float* d_out, void* stream);
/* The per-layer loop of `Lens.eval_clarity` over a concept_db dict (lens.py:391-419) as ONE launch: L layers with the same
 * (n, D) and their own component counts.  h_d_Vs / h_d_outs: host arrays of L device pointers ((C_l,n,D) inputs, (C_l) outputs). */
int sl_clarity_multi(const float* const* h_d_Vs, const int64_t* h_Cs, int L, int64_t n, int64_t D, float* const* h_d_outs,
                     void* stream);

/* ---- K8: redundancy_score (scores.py:50-81): V (Bt,C,D) -> out (Bt) ----------------------- */
int sl_redundancy(const float* d_V, int64_t Bt, int64_t C, int64_t D, float* d_out, void* d_ws, size_t ws_bytes,
                  void* stream);
size_t sl_redundancy_ws_bytes(int64_t Bt, int64_t C, int64_t D);

/* ---- K9: polysemanticity_score (scores.py:131-185): V (C,n,D) -> out (C) float64 ----------
 * Per component: scikit-learn KMeans(n_clusters=2, n_init, random_state) restated on the
 * component's n x n Gram matrix (k-means++ seeding with the caller-supplied random draws,
 * Lloyd iterations with sklearn's tolerance / strict-convergence rules, best-of-n_init by
 * inertia), then 1 - cos(center_1, center_2); rows whose smaller cluster has < 2 samples use
 * the reference's fallback (scores.py:173-184).
 * h_first_center (n_init) int32 and h_rand (n_init,2) float64 are the draws
 * numpy.random.RandomState(random_state) yields for `choice(n, p=uniform)` and
 * `uniform(size=2)` of each init (data independent; produced by the host wrapper).
 * replace_empty_clusters: apply that fallback (the reference's default) or not.
 * d_min_count (C) int32, may be NULL: size of the smaller cluster of the chosen clustering. */
int sl_poly2means(const float* d_V, int64_t C, int64_t n, int64_t D, const int32_t* h_first_center, int n_init,
                  const double* h_rand, int replace_empty_clusters, double* d_out, int32_t* d_min_count, void* d_ws,
                  size_t ws_bytes, void* stream);
size_t sl_poly2means_ws_bytes(int64_t C, int64_t n, int64_t D);

/* The same for any `n_clusters` in [2, 16] (scores.py:132,167 forwards the argument to scikit-learn) and n <= 1024:
 * sklearn's k-means++ with 2 + int(ln k) local trials per further centre, Lloyd with relocation of every empty cluster
 * (ascending cluster id, farthest points first), score = 1 - clarity_score of the k centres.  h_rand is
 * (n_init, n_clusters - 1, sl_kmeans_trials(n_clusters)) float64: RandomState.uniform draws in sklearn's order.
 * State lives in the workspace (L2) rather than LDS: the fallback for what sl_poly2means does not cover. */
int sl_kmeans_trials(int n_clusters);
int sl_polykmeans(const float* d_V, int64_t C, int64_t n, int64_t D, int n_clusters, const int32_t* h_first_center, int n_init,
                  const double* h_rand, int replace_empty_clusters, double* d_out, int32_t* d_min_count, void* d_ws,
                  size_t ws_bytes, void* stream);
size_t sl_polykmeans_ws_bytes(int64_t C, int64_t n, int64_t D, int n_clusters, int n_init);

/* ---- K10: template-difference mean of text embeddings (lens.py:196-199) ------------------
 * E (Q*T,D) read as "(q t) d", E0 (T,D); out (Q,D) = mean_t(E[q,t] - E0[t]). */
int sl_template_mean(const float* d_E, const float* d_E0, int64_t Q, int64_t T, int64_t D, float* d_out,
                     void* stream);

/* ---- K11: primitives of a native CLIP-family transformer tower ----------------
 * The reference's OpenClip.encode_image / encode_text (foundation_models/clip.py:103-135) forward to
 * the third-party open_clip model; these entry points run the same pre-LN transformer arithmetic in
 * fp32 on the device (orchestrated by semanticlens_amd/foundation_models/native_clip.py). */
#define SL_ACT_NONE 0
#define SL_ACT_GELU 1      /* exact erf GELU (torch.nn.GELU()) */
#define SL_ACT_QUICKGELU 2 /* x * sigmoid(1.702 x) (OpenAI CLIP checkpoints) */
#define SL_ACT_GELU_TANH 3 /* tanh approximation (torch gelu(approximate="tanh"); SigLIP towers) */
/* out = act(x W^T + bias) (+ residual): x (M,K), W (N,K) row-major (torch Linear.weight), bias (N) or
 * NULL, residual (M,ldo) or NULL (may alias out), out row stride ldo >= N.  fp32-input MFMA GEMM.
 * Row scatter (patch embedding): when rows_per_group > 0, row r is written to
 *   (r / rows_per_group) * group_stride + row_offset + r % rows_per_group
 * and d_rowadd[(row_offset + r % rows_per_group), :] of a (T,N) table (positional embedding) is added. */
int sl_linear(const float* d_x, int64_t M, int64_t K, const float* d_w, int64_t N, const float* d_bias, int act,
              const float* d_residual, float* d_out, int64_t ldo, int64_t rows_per_group, int64_t group_stride,
              int64_t row_offset, const float* d_rowadd, void* stream);
/* Split matrices: every fp32 value v is carried as hi = bf16(v), lo = bf16(v - hi).  An (R,K) matrix is stored as R
 * rows of 2*Kp bf16 values (Kp = K rounded up to 32, zero padded, the padding must stay zero); each 32-wide k-tile of
 * a row is one 128-byte line [hi(32) | lo(32)]; the buffer is 128-byte aligned and holds sl_split_elems(R,K) uint16.
 * sl_linear_bf16x3 multiplies such operands with three bf16 MFMAs per product (hi*hi + hi*lo + lo*hi, fp32
 * accumulate): fp32-class accuracy at ~2.4x the speed of sl_linear.  Producers (sl_layernorm, sl_attention,
 * sl_patchify, and sl_linear_bf16x3 itself) can emit the split form directly through d_out_split (then d_out is
 * NULL); sl_split_bf16 converts an fp32 matrix (optionally scaling each row first) and writes the zero padding. */
size_t sl_split_elems(int64_t R, int64_t K);
int sl_split_bf16(const float* d_x, const float* d_row_scale, int64_t R, int64_t K, uint16_t* d_split, void* stream);
int sl_linear_bf16x3(const uint16_t* d_x_split, int64_t M, int64_t K, const uint16_t* d_w_split, int64_t N,
                     const float* d_bias, int act, const float* d_residual, float* d_out, uint16_t* d_out_split, int64_t ldo,
                     int64_t rows_per_group, int64_t group_stride, int64_t row_offset, const float* d_rowadd, void* stream);
/* LayerNorm over the last dim (biased variance, like torch.nn.LayerNorm); row strides in elements (fp32 output);
 * d_out_split: (rows, cols) split matrix. */
int sl_layernorm(const float* d_x, int64_t rows, int64_t cols, int64_t x_row_stride, const float* d_gamma,
                 const float* d_beta, float eps, float* d_out, uint16_t* d_out_split, int64_t out_row_stride, void* stream);
/* softmax(q k^T / sqrt(head_dim)) v per (batch, head); d_qkv (B*T, 3*H*head_dim) rows [q | k | v]
 * (torch MultiheadAttention in_proj layout), out (B*T, H*head_dim) fp32 or split; causal != 0 masks keys j > i.
 * head_dim in {32, 64, 72, 80, 88, 96, 104, 128}; any sequence length (K/V stream through LDS in chunks). */
int sl_attention(const float* d_qkv, int64_t B, int64_t T, int64_t H, int64_t head_dim, int causal, float* d_out,
                 uint16_t* d_out_split, void* stream);
/* sl_attention in the split-bf16 x3 arithmetic of sl_linear_bf16x3: both products (q k^T and p v) as a_lo b_hi + a_hi b_lo +
 * a_hi b_hi on the bf16 matrix cores, fp32 softmax and accumulation (~1e-5 relative); same layouts, arguments and head_dims.
 * What NativeClip / NativeSigLip call between two sl_linear_bf16x3 (reference: the attention inside open_clip's towers,
 * foundation_models/clip.py:103-135). */
int sl_attention_bf16x3(const float* d_qkv, int64_t B, int64_t T, int64_t H, int64_t head_dim, int causal, float* d_out,
                        uint16_t* d_out_split, void* stream);
/* Attention pooling with one query per head (the MAP head of SigLIP image towers, clip.py:190-211 SigLipV2 /
 * open_clip attn_pool): out (B, H*head_dim) = softmax(q k_t / sqrt(head_dim)) v over the T tokens of each image.
 * d_q (H*head_dim) is the projected probe; key row (b, t) is d_kv + (b*T + t) * kv_row_stride, its value row v_offset
 * elements further.  head_dim: a multiple of 4 up to 128. */
int sl_attention_pool(const float* d_q, const float* d_kv, int64_t kv_row_stride, int64_t v_offset, int64_t B, int64_t T,
                      int64_t H, int64_t head_dim, float* d_out, void* stream);
/* (B,C,Hi,Wi) image -> (B*(Hi/P)*(Wi/P), C*P*P) patch rows, k = c*P*P + py*P + px (Conv2d weight order). */
int sl_patchify(const float* d_img, int64_t B, int64_t C, int64_t Hi, int64_t Wi, int64_t P, float* d_out,
                uint16_t* d_out_split, void* stream);
/* The same pooling with ONE QUERY PER IMAGE: d_q + b * q_batch_stride is image b's projected query (H*head_dim) — the attention
 * pool of CLIP's ResNet towers (open_clip ModifiedResNet.attnpool behind foundation_models/clip.py:52-62 `OpenClip("RN50", ...)`:
 * the query is the image's mean token).  q_batch_stride 0 = sl_attention_pool. */
int sl_attention_pool_q(const float* d_q, int64_t q_batch_stride, const float* d_kv, int64_t kv_row_stride, int64_t v_offset,
                        int64_t B, int64_t T, int64_t H, int64_t head_dim, float* d_out, void* stream);
/* Token rows of that pool from the NCHW trunk output: d_map (B,C,S) -> d_out (B, S+1, C) with
 * out[b][0] = mean_s map[b][:, s] + pos[0], out[b][1+s] = map[b][:, s] + pos[1+s]; d_pos (S+1, C). */
int sl_tokens_from_map(const float* d_map, int64_t B, int64_t C, int64_t S, const float* d_pos, float* d_out, void* stream);
/* out[g * group_stride_elems + c] = v[c] + add[c] for g < G (class-token row of every image). */
int sl_broadcast_row(const float* d_v, const float* d_add, int64_t G, int64_t group_stride_elems, int64_t N, float* d_out,
                     void* stream);
/* out[b][t][:] = table[ids[b][t]][:] + pos[t][:]  (token + positional embedding of the text tower). */
int sl_embed_tokens(const float* d_table, int64_t vocab, const int64_t* d_ids, int64_t B, int64_t T, int64_t W,
                    const float* d_pos, float* d_out, void* stream);

/* ---- K12: image preprocessing on the device -----------------------------------
 * Replaces the per-sample host transform the reference applies before encode_image
 * (foundation_models/clip.py:157-163 `self.preprocessor(image)`, i.e. open_clip's inference transform:
 * Resize(S, BICUBIC) -> CenterCrop(S) -> ToTensor -> Normalize; "squash": Resize((S,S)) without crop).
 * Bit-identical to Pillow's antialiased resize + torchvision's crop/normalise arithmetic.
 *
 * sl_preprocess_plan is a pure host function: h_hw (B,2) int32 heights/widths of the raw RGB images,
 * h_pixel_offsets (B) byte offset of each image in the packed pixel buffer (NULL: tightly packed in order);
 * writes h_plan (B, SL_PP_PLAN_STRIDE) int64 (upload it unchanged) and h_info[SL_PP_INFO_*].
 * sl_preprocess: d_pixels packed (h,w,3) uint8 images, d_plan the uploaded plan, max_h / coef_bytes from h_info,
 * h_mean/h_std 3 floats each (host); d_out (B,3,S,S) fp32 and/or d_out_u8 (B,S,S,3) resized+cropped bytes
 * (either may be NULL); d_ws of h_info[SL_PP_INFO_WS_BYTES] bytes. */
#define SL_PP_SHORTEST 0
#define SL_PP_SQUASH 1
#define SL_PP_BICUBIC 0
#define SL_PP_BILINEAR 1
#define SL_PP_PLAN_STRIDE 16
#define SL_PP_INFO_WS_BYTES 0
#define SL_PP_INFO_COEF_BYTES 1
#define SL_PP_INFO_MAX_H 2
#define SL_PP_INFO_PIXEL_BYTES 3
int sl_preprocess_plan(const int32_t* h_hw, const int64_t* h_pixel_offsets, int64_t B, int S, int resize_mode,
                       int interp, int64_t* h_plan, int64_t* h_info);
int sl_preprocess(const uint8_t* d_pixels, const int64_t* d_plan, int64_t B, int S, int interp, int64_t max_h,
                  int64_t coef_bytes, const float* h_mean, const float* h_std, float* d_out, uint8_t* d_out_u8,
                  void* d_ws, size_t ws_bytes, void* stream);

/* ---- measurement --------------------------------------------------------------------------
 * When enabled, every launch of a profiled kernel family is bracketed by HIP events on its
 * own stream.  sl_prof_read synchronises those events and returns the totals. */
#define SL_PROF_REDUCE 0 /* K1/K2 reduce kernels */
#define SL_PROF_MERGE 1  /* K3/K4 merge kernels */
#define SL_PROF_GEMM 2   /* K6 cosine GEMM */
#define SL_PROF_GATHER 3
#define SL_PROF_SCORES 4
#define SL_PROF_NFAM 5
int sl_prof_enable(int on);
int sl_prof_reset(void);
/* total_ms: sum of event-bracketed durations; launches: count; bytes: algorithmic bytes
 * (or flops for SL_PROF_GEMM) summed over those launches */
int sl_prof_read(int family, double* total_ms, int64_t* launches, double* work);

#ifdef __cplusplus
}
#endif
#endif /* SEMANTICLENS_AMD_H */
