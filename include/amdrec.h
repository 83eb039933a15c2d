/* libamdrec - C ABI of the MI355X-native retrieval + ranking hot path.
 *
 * The reference (saitejasrivilli/movie-recommender-demo) is pure Python and defines no FFI:
 * its boundary is the Python class surface (two_tower_model.py, transformer_ranker.py,
 * faiss_retrieval.py, inference.py) plus the faiss / ATen calls behind it.  Each entry point
 * below replaces one of those third-party call sites (cited per function); the Python
 * drop-ins in movie-recommender-demo_amd/amdrec bind them with ctypes.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer on the current HIP device unless marked "host";
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued
 *    asynchronously on it, nothing synchronises with the host, nothing allocates: scratch
 *    memory is a caller-provided, 256-byte aligned `workspace` sized by the matching
 *    *_workspace() query;
 *  - return value: 0 = ok, < 0 = error (AMDREC_E*); amdrec_last_error() returns the message
 *    of the calling thread's last error;
 *  - matrices are row-major float32; `ld*` = leading dimension in elements.
 */
#ifndef AMDREC_H
#define AMDREC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMDREC_ABI_VERSION 10
#define AMDREC_MAX_K 2048

int amdrec_abi_version(void);
const char* amdrec_last_error(void);

/* ---- retrieval: exact inner-product top-k over a flat corpus ---------------------------
 * Replaces faiss IndexFlatIP.search as called by FAISSIndex.search (faiss_retrieval.py:155):
 * scores = queries . corpus^T, k largest per query, sorted descending (ties: lower position
 * first).  out_pos = position in the corpus + pos_offset (the shard's first global row),
 * unfilled slots (k > nrows) = -1 / -inf like faiss.  Rows are expected L2-normalised by the
 * caller (amdrec_l2_normalize), as FAISSIndex.add/search do (:114-115, :146-147).
 * n_fixup (device int, may be NULL) receives the number of queries that took the slow
 * exact fix-up path.  dim % 4 == 0, dim <= 2048, 1 <= k <= AMDREC_MAX_K. */
int amdrec_flat_search_workspace(int64_t nq, int64_t nrows, int k, size_t* bytes /*host*/);
int amdrec_flat_search(const float* corpus, int64_t nrows, int64_t ld_corpus, int dim,
                       const float* queries, int64_t nq, int64_t ld_queries, int k,
                       int64_t pos_offset, float* out_scores /*[nq][k]*/,
                       int64_t* out_pos /*[nq][k]*/, void* workspace, size_t workspace_bytes,
                       int* n_fixup, void* stream);

/* Mixed-precision form of the same search (same result contract: the exact fp32 top-k).  The sample and filter
 * passes read `corpus_bf16`, a bf16 (round-to-nearest) copy of the corpus made by amdrec_bf16_rows, with the bf16
 * MFMA; candidates are re-scored in fp32 from `corpus` and the result is certified against the error bound
 * eps = ||dq|| (M + D) + ||q|| D + 2 dim 2^-24 ||q|| (M + D),  dq = bf16(q) - q,  M = max_norm[0] = the largest row norm
 * and D = max_norm[1] = the largest row rounding-error norm ||x - bf16(x)|| of the corpus (max_norm: device float[2],
 * accumulated by amdrec_bf16_rows; at worst eps = (2^-7 + 2^-16) ||q|| M, the both-operands-half-an-ulp-off case);
 * uncertified queries take the exact fp32 fix-up scan.  dim % 8 == 0.
 * Turns the filter pass from fp32-MFMA-bound into memory-bound (half the bytes, 16x the MFMA rate). */
int amdrec_bf16_rows(const float* x, int64_t rows, int64_t ld, int dim, uint16_t* out /*[rows][ld_out] bf16*/,
                     int64_t ld_out, float* max_norm /*device float[2], in/out (atomic max), may be NULL*/, void* stream);
int amdrec_flat_search_mixed_workspace(int64_t nq, int64_t nrows, int k, int dim, size_t* bytes /*host*/);
int amdrec_flat_search_mixed(const float* corpus, int64_t nrows, int64_t ld_corpus, int dim,
                             const uint16_t* corpus_bf16, int64_t ld_bf16, const float* max_norm,
                             const float* queries, int64_t nq, int64_t ld_queries, int k,
                             int64_t pos_offset, float* out_scores /*[nq][k]*/,
                             int64_t* out_pos /*[nq][k]*/, void* workspace, size_t workspace_bytes,
                             int* n_fixup, void* stream);

/* ---- retrieval: IVF-Flat (faiss IndexIVFFlat, METRIC_INNER_PRODUCT, IndexFlatIP quantizer:
 * faiss_retrieval.py:50-55, searched at :150-155 with index.nprobe = nprobe) ---------------------
 * Layout: corpus rows stored list-contiguous `lists[N][ld]` (list l = rows [list_off[l], list_off[l+1]))
 * with row_pos[r] = position of stored row r in insertion order.  A search is
 *   1. amdrec_flat_search(centroids, nlist, ..., k = nprobe)  -> probes[nq][nprobe]  (-1 = none)
 *   2. pool_base[q][p] = exclusive prefix sum of the probed lists' lengths (host-side plumbing)
 *   3. amdrec_ivf_scan   -> pool_keys[q][pool_base[q][p] + i] = key(score, row_pos + pos_offset)
 *   4. amdrec_ivf_select -> exact k best of each query's pool, (score desc, position asc).
 * The scan is exact over the probed lists; which lists are probed depends on the trained centroids. */
int amdrec_ivf_scan(const float* lists, int64_t ld, int dim, const int64_t* row_pos,
                    const int64_t* list_off, const float* queries, int64_t nq, int64_t ld_queries,
                    const int64_t* probes, const int64_t* pool_base, int nprobe, uint64_t* pool_keys,
                    int64_t pool_ld, int64_t pos_offset, void* stream);
/* Batched form of step 3: the (query, probe) pairs sorted by list (pair_query / pair_probe, group_off
 * [nlist+1]); qtile_prefix[nlist+1] = prefix sum of ceil(group size / qtile); qtile_bound >= qtile_prefix[nlist]
 * (a host-side upper bound, e.g. pairs/qtile + nlist, <= 65535).  Every list is read once per qtile-query group
 * by an fp32-MFMA GEMM tile instead of once per probing query; qtile = 64, or 32 when the groups are sparse (few probing
 * queries per list) - the value given to amdrec_ivf_group.  Same pool layout and keys as amdrec_ivf_scan. */
int amdrec_ivf_scan_grouped(const float* lists, int64_t ld, int dim, const int64_t* row_pos,
                            const int64_t* list_off, int nlist, int64_t max_list_rows, const float* queries,
                            int64_t ld_queries, const int64_t* group_off, const int64_t* qtile_prefix,
                            int64_t qtile_bound, int qtile, const int64_t* pair_query, const int64_t* pair_probe,
                            const int64_t* pool_base, int nprobe, uint64_t* pool_keys, int64_t pool_ld,
                            int64_t pos_offset, const float* tau /*or NULL*/, int64_t ld_tau,
                            int64_t* pool_fill /*[nq] or NULL*/, void* stream);
/* Filter mode of the grouped scan (tau != NULL): a row's key is kept only if its score >= tau[q * ld_tau], appended to
 * query q's pool row at pool_fill[q]++ (seeded by the caller with the keys already there; pool_base is not used).
 * With tau[q] = the k-th score over a SUBSET of the probed lists (a lower bound of the final k-th score) the union of
 * that subset's keys and the kept rows of the remaining lists contains the exact top-k: amdrec.ivf scans the nearest
 * eighth of the probes unfiltered, selects, and scans the rest with this filter - the pool shrinks ~5x. */
/* The same filter mode with a bf16 PREFILTER (ABI v10; same call site, faiss_retrieval.py:150-155): the (list, query tile)
 * GEMMs run on `lists_bf16` / `queries_bf16` (round-to-nearest copies made by amdrec_bf16_rows; the lists' copy also yields
 * max_norm[2] = {largest row norm, largest row rounding-error norm}) with the bf16 MFMA and only nominate rows: a row whose
 * bf16 score is >= tau_lo[q] is re-scored in fp32 from `lists` / `queries` and appended to query q's pool row at
 * pool_fill[q]++ if that exact score passes tau[q * ld_tau].  With tau_lo[q] = tau[q] - eps_q (amdrec_ivf_filter_bounds: the
 * flat search's error bound, see amdrec_flat_search_mixed) every row with fp32 score >= tau is nominated, so the pool
 * receives the same rows as amdrec_ivf_scan_grouped's filter mode, with fp32 keys.  dim % 8 == 0. */
int amdrec_ivf_filter_bounds(const float* queries, int64_t nq, int64_t ld_queries, int dim,
                             const uint16_t* queries_bf16, int64_t ld_queries_bf16, const float* max_norm /*device float[2]*/,
                             const float* tau, int64_t ld_tau, float* tau_lo /*[nq]*/, void* stream);
int amdrec_ivf_scan_grouped_mixed(const float* lists, int64_t ld, const uint16_t* lists_bf16, int64_t ld_bf16, int dim,
                                  const int64_t* row_pos, const int64_t* list_off, int nlist, int64_t max_list_rows,
                                  const float* queries, int64_t ld_queries, const uint16_t* queries_bf16,
                                  int64_t ld_queries_bf16, const int64_t* group_off, const int64_t* qtile_prefix,
                                  int64_t qtile_bound, int qtile, const int64_t* pair_query, uint64_t* pool_keys,
                                  int64_t pool_ld, int64_t pos_offset, const float* tau, int64_t ld_tau,
                                  const float* tau_lo /*[nq]*/, int64_t* pool_fill /*[nq]*/, void* stream);
/* Step 1 as a dense key table (replaces the `index.search` of faiss's IndexFlatIP quantizer inside IndexIVFFlat.search,
 * faiss_retrieval.py:50-55, :150-155): keys[q][c] = 64-bit (score of query q against centroid c, ~c) for every centroid -
 * the pool format of amdrec_ivf_select, which then yields the nprobe best centroids per query (score desc, lower centroid
 * first).  Same fp32-MFMA scores as amdrec_flat_search over the centroid table, without its candidate machinery.
 * keys: 16-byte aligned; ld_keys even and >= nlist (an odd nlist's last pair writes one padding key). */
int amdrec_ivf_coarse_keys(const float* centroids, int nlist, int64_t ld_centroids, int dim, const float* queries,
                           int64_t nq, int64_t ld_queries, uint64_t* keys /*[nq][ld_keys]*/, int64_t ld_keys, void* stream);
int amdrec_ivf_select(const uint64_t* pool_keys, int64_t pool_ld, const int64_t* pool_count /*[nq]*/,
                      int64_t nq, int k, float* out_scores /*[nq][k]*/, int64_t* out_pos /*[nq][k]*/,
                      void* stream);
/* The same selection for FEW queries with LARGE pools (one request against nlist 100 / nprobe 10 at 1M ads: 100 000 keys):
 * `slices` workgroups per query select the k best of a slice of the pool each into workspace[nq][slices][k] (64-bit keys:
 * nq * slices * k * 8 bytes), the last one to finish (a ticket per query: tickets[nq], int32, zero on entry, zero again
 * on return) selects the k best of those.  Same result as amdrec_ivf_select. */
int amdrec_ivf_select_split(const uint64_t* pool_keys, int64_t pool_ld, const int64_t* pool_count /*[nq]*/, int64_t nq, int k,
                            int slices, float* out_scores, int64_t* out_pos, void* workspace, size_t workspace_bytes,
                            int32_t* tickets, void* stream);
/* Step 2 as kernels (no host sync, capturable): from probes[nq][nprobe] and the list lengths, the pool layout
 * (pool_base[nq][nprobe], pool_count[nq]) and the (query, probe) pairs grouped by list for amdrec_ivf_scan_grouped
 * (pair_query / pair_probe [nq*nprobe], group_off / qtile_prefix [nlist+1]; order inside a group is unspecified - it
 * does not influence any result).  workspace: >= 4*(nlist+1) + 4*nq*nprobe + 512 bytes. */
int amdrec_ivf_group(const int64_t* probes /*[nq][ld_probes], the first nprobe columns*/, int64_t ld_probes, int64_t nq,
                     int nprobe, int nlist, const int64_t* list_len /*[nlist]*/,
                     int64_t* pool_base, int64_t* pool_count, int64_t* pair_query, int64_t* pair_probe,
                     int64_t* group_off, int64_t* qtile_prefix, int qtile /*32 or 64*/, void* workspace,
                     size_t workspace_bytes, void* stream);

/* Index build (faiss IndexIVFFlat.train / .add behind FAISSIndex.train / .add, faiss_retrieval.py:83-95, :118).
 * amdrec_ivf_assign: assign[r] = arg max_c <x[r], centroids[c]> (the IndexFlatIP quantizer; ties -> lower c;
 * fp32 MFMA GEMM with an arg-max epilogue), best_score optional.  workspace: >= 8*rows + 256 bytes.
 * amdrec_ivf_kmeans_step: one spherical Lloyd iteration in place: assign, per-centroid sum (64-bit fixed-point integer
 * atomics: order-independent, so training is bit-reproducible), centroid = sum / |sum| (empty clusters keep theirs). */
int amdrec_ivf_assign(const float* x, int64_t rows, int64_t ld, int dim, const float* centroids, int nlist,
                      int64_t ld_centroids, int64_t* assign /*[rows]*/, float* best_score /*[rows] or NULL*/,
                      void* workspace, size_t workspace_bytes, void* stream);
int amdrec_ivf_kmeans_workspace(int64_t rows, int dim, int nlist, size_t* bytes /*host*/);
int amdrec_ivf_kmeans_step(const float* x, int64_t rows, int64_t ld, int dim, float* centroids /*in/out*/, int nlist,
                           int64_t ld_centroids, void* workspace, size_t workspace_bytes, void* stream);

/* Cross-shard merge (absent in the single-device reference): for queries
 * [q0, q0+nq) merge n_lists per-shard top-k lists (scores/positions of list g start
 * g*list_stride_bytes after the base pointers; each list is [nq_total][k]; positions are global int32
 * (the exchange format: 8 bytes per candidate on the wire), -1 = unfilled) into the global top-k, same
 * order rule as amdrec_flat_search.  n_lists*k <= 16384. */
int amdrec_topk_merge(const float* scores, const int32_t* pos, int n_lists, int64_t list_stride_bytes,
                      int64_t q0, int64_t nq, int k, float* out_scores /*[nq][k]*/,
                      int64_t* out_pos /*[nq][k]*/, void* stream);
/* Same with SHORT lists: every shard sends only its best list_k <= k rows per query (each list is [nq_total][list_k]),
 * which cuts a shard's exact re-scoring and the wire bytes by k / list_k.  The merged top-k is the exact global top-k iff
 * no shard was cut off above the merged k-th score; *n_inexact (device int32, NOT reset: accumulates) is incremented for
 * every query where that cannot be shown - a full list whose last score reaches the merged k-th score (ties included),
 * or fewer than k merged entries while some list is full.  The caller repeats such a batch with list_k = k
 * (amdrec.sharded.ShardedRecommender does).  n_lists*list_k <= 16384. */
int amdrec_topk_merge_partial(const float* scores, const int32_t* pos, int n_lists, int list_k,
                              int64_t list_stride_bytes, int64_t q0, int64_t nq, int k,
                              float* out_scores /*[nq][k]*/, int64_t* out_pos /*[nq][k]*/,
                              int32_t* n_inexact /*device*/, void* stream);

/* ---- two-tower encoders (eval mode) ------------------------------------------------------
 * Replaces the ATen call chain of UserTower.forward / AdTower.forward
 * (two_tower_model.py:98-121, :167-184): per-column embedding lookup + concat
 * (EmbeddingLayer.forward :33-49) [+ numerical tail :113] -> (Linear, BatchNorm1d(eval), ReLU,
 * Dropout(identity)) x n -> Linear -> F.normalize(p=2, dim=1).
 * BatchNorm is folded into the preceding Linear by the host (exact in eval mode):
 *   W' = W * g / sqrt(var + eps),  b' = (b - mean) * g / sqrt(var + eps) + beta.
 * Weights are [out][ldw] row-major with K zero-padded to ldw (multiple of 32). */
#define AMDREC_MAX_LAYERS 8
#define AMDREC_MAX_TASKS 4

typedef struct {
    int32_t n_feat;              /* categorical columns (6 user / 20 ad) */
    int32_t emb_dim;             /* 16; power of two >= 4 */
    int32_t n_num;               /* numerical tail width (13 user / 0 ad) */
    int32_t n_layers;            /* Linear layers incl. the output layer */
    int32_t dims[AMDREC_MAX_LAYERS + 1]; /* dims[0] = n_feat*emb_dim + n_num, dims[l+1] = width of layer l */
    int32_t ldw[AMDREC_MAX_LAYERS];
    const float* tables;         /* all embedding tables back to back [sum(card)][emb_dim] */
    const int32_t* table_off;    /* device [n_feat]: first row of column f */
    const int32_t* cards;        /* device [n_feat]: cardinality of column f */
    const float* w[AMDREC_MAX_LAYERS];
    const float* b[AMDREC_MAX_LAYERS];
    /* != 0: amdrec_l2_normalize applied to the output rows in the same call - the rows go to an inner-product search
     * that normalises its queries (faiss.normalize_L2 on the tower's already normalised output, faiss_retrieval.py:147
     * after two_tower_model.py:121): the same two roundings as the two calls, one launch fewer. */
    int32_t renormalize;
} amdrec_tower_params;

int amdrec_tower_workspace(const amdrec_tower_params* p /*host*/, int64_t rows, size_t* bytes /*host*/);
/* cat int64 [rows][n_feat] (the reference casts .long()), num float32 [rows][n_num] or NULL,
 * out float32 [rows][ld_out] unit rows.  bad_index_flag (device int, may be NULL) is set to 1 if
 * any categorical index is outside [0, card) - torch raises IndexError there; the kernels clamp
 * so nothing is read out of bounds. */
int amdrec_tower_forward(const amdrec_tower_params* p /*host*/, const int64_t* cat, const float* num,
                         int64_t rows, float* out, int64_t ld_out, int* bad_index_flag,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ---- TransformerRanker.forward (eval mode) -> logits --------------------------------------
 * Replaces the ATen call chain of transformer_ranker.py:332-380.  With seq_len == 1 (:358) the
 * 8-head attention is exactly W_o(W_v x + b_v) + b_o (softmax over one key == 1), so W_q / W_k
 * are not parameters here.  pos[0] is folded into b_proj and cross weights are passed
 * transposed ([out][in]) by the host. */
typedef struct {
    /* [d_model][ldw_dm].  w_v may be NULL: then w_o / b_o hold the pre-multiplied W_o*W_v and
     * W_o*b_v + b_o (exact algebra at seq_len 1; only the fp32 rounding order differs) and the
     * attention block is ONE GEMM. */
    const float *w_v, *b_v, *w_o, *b_o;
    const float *ln1_g, *ln1_b;
    const float *w_1, *b_1;               /* [d_ff][ldw_dm] */
    const float *w_2, *b_2;               /* [d_model][ldw_ff] */
    const float *ln2_g, *ln2_b;
    int32_t ldw_dm, ldw_ff;
    /* Optional host-split planes of w_o / w_1 / w_2 for the error-compensated bf16-MFMA GEMM ("x6", used for passes of
     * more than 8192 rows; NULL -> fp32 MFMA).  Layout [out][ldw/16][3][16] bf16: for every 16-column group the planes
     * h, m, l of the exact 3-way truncation split w = h + m + l (h = w with the low 16 bits cleared, m = (w - h)
     * likewise, l = w - h - m; each plane stored as the high 16 bits of that fp32 value).  Same result contract as the
     * fp32 path: error at the level of an fp32 fma chain. */
    const uint16_t *w_o_x6, *w_1_x6, *w_2_x6;
} amdrec_encoder_layer;

/* Weights of the fp16x3 "row-owner" engine (csrc/rowowner.hpp; optional: stream == NULL -> engine off).  The engine
 * runs everything after the feature projection - the encoder layers (transformer_ranker.py:136-155 with the seq-1
 * attention :59-88), the cross layers (:199-202) and the heads (:375-378) - in ONE kernel with the activations in
 * registers; every fp32 operand is split into two fp16 planes (after an exact power-of-two scaling) and a product is
 * three fp16 MFMAs accumulated in fp32: fp32 in, fp32 out, error at the level of an fp32 fma chain.
 * `stream` = all weight matrices as 1 KB MFMA fragment sets in consumption order (the packing, incl. the k permutation
 * inside a 16-wide k-step, is defined by amdrec/weights.py pack_x3_stream and mirrored by rowowner.hpp); `sw_*` = the
 * power-of-two scale of each packed matrix; `hn*` / `hb*` = 16 * max_j ||w_j||_2 and max_j |b_j| of the first matrix of
 * each two-stage block (FFN per layer, heads): the kernel bounds a row's hidden activations by hn * max|x| + hb to scale them.
 * Requires d_model == 256, d_ff % 32 == 0, head_h1 % 32 == 0, head_h2 == 64 and the pre-multiplied attention
 * (w_v == NULL); used for passes of at least `min_rows` rows (0 = default 8193). */
typedef struct {
    const void* stream;
    int64_t chunks;              /* stream length in 16 KB chunks */
    int64_t min_rows;
    const float* params;         /* every bias / LayerNorm weight / head vector of the chain as one device blob (layout:
                                  * csrc/ranker_x3.hip x3_param_floats; amdrec/weights.py pack_x3_params), copied into LDS
                                  * once per workgroup */
    int64_t n_params;            /* floats, padded to a multiple of 1024; must fit 44 KB of LDS */
    int64_t variant;             /* 32: rowowner.hpp (32 rows per wave, one wave per SIMD); 16: rowowner16.hpp (16 rows per
                                  * wave, two waves per SIMD); the stream is packed for exactly one of them */
    float sw_ov[AMDREC_MAX_LAYERS], sw_1[AMDREC_MAX_LAYERS], sw_2[AMDREC_MAX_LAYERS];
    float hn[AMDREC_MAX_LAYERS], hb[AMDREC_MAX_LAYERS];
    float sw_cross[AMDREC_MAX_LAYERS];
    float sw_h1, sw_h2, hn_head, hb_head;
    /* Optional second packing of the SAME fragment sets for the column-split kernel (csrc/rowowner16c.hpp; variant 16
     * only): a workgroup of four waves owns 16 rows and the waves split each layer's output features, so that one
     * request's 500 candidates (inference.py:241-250) spread over 32 CUs instead of 8 - results bit-identical to the
     * 16-row kernel.  Every 16 KB chunk holds one group of four fragment sets per wave (amdrec/weights.py x3c_stream_*).
     * Used for passes of at most cs_max_rows rows (0 = default 4096, < 0 = never); NULL -> off. */
    const void* stream_cs;
    int64_t chunks_cs;
    int64_t cs_max_rows;
} amdrec_x3_weights;

typedef struct {
    int32_t n_user_feat, n_ad_feat, emb_dim, n_num;
    int32_t d_model, d_ff, n_layers, n_cross, n_tasks, head_h1, head_h2;
    float ln_eps;
    int32_t ldw_proj, ldw_cross, ldw_head1, ldw_head2;
    const float* tables;            /* user tables then ad tables, [sum(card)][emb_dim] */
    const int32_t* table_off;       /* device [n_user_feat + n_ad_feat] */
    const int32_t* cards;           /* device [n_user_feat + n_ad_feat] */
    const float* w_proj;            /* [d_model][ldw_proj], columns = [user emb | ad emb | numerical] */
    const float* b_proj;            /* bias + positional_encoding[0,0,:] */
    /* Optional split of w_proj for the broadcast form (user_rowdiv > 1): the [user emb | numerical] columns
     * and the [ad emb] columns as two matrices.  When both are set the user part is computed once per USER
     * row and added to each of its candidates' ad part (transformer_ranker.py:355 is linear in its input). */
    const float* w_proj_user;       /* [d_model][ldw_proj_user] or NULL */
    const float* w_proj_ad;         /* [d_model][ldw_proj_ad]   or NULL */
    int32_t ldw_proj_user, ldw_proj_ad;
    amdrec_encoder_layer layers[AMDREC_MAX_LAYERS];
    const float* cross_wt[AMDREC_MAX_LAYERS];  /* cross_weights[i]^T : [out][ldw_cross] */
    const float* cross_b[AMDREC_MAX_LAYERS];
    const float* head_w1;           /* first Linear of all heads stacked: [n_tasks*head_h1][ldw_head1] */
    const float* head_b1;
    const float* head_w2[AMDREC_MAX_TASKS];   /* [head_h2][ldw_head2] */
    const float* head_b2[AMDREC_MAX_TASKS];
    const float* head_w3[AMDREC_MAX_TASKS];   /* [head_h2] */
    const float* head_b3[AMDREC_MAX_TASKS];   /* [1] */
    /* Optional candidate-side cache for the broadcast form: ad_proj_cache[a] = w_proj_ad . emb(ad_cat[a]) for every
     * row of the resident ad-feature table (amdrec_ranker_project_ads; like the ad-tower embeddings it depends on
     * the ad and the weights only, so it is computed once per index build).  When set (with w_proj_user/w_proj_ad)
     * the ad half of the projection GEMM becomes a row gather: x0[r] = ad_proj_cache[ad row of r] + U[user of r],
     * the same two addends in the same order as the uncached path (bit-identical). */
    const float* ad_proj_cache;     /* [n_ad_rows][ld_ad_proj_cache] or NULL */
    int64_t ld_ad_proj_cache;
    /* Optional x6 planes (see amdrec_encoder_layer) of cross_wt[i] and head_w1. */
    const uint16_t* cross_wt_x6[AMDREC_MAX_LAYERS];
    const uint16_t* head_w1_x6;
    amdrec_x3_weights x3;
} amdrec_ranker_params;

int amdrec_ranker_workspace(const amdrec_ranker_params* p /*host*/, int64_t rows, size_t* bytes /*host*/);
/* Row r of the batch uses user_cat/numerical row r / user_rowdiv (user_rowdiv = 1: one feature
 * row per batch row as in forward(); = stage1_k: one user broadcast over its candidates, replacing
 * Tensor.repeat at inference.py:241-242) and ad_cat row (ad_rowmap ? ad_rowmap[r] : r)
 * (ad_rowmap = candidate ids into a resident ad-feature table: the lookup the reference stubs
 * out at inference.py:244-248).  out_logits[t * ld_logits + r], t = ctr, engagement, revenue.
 * n_user_rows / n_ad_rows = rows of user_cat / ad_cat (for index validation only). */
int amdrec_ranker_forward(const amdrec_ranker_params* p /*host*/, const int64_t* user_cat,
                          const float* numerical, int64_t user_rowdiv, const int64_t* ad_cat,
                          const int64_t* ad_rowmap, int64_t rows, float* out_logits, int64_t ld_logits,
                          int* bad_index_flag, int64_t n_user_rows, int64_t n_ad_rows, void* workspace,
                          size_t workspace_bytes, void* stream);

/* Test / debugging entry of the fp16x3 engine: run the first n_phases phases of the chain (phase order: per encoder
 * layer {attention + LN1, FFN + LN2}, then the cross layers, then the heads; n_phases < 0 = all) on dense projected
 * rows X [rows][ldx] and return the rows after the last executed phase in x_out [rows][ld_out] (may be NULL) and, when
 * the heads ran, the logits.  workspace: >= ceil(rows / 128) * 128 * 1024 bytes.  Lets the parity tests localise an
 * error to one phase; amdrec_ranker_forward uses the same kernel for whole passes. */
int amdrec_ranker_x3_prefix(const amdrec_ranker_params* p /*host*/, const float* X, int64_t ldx, int64_t rows,
                            int n_phases, float* x_out, int64_t ld_out, float* logits, int64_t ld_logits,
                            void* workspace, size_t workspace_bytes, void* stream);

/* Fills the candidate-side cache described at amdrec_ranker_params.ad_proj_cache: out[a] = w_proj_ad . emb(ad_cat[a])
 * (no bias; the user half carries it), a = 0..n_ads-1.  workspace: >= 4*d_model + 256 bytes. */
int amdrec_ranker_project_ads(const amdrec_ranker_params* p /*host*/, const int64_t* ad_cat, int64_t n_ads,
                              float* out /*[n_ads][ld_out]*/, int64_t ld_out, void* workspace,
                              size_t workspace_bytes, void* stream);

/* faiss.normalize_L2 (faiss_retrieval.py:115, :147): every row scaled by 1/||row||_2, rows of
 * zero norm left as they are.  out may alias in.  dim % 4 == 0. */
int amdrec_l2_normalize(const float* in, int64_t ld_in, float* out, int64_t ld_out, int64_t rows,
                        int dim, void* stream);

/* FAISSIndex.search's id remap (faiss_retrieval.py:159-160, a Python list comprehension):
 * out[i] = id_map[pos[i]]; pos == -1 reads id_map[n_map-1] exactly like Python's id_map[-1]. */
int amdrec_remap_ids(const int64_t* pos, const int64_t* id_map, int64_t n_map, int64_t* out,
                     int64_t n, void* stream);

/* Optional per-launch timing: HIP events recorded on the launch stream around every GEMM-shaped
 * kernel launch, accumulated per kernel tag ("<epilogue>_<BP>x<BQ>", DESIGN.md maps tags to
 * symbol names).  flops / bytes are the ALGORITHMIC 2*M*N*K and operand+result bytes of the
 * launches.  amdrec_profile_report synchronises with the recorded events. */
typedef struct {
    char name[64];
    int64_t launches;
    double total_ms, flops, bytes;
} amdrec_profile_entry;
int amdrec_profile_enable(int on);   /* also clears the counters */
/* Time only the launches whose tag starts with tag_prefix (NULL or "" = all): an event pair costs the stream ~10 us of
 * idle GPU per launch, so a benchmark times every kernel in a pre-pass and only its dominant one in the timed region. */
int amdrec_profile_only(const char* tag_prefix);
int amdrec_profile_report(amdrec_profile_entry* out /*host*/, int max_entries, int* n /*host*/);

/* Request-side numerical prep (inference.py:186-195): out[r][c] = (log1p(|x[r][c]|) - mean[c]) / scale[c],
 * float32 (the reference produces float64 there and crashes its own float32 model). out may alias x. */
int amdrec_prep_numerical(const float* x, const float* mean /*[cols]*/, const float* scale /*[cols]*/,
                          float* out, int64_t rows, int cols, void* stream);

/* Stage-2 selection (inference.py:258-263: sigmoid -> np.argsort(ctr)[::-1][:top_k]): per user
 * the top_k of its k_c candidates by the LOGIT of task `rank_task` (sigmoid is monotone), order
 * (logit desc, candidate slot asc).  out_ids[u][i] = cand_ids[u][slot], out_scores[t][u][i] =
 * sigmoid(logits[t][u*k_c + slot]), out_slots (optional) = the winning slots. */
int amdrec_select_topk(const float* logits /*[n_tasks][ld]*/, int64_t ld_logits, int n_tasks,
                       int rank_task, const int64_t* cand_ids /*[n_users][k_c]*/, int64_t n_users,
                       int k_c, int top_k, int64_t* out_ids /*[n_users][top_k]*/,
                       float* out_scores /*[n_tasks][n_users][top_k]*/, int32_t* out_slots, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AMDREC_H */
