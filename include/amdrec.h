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

#define AMDREC_ABI_VERSION 1
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

/* faiss.normalize_L2 (faiss_retrieval.py:115, :147): every row scaled by 1/||row||_2, rows of
 * zero norm left as they are.  out may alias in.  dim % 4 == 0. */
int amdrec_l2_normalize(const float* in, int64_t ld_in, float* out, int64_t ld_out, int64_t rows,
                        int dim, void* stream);

/* FAISSIndex.search's id remap (faiss_retrieval.py:159-160, a Python list comprehension):
 * out[i] = id_map[pos[i]]; pos == -1 reads id_map[n_map-1] exactly like Python's id_map[-1]. */
int amdrec_remap_ids(const int64_t* pos, const int64_t* id_map, int64_t n_map, int64_t* out,
                     int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AMDREC_H */
