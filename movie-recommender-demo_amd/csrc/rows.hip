// Small per-row kernels around the GEMM-shaped stages (HBM-bound, coalesced float4 traffic).
#include "../../include/amdrec.h"
#include "common.hpp"

namespace amdrec {

// faiss.normalize_L2 (faiss_retrieval.py:115, :147): x *= 1/sqrt(sum x^2) for rows with
// non-zero norm.  One wave per row, float4 lanes; out may alias in.
__global__ __launch_bounds__(256) void l2_normalize_kernel(const float* in, long long ldi, float* out,
                                                           long long ldo, long long rows, int d) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const f32x4* src = reinterpret_cast<const f32x4*>(in + row * ldi);
    f32x4* dst = reinterpret_cast<f32x4*>(out + row * ldo);
    const int d4 = d >> 2;
    float ss = 0.f;
    for (int c = lane; c < d4; c += 64) {
        f32x4 v = src[c];
        ss += sumsq4(v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float inv = ss > 0.f ? 1.0f / sqrtf(ss) : 1.0f;
    for (int c = lane; c < d4; c += 64) {
        f32x4 v = src[c];
        dst[c] = f32x4{v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv};
    }
}

// FAISSIndex.search id remap (faiss_retrieval.py:159-160): ids[i] = id_map[pos[i]];
// pos == -1 (unfilled slot) reads id_map[n-1], exactly like Python's id_map[-1].
__global__ void remap_ids_kernel(const long long* pos, const long long* id_map, long long n_map, long long* out,
                                 long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    long long p = pos[i];
    if (p < 0) p += n_map;
    out[i] = (p >= 0 && p < n_map) ? id_map[p] : -1;
}

// Request-side numerical feature prep (inference.py:186-195 / data_preprocessing.py: log1p(|x|) then
// StandardScaler.transform): out = (log1p(|x|) - mean[c]) / scale[c], float32 out (the reference's float64
// result crashes its own float32 model).
__global__ void prep_numerical_kernel(const float* x, const float* mean, const float* scale, float* out, long long rows,
                                      int cols) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int c = (int)(i % cols);
    out[i] = (log1pf(fabsf(x[i])) - mean[c]) / scale[c];
}

// Stage-2 selection (inference.py:258-263, faiss_retrieval.py:355-358): per user, the top_k of
// its k_c candidates by the ranking task's LOGIT (sigmoid is monotone; ranking on logits avoids
// the ties of saturated sigmoids), order (logit desc, candidate slot asc); then sigmoid of every
// task's logit at the winners and the winners' ad ids.  One block per user, bitonic sort of
// 64-bit (logit, ~slot) keys in LDS.
__global__ __launch_bounds__(256) void select_topk_kernel(const float* logits, long long ld, int n_tasks,
                                                          int rank_task, const long long* cand_ids, int k_c,
                                                          int top_k, long long* out_ids, float* out_scores,
                                                          int* out_slots, long long n_users) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    const long long u = blockIdx.x;
    int P = 2;
    while (P < k_c) P <<= 1;
    const float* lr = logits + (long long)rank_task * ld + u * k_c;
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        unsigned long long key = 0ull;
        if (i < k_c) {
            float v = lr[i];
            if (v == v) key = make_key(v, (uint32_t)i);     // NaN logits rank last
            else key = (unsigned long long)(~(uint32_t)i) | (1ull << 32);
        }
        keys[i] = key;
    }
    __syncthreads();
    for (int sz = 2; sz <= P; sz <<= 1)
        for (int st = sz >> 1; st > 0; st >>= 1) {
            for (int i = threadIdx.x; i < P; i += blockDim.x) {
                int j = i ^ st;
                if (j > i) {
                    unsigned long long a = keys[i], b = keys[j];
                    bool desc = (i & sz) == 0;
                    if (desc ? (a < b) : (a > b)) { keys[i] = b; keys[j] = a; }
                }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < top_k; i += blockDim.x) {
        const bool valid = i < k_c;
        const int slot = valid ? (int)key_pos(keys[i]) : -1;
        out_ids[u * top_k + i] = valid ? cand_ids[u * k_c + slot] : -1;
        if (out_slots) out_slots[u * top_k + i] = slot;
        for (int t = 0; t < n_tasks; ++t) {
            float x = valid ? logits[(long long)t * ld + u * k_c + slot] : -INFINITY;
            out_scores[((long long)t * n_users + u) * top_k + i] = 1.0f / (1.0f + expf(-x));
        }
    }
}

// The same for small top_k (<= 32: the serving path's top-10 of 500) without a sort: ONE WAVE per user holds the user's
// keys in registers (KPL per lane, k_c <= 64 * KPL) and extracts the maximum top_k times (wave max by lane exchange, the
// unique holder retires its key).  Keys are unique (the slot is part of the key), so the order is the sort's.  The
// 512-key bitonic sort above takes ~20 us of barriers whatever the batch; this takes ~3.
template <int KPL>
__global__ __launch_bounds__(256) void select_topk_small_kernel(const float* logits, long long ld, int n_tasks, int rank_task,
                                                                const long long* cand_ids, int k_c, int top_k,
                                                                long long* out_ids, float* out_scores, int* out_slots,
                                                                long long n_users) {
    const int lane = threadIdx.x & 63;
    const long long u = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (u >= n_users) return;                                  // wave-uniform
    const float* lr = logits + (long long)rank_task * ld + u * k_c;
    unsigned long long key[KPL];
#pragma unroll
    for (int j = 0; j < KPL; ++j) {
        const int i = lane + 64 * j;
        unsigned long long kk = 0ull;
        if (i < k_c) {
            const float v = lr[i];
            kk = (v == v) ? make_key(v, (uint32_t)i) : ((unsigned long long)(~(uint32_t)i) | (1ull << 32));   // NaN ranks last
        }
        key[j] = kk;
    }
    unsigned long long mine = 0ull;                            // lane r keeps winner r: the dependent loads (candidate id, the
    for (int r = 0; r < top_k; ++r) {                          // tasks' logits) go out together after the loop, one latency
        unsigned long long m = key[0];                         // instead of one per rank (they were 10 of the kernel's 15 us)
#pragma unroll
        for (int j = 1; j < KPL; ++j) m = key[j] > m ? key[j] : m;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(m, o, 64);
            m = other > m ? other : m;
        }
#pragma unroll
        for (int j = 0; j < KPL; ++j) key[j] = key[j] == m ? 0ull : key[j];      // retire the winner (0 = empty)
        if (lane == r) mine = m;
    }
    if (lane < top_k) {
        const int r = lane;
        const bool valid = r < k_c && mine != 0ull;
        const int slot = valid ? (int)key_pos(mine) : -1;
        out_ids[u * top_k + r] = valid ? cand_ids[u * k_c + slot] : -1;
        if (out_slots) out_slots[u * top_k + r] = slot;
        for (int t = 0; t < n_tasks; ++t) {
            const float x = valid ? logits[(long long)t * ld + u * k_c + slot] : -INFINITY;
            out_scores[((long long)t * n_users + u) * top_k + r] = 1.0f / (1.0f + expf(-x));
        }
    }
}

}  // namespace amdrec

using namespace amdrec;

extern "C" int amdrec_prep_numerical(const float* x, const float* mean, const float* scale, float* out, int64_t rows,
                                     int cols, void* stream) {
    REQUIRE(cols >= 1, "cols must be >= 1");
    if (rows <= 0) return AMDREC_OK;
    REQUIRE(x && mean && scale && out, "null pointer");
    const long long n = rows * cols;
    hipLaunchKernelGGL(prep_numerical_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), x, mean, scale, out, (long long)rows, cols);
    HIP_TRY(hipGetLastError());
    return AMDREC_OK;
}

extern "C" int amdrec_select_topk(const float* logits, int64_t ld_logits, int n_tasks, int rank_task,
                                  const int64_t* cand_ids, int64_t n_users, int k_c, int top_k, int64_t* out_ids,
                                  float* out_scores, int32_t* out_slots, void* stream) {
    REQUIRE(n_tasks >= 1 && rank_task >= 0 && rank_task < n_tasks, "bad task index");
    REQUIRE(k_c >= 1 && k_c <= AMDREC_MAX_K, "candidates per user must be in [1,%d]", AMDREC_MAX_K);
    REQUIRE(top_k >= 1 && top_k <= AMDREC_MAX_K, "top_k out of range");
    if (n_users <= 0) return AMDREC_OK;
    REQUIRE(logits && cand_ids && out_ids && out_scores, "null pointer");
    REQUIRE(ld_logits >= n_users * k_c, "ld_logits too small");
    if (top_k <= 32 && k_c <= 512) {                            // one wave per user, no sort
        const dim3 grid((unsigned)((n_users + 3) / 4));
        hipStream_t st = reinterpret_cast<hipStream_t>(stream);
        if (k_c <= 128)
            hipLaunchKernelGGL(select_topk_small_kernel<2>, grid, dim3(256), 0, st, logits, (long long)ld_logits, n_tasks, rank_task,
                               (const long long*)cand_ids, k_c, top_k, (long long*)out_ids, out_scores, out_slots, (long long)n_users);
        else
            hipLaunchKernelGGL(select_topk_small_kernel<8>, grid, dim3(256), 0, st, logits, (long long)ld_logits, n_tasks, rank_task,
                               (const long long*)cand_ids, k_c, top_k, (long long*)out_ids, out_scores, out_slots, (long long)n_users);
        HIP_TRY(hipGetLastError());
        return AMDREC_OK;
    }
    int P = 2;
    while (P < k_c) P <<= 1;
    hipLaunchKernelGGL(select_topk_kernel, dim3((unsigned)n_users), dim3(256), (size_t)P * 8,
                       reinterpret_cast<hipStream_t>(stream), logits, (long long)ld_logits, n_tasks, rank_task,
                       (const long long*)cand_ids, k_c, top_k, (long long*)out_ids, out_scores, out_slots,
                       (long long)n_users);
    HIP_TRY(hipGetLastError());
    return AMDREC_OK;
}

extern "C" int amdrec_l2_normalize(const float* in, int64_t ld_in, float* out, int64_t ld_out, int64_t rows,
                                   int dim, void* stream) {
    REQUIRE(dim >= 4 && dim % 4 == 0, "dim=%d must be a positive multiple of 4", dim);
    if (rows <= 0) return AMDREC_OK;
    REQUIRE(ld_in % 4 == 0 && ld_out % 4 == 0 && ld_in >= dim && ld_out >= dim, "bad leading dimension");
    REQUIRE(in && out, "null pointer");
    REQUIRE(((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0, "pointers must be 16-byte aligned");
    hipLaunchKernelGGL(l2_normalize_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), in, (long long)ld_in, out, (long long)ld_out,
                       (long long)rows, dim);
    HIP_TRY(hipGetLastError());
    return AMDREC_OK;
}

extern "C" int amdrec_remap_ids(const int64_t* pos, const int64_t* id_map, int64_t n_map, int64_t* out,
                                int64_t n, void* stream) {
    if (n <= 0) return AMDREC_OK;
    REQUIRE(pos && out && (id_map || n_map == 0), "null pointer");
    hipLaunchKernelGGL(remap_ids_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), (const long long*)pos, (const long long*)id_map,
                       (long long)n_map, (long long*)out, (long long)n);
    HIP_TRY(hipGetLastError());
    return AMDREC_OK;
}
