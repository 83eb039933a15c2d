// IVF-Flat search (replaces faiss IndexIVFFlat(METRIC_INNER_PRODUCT).search behind FAISSIndex.search,
// faiss_retrieval.py:50-55, :150-155).  The coarse step (nprobe best centroids by inner product)
// reuses amdrec_flat_search on the centroid table; here:
//   ivf_scan   : one workgroup per (query, probed list): inner product of the query with every row
//                of the list (rows stored list-contiguous), written as 64-bit (score, ~position)
//                keys into the query's pool.  Wave-per-row float4 streaming: HBM/L2-bound.
//   ivf_select : per query, exact k largest keys of its pool: 64-bit LDS-histogram radix select
//                (6 passes), compaction, bitonic sort.  Keys are unique (lists are disjoint), so the
//                selection is exact and deterministic for every input - no sampling, no fix-up.
#include "gemm_core.hpp"
#include "topk_utils.hpp"
#include "../../include/amdrec.h"

namespace amdrec {

__global__ __launch_bounds__(256) void ivf_scan_kernel(const float* xs, long long ld, int d, const long long* spos,
                                                       const long long* list_off, const float* Q, long long ldq,
                                                       const long long* probes, const long long* base, int nprobe,
                                                       unsigned long long* keys, long long pool_ld,
                                                       long long pos_offset) {
    __shared__ __attribute__((aligned(16))) float qv[2048];
    const int q = blockIdx.y, p = blockIdx.x;
    const long long l = probes[(long long)q * nprobe + p];
    if (l < 0) return;
    const long long r0 = list_off[l], r1 = list_off[l + 1];
    if (r1 <= r0) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < d; i += 256) qv[i] = Q[(long long)q * ldq + i];
    __syncthreads();
    unsigned long long* dst = keys + (long long)q * pool_ld + base[(long long)q * nprobe + p];
    const int d4 = d >> 2;
    constexpr int NW = 4, U = 8;
    for (long long row0 = r0; row0 < r1; row0 += NW * U) {
        float part[U];
        const f32x4* xr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            long long r = row0 + w * U + u;
            r = r < r1 ? r : r1 - 1;                           // clamped: branch-free loads, result dropped below
            xr[u] = reinterpret_cast<const f32x4*>(xs + r * ld);
            part[u] = 0.f;
        }
        // column chunk outermost so that the U row loads of a chunk are in flight together (with the row loop
        // outside, each row's load had to return before the next row's was issued)
        for (int c = lane; c < d4; c += 64) {
            const f32x4 y = *reinterpret_cast<const f32x4*>(&qv[4 * c]);
            f32x4 x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = xr[u][c];
#pragma unroll
            for (int u = 0; u < U; ++u)        // explicit fma chain: the same rounding sequence in every slot u
                part[u] = __builtin_fmaf(x[u][3], y[3], __builtin_fmaf(x[u][2], y[2], __builtin_fmaf(x[u][1], y[1],
                                         __builtin_fmaf(x[u][0], y[0], part[u]))));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float a = part[u];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
            const long long r = row0 + w * U + u;
            if (lane == 0 && r < r1) {
                if (!(a == a)) a = -INFINITY;       // NaN scores rank last
                dst[r - r0] = make_key(a, (uint32_t)(spos[r] + pos_offset));
            }
        }
    }
}

// ---- batched scan: queries grouped by list, one fp32-MFMA GEMM tile per (list, 64 queries, 256 rows) ----
// The per-pair scan above re-reads a list once per probing query; with hundreds of queries per
// batch every list is probed by many of them, so the (query, probe) pairs are sorted by list (host-side
// integer plumbing) and each list is read ONCE per 64-query group: P = gathered queries (registers),
// Q = list rows (lanes -> a store instruction writes 2 x 32 consecutive keys of two queries' pools).
struct GatherRows {
    const float* base;
    const long long* idx;   // row r of the group -> base[idx[r]]
    long long n;
    int ld, K;
    using RowState = int;                     // the gathered row, resolved once per staged row
    __device__ __forceinline__ RowState row_state(long long r) const { return (int)idx[r < n ? r : n - 1]; }
    __device__ __forceinline__ bool k_valid(int k) const { return k < K; }
    __device__ __forceinline__ f32x4 load(RowState row, int k) const {
        return *reinterpret_cast<const f32x4*>(base + (long long)row * ld + k);
    }
};

struct EpiIvfKeys {
    static constexpr const char* name = "ivf_scan";
    static constexpr double out_bytes_per_elem = 2.0;
    static constexpr size_t lds_bytes(int) { return 0; }
    const long long* spos;      // positions of this list's rows
    long long list_rows;
    const long long* pair_q;    // this list's group: query of member j
    const long long* pair_p;    //                    probe slot of member j
    long long g;
    const long long* base;      // [nq][nprobe] pool offsets
    int nprobe;
    unsigned long long* keys;
    long long pool_ld, pos_offset;
    template <class A>
    __device__ void operator()(A& acc, float*) const {
        constexpr int TP = A::TP, TQ = A::TQ;
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int i = 0; i < TP; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long jj = acc.p(i, r, lane);          // member of the group (depends on the lane half)
                if (jj >= g) continue;
                const long long q = pair_q[jj];
                unsigned long long* dst = keys + q * pool_ld + base[q * nprobe + pair_p[jj]];
#pragma unroll
                for (int j = 0; j < TQ; ++j) {
                    const long long row = acc.q(j, lane);        // row inside the list
                    if (row >= list_rows) continue;
                    float sc = acc.v[i][j][r];
                    if (!(sc == sc)) sc = -INFINITY;
                    dst[row] = make_key(sc, (uint32_t)(spos[row] + pos_offset));
                }
            }
    }
};

using ShapeIvf = Shape<2, 2, 1, 4>;   // 64 queries x 256 list rows per workgroup

__global__ __launch_bounds__(ShapeIvf::NT, 2) void ivf_group_scan_kernel(
    const float* xs, long long ld, int d, int ksteps, const long long* spos, const long long* list_off,
    const float* Q, long long ldq, const long long* goff, const long long* qt_prefix, int nlist,
    const long long* pair_q, const long long* pair_p, const long long* base, int nprobe, unsigned long long* keys,
    long long pool_ld, long long pos_offset) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const long long y = blockIdx.y;
    if (y >= qt_prefix[nlist]) return;
    int lo = 0, hi = nlist;                  // list l with qt_prefix[l] <= y < qt_prefix[l+1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (qt_prefix[mid] <= y) lo = mid; else hi = mid;
    }
    const int l = lo;
    const long long r0 = list_off[l], len = list_off[l + 1] - r0;
    const long long row0 = (long long)blockIdx.x * ShapeIvf::BQ;
    if (row0 >= len) return;
    const long long g0 = goff[l], g = goff[l + 1] - g0;
    GatherRows lp{Q, pair_q + g0, g, (int)ldq, d};
    DenseRows lq{xs + r0 * ld, len, (int)ld, d, 30, 1ll << 30};
    EpiIvfKeys epi{spos + r0, len, pair_q + g0, pair_p + g0, g, base, nprobe, keys, pool_ld, pos_offset};
    gemm_block<ShapeIvf>(lp, lq, epi, ksteps, (y - qt_prefix[l]) * ShapeIvf::BP, row0, smem);
}

// k largest of keys[q][0..n_q) -> sorted (score desc, position asc); fewer than k -> padded (-inf, -1)
__global__ __launch_bounds__(512) void ivf_select_kernel(const unsigned long long* keys, long long pool_ld,
                                                         const long long* n_pool, int k, float* outD,
                                                         long long* outI) {
    __shared__ int hist[2048];
    __shared__ int scratch[514];
    __shared__ __attribute__((aligned(16))) unsigned long long buf[2048];
    __shared__ int count;
    const long long q = blockIdx.x;
    const int tid = threadIdx.x;
    const long long n = n_pool[q];
    const unsigned long long* row = keys + q * pool_ld;
    unsigned long long kstar = 0ull;          // select everything by default (n <= k)
    if (n > k) {
        unsigned long long prefix = 0ull, pmask = 0ull;
        int rr = k;
        const int shifts[6] = {53, 42, 31, 20, 9, 0};
        const int bits[6] = {11, 11, 11, 11, 11, 9};
        for (int pass = 0; pass < 6; ++pass) {
            for (int i = tid; i < 2048; i += 512) hist[i] = 0;
            __syncthreads();
            const unsigned long long bm = (1ull << bits[pass]) - 1;
            for (long long i = tid; i < n; i += 512) {
                const unsigned long long key = row[i];
                if ((key & pmask) == prefix) atomicAdd(&hist[(int)((key >> shifts[pass]) & bm)], 1);
            }
            __syncthreads();
            const int bin = find_bin_desc<2048, 512>(hist, rr, scratch);   // always found: n > k >= rr
            prefix |= (unsigned long long)bin << shifts[pass];
            pmask |= bm << shifts[pass];
        }
        kstar = prefix;                        // the k-th largest key itself
    }
    if (tid == 0) count = 0;
    int P = 2;
    while (P < k) P <<= 1;
    for (int i = tid; i < P; i += 512) buf[i] = 0ull;
    __syncthreads();
    for (long long i = tid; i < n; i += 512) {
        const unsigned long long key = row[i];
        if (key >= kstar) {
            const int pos = atomicAdd(&count, 1);
            if (pos < 2048) buf[pos] = key;    // exactly min(n, k) keys pass (keys are unique)
        }
    }
    __syncthreads();
    bitonic_desc(buf, P);
    const int have = count < k ? count : k;
    write_result(buf, have, k, q, outD, outI, 0);
}

}  // namespace amdrec

using namespace amdrec;

extern "C" int amdrec_ivf_scan(const float* lists, int64_t ld, int dim, const int64_t* row_pos,
                               const int64_t* list_off, const float* queries, int64_t nq, int64_t ld_queries,
                               const int64_t* probes, const int64_t* pool_base, int nprobe, uint64_t* pool_keys,
                               int64_t pool_ld, int64_t pos_offset, void* stream) {
    REQUIRE(dim >= 4 && dim % 4 == 0 && dim <= 2048, "dim=%d must be a multiple of 4 in [4,2048]", dim);
    REQUIRE(nprobe >= 1 && nprobe <= 65535, "nprobe out of range");
    if (nq <= 0) return AMDREC_OK;
    REQUIRE(nq <= 65535, "at most 65535 queries per call");
    REQUIRE(lists && row_pos && list_off && queries && probes && pool_base && pool_keys, "null pointer");
    REQUIRE(ld % 4 == 0 && ld >= dim && ld_queries >= dim, "bad leading dimension");
    hipLaunchKernelGGL(ivf_scan_kernel, dim3(nprobe, (unsigned)nq), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       lists, (long long)ld, dim, (const long long*)row_pos, (const long long*)list_off, queries,
                       (long long)ld_queries, (const long long*)probes, (const long long*)pool_base, nprobe,
                       (unsigned long long*)pool_keys, (long long)pool_ld, (long long)pos_offset);
    HIP_TRY(hipGetLastError());
    return AMDREC_OK;
}

extern "C" int amdrec_ivf_scan_grouped(const float* lists, int64_t ld, int dim, const int64_t* row_pos,
                                       const int64_t* list_off, int nlist, int64_t max_list_rows,
                                       const float* queries, int64_t ld_queries, const int64_t* group_off,
                                       const int64_t* qtile_prefix, int64_t qtile_bound, const int64_t* pair_query,
                                       const int64_t* pair_probe, const int64_t* pool_base, int nprobe,
                                       uint64_t* pool_keys, int64_t pool_ld, int64_t pos_offset, void* stream) {
    REQUIRE(dim >= 4 && dim % 4 == 0 && dim <= 2048, "dim=%d must be a multiple of 4 in [4,2048]", dim);
    REQUIRE(nlist >= 1 && nprobe >= 1, "bad nlist/nprobe");
    if (qtile_bound <= 0 || max_list_rows <= 0) return AMDREC_OK;
    REQUIRE(qtile_bound <= 65535, "too many (list, query-tile) groups for one launch: chunk the queries");
    REQUIRE(lists && row_pos && list_off && queries && group_off && qtile_prefix && pair_query && pair_probe &&
                pool_base && pool_keys, "null pointer");
    REQUIRE(ld % 4 == 0 && ld >= dim && ld_queries >= dim && ld_queries % 4 == 0, "bad leading dimension");
    static PerDeviceOnce attr_done;
    if (attr_done.pending()) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(ivf_group_scan_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)ShapeIvf::LDS_BYTES));
        attr_done.mark();
    }
    const unsigned gx = (unsigned)((max_list_rows + ShapeIvf::BQ - 1) / ShapeIvf::BQ);
    hipLaunchKernelGGL(ivf_group_scan_kernel, dim3(gx, (unsigned)qtile_bound), dim3(ShapeIvf::NT), ShapeIvf::LDS_BYTES,
                       reinterpret_cast<hipStream_t>(stream), lists, (long long)ld, dim, (dim + BK - 1) / BK,
                       (const long long*)row_pos, (const long long*)list_off, queries, (long long)ld_queries,
                       (const long long*)group_off, (const long long*)qtile_prefix, nlist,
                       (const long long*)pair_query, (const long long*)pair_probe, (const long long*)pool_base, nprobe,
                       (unsigned long long*)pool_keys, (long long)pool_ld, (long long)pos_offset);
    HIP_TRY(hipGetLastError());
    return AMDREC_OK;
}

extern "C" int amdrec_ivf_select(const uint64_t* pool_keys, int64_t pool_ld, const int64_t* pool_count, int64_t nq,
                                 int k, float* out_scores, int64_t* out_pos, void* stream) {
    REQUIRE(k >= 1 && k <= AMDREC_MAX_K, "k=%d outside [1,%d]", k, AMDREC_MAX_K);
    if (nq <= 0) return AMDREC_OK;
    REQUIRE(pool_keys && pool_count && out_scores && out_pos, "null pointer");
    hipLaunchKernelGGL(ivf_select_kernel, dim3((unsigned)nq), dim3(512), 0, reinterpret_cast<hipStream_t>(stream),
                       (const unsigned long long*)pool_keys, (long long)pool_ld, (const long long*)pool_count, k,
                       out_scores, (long long*)out_pos);
    HIP_TRY(hipGetLastError());
    return AMDREC_OK;
}
