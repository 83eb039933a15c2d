// Exact inner-product top-k over a flat corpus (replaces faiss IndexFlatIP.search as called
// from FAISSIndex.search, faiss_retrieval.py:155).
//
// Algorithm ("threshold-prefiltered exact top-k"), all on device, no host sync:
//   1. sample pass : scores of every query against an evenly spaced subset of corpus tiles
//                    (fp32 MFMA GEMM, dense store)                     ~3 % of the corpus
//   2. threshold   : per query tau_q ~ the r-th largest sample score (r-th largest of 256 per-group
//                    maxima); the expected number of corpus rows with score >= tau_q is ~max(2k, k+900)
//   3. filter pass : fp32 MFMA GEMM over the whole corpus; the epilogue appends
//                    (score, row) keys with score >= tau_q to a per-query candidate list
//   4. finalize    : per query, if k <= count <= capacity the exact top-k is inside the
//                    list (every row >= tau is there) -> bitonic sort of 64-bit keys in LDS
//   5. fix-up      : queries whose count fell outside [k, capacity] (adversarial order,
//                    massive ties) are re-done by an exact streaming scan (16 slices per
//                    query, running threshold) + merge.  Blocks of healthy queries exit at once.
// The result is exact for every input; only the speed depends on the data.
// Order: score descending, equal scores -> lower row first (deterministic).
//
// Mixed-precision variant (amdrec_flat_search_mixed): passes 1-3 run on a bf16 copy of the corpus and of the
// queries with v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate, half the HBM bytes), which turns the filter from
// MFMA-bound into memory-bound; the result stays the exact fp32 one because
//   * eps_q = ||dq|| (M + D) + ||q|| D bounds |approx - exact| for every row, from the MEASURED rounding errors of the
//     query (dq) and of the corpus (D = max row error norm, M = max row norm; see eps_bound below): at worst the classical
//     (2u + u^2) |q| |x| with u = 2^-8, ~4x smaller on real data; plus the fp32 accumulation slack,
//   * finalize radix-selects a_k = the k-th largest approximate score, keeps the candidates with approx >= a_k - 2 eps
//     (anything below is beaten by k rows), RE-SCORES them in fp32 from the fp32 corpus and sorts them,
//   * and certifies: every row outside the list has approx < tau, i.e. exact < tau + eps; if the k-th re-scored
//     value is >= tau + eps nothing outside can enter the top-k.  A query that fails the certificate (or the
//     count test) goes to the same exact fp32 fix-up scan as before.
// For dims 32 / 64 / 128 / 256 the corpus pass of this variant is scan_filter_kernel (below), not a generic GEMM tile.
#include <type_traits>
#include "gemm_core.hpp"
#include "topk_utils.hpp"
#include "../../include/amdrec.h"

namespace amdrec {

constexpr int CAND_CAP = 8192;     // candidate keys per query (64 KB LDS sort)
constexpr long long CAND_STRIDE = 2 * CAND_CAP;   // mixed search: keys between the queries' blocks ([segments | overflow])
constexpr int SAMPLE_RANK = 64;    // r
constexpr int KMAX = 2048;
constexpr int FIX_BUF = 4096;      // fix-up scan buffer (keys)
constexpr int FIX_GRID_Q = 256;    // queries worked on at a time by the fix-up kernels
constexpr int SAMPLE_G = 256;      // rows per sample block (== BP of every search shape / divides it)

// ---- epilogues (lane <-> query, registers <-> corpus rows) ---------------------------
struct EpiStoreScores {
    static constexpr const char* name = "search_sample";
    static constexpr double out_bytes_per_elem = 1.0;
    static constexpr size_t lds_bytes(int) { return 0; }
    float* out;          // [nq][ld]
    long long ld;
    int nq;
    long long n_sample;  // sample rows
    DenseRows map;       // to test validity of the mapped row
    template <class A>
    __device__ void operator()(A& acc, float*) const {
        constexpr int TP = A::TP, TQ = A::TQ;
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int j = 0; j < TQ; ++j) {
            int q = acc.q(j, lane);
            if (q >= nq) continue;
#pragma unroll
            for (int i = 0; i < TP; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    int p = acc.p(i, 4 * g, lane);   // 4 consecutive rows p..p+3
                    if (p >= n_sample) continue;
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float s = acc.v[i][j][4 * g + e];
                        v[e] = (map.map(p + e) < map.rows) ? s : -INFINITY;
                    }
                    *reinterpret_cast<f32x4*>(out + (long long)q * ld + p) = v;
                }
        }
    }
};

// Filter epilogue: keys of (score >= tau_q) elements are appended to the per-query candidate lists.  A returning
// global atomic per hit made the epilogue a chain of ~10 dependent 1-2 us round trips per wave (measured: the bf16
// filter spent more time here than in its K loop), so hits are first collected in an LDS list (LDS atomics), and after
// one barrier each hit is appended by its own lane: the global atomics of a block go out together.  Hits beyond the
// LDS list (threshold-less small corpora, adversarial data) take the direct path.
constexpr int HIT_CAP = 2048;
struct EpiFilter {
    static constexpr const char* name = "search_filter";
    static constexpr double out_bytes_per_elem = 0.0;
    static constexpr size_t lds_bytes(int) { return 16 + (size_t)HIT_CAP * 12; }
    const float* tau;            // [nq]
    unsigned long long* cand;    // [nq][stride], the first cap of each block are used
    int* cnt;                    // [nq]
    int cap, nq;
    long long nrows;
    long long stride;
    __device__ __forceinline__ void append(int q, unsigned long long key) const {
        const int pos = atomicAdd(&cnt[q], 1);
        if (pos < cap) cand[(long long)q * stride + pos] = key;
    }
    template <class A>
    __device__ void operator()(A& acc, float* smem) const {
        constexpr int TP = A::TP, TQ = A::TQ;
        const int lane = threadIdx.x & 63;
        int* lcount = reinterpret_cast<int*>(smem);
        unsigned long long* hkey = reinterpret_cast<unsigned long long*>(smem + 4);      // 16-byte offset
        int* hq = reinterpret_cast<int*>(hkey + HIT_CAP);
        if (threadIdx.x == 0) *lcount = 0;          // the K loop's last barrier released the staging tiles
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TQ; ++j) {
            int q = acc.q(j, lane);
            float t = (q < nq) ? tau[q] : __builtin_nanf("");      // NaN: no score compares >= it
#pragma unroll
            for (int i = 0; i < TP; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float s = acc.v[i][j][r];
                    int p = acc.p(i, r, lane);
                    if (s >= t && p < nrows) {
                        const unsigned long long key = make_key(s, (uint32_t)p);
                        const int slot = atomicAdd(lcount, 1);
                        if (slot < HIT_CAP) { hkey[slot] = key; hq[slot] = q; }
                        else append(q, key);
                    }
                }
        }
        __syncthreads();
        const int n = *lcount < HIT_CAP ? *lcount : HIT_CAP;
        for (int h = threadIdx.x; h < n; h += blockDim.x) append(hq[h], hkey[h]);
    }
};

// Threshold from the sample: tau[q] only has to BOUND the candidate count (any value with
// k <= #{score >= tau} <= capacity gives the exact result), so instead of an exact radix select of the
// r-th largest sample score (3 histogram passes, measured 157 us per launch) each of 256 thread groups
// takes the maximum of its strided share of the sample and tau is the r-th largest of those 256 maxima:
// one coalesced pass.  The true sample rank of that value is >= r (two of the top values may share a
// group; expected r + r^2/512), i.e. the threshold errs on the side of MORE candidates, far inside the capacity.
// NaN counts as lowest; fewer than r finite maxima -> -inf.
constexpr int THR_NT = 1024;
__global__ __launch_bounds__(THR_NT) void sample_threshold_kernel(const float* S, long long ld, long long n, int r,
                                                                  float* tau) {
    __shared__ float mx[THR_NT];
    const int q = blockIdx.x, tid = threadIdx.x;
    const float* row = S + (long long)q * ld;
    float m = -INFINITY;
    // four independent 16-byte loads in flight per thread (one block per query: the pass is latency-bound)
    for (long long i0 = 4ll * tid; i0 < n; i0 += 16 * THR_NT) {  // n % 256 == 0, rows 16-byte aligned
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long i = i0 + 4ll * u * THR_NT;
            v[u] = i < n ? *reinterpret_cast<const f32x4*>(row + i) : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) m = (v[u][e] > m) ? v[u][e] : m;   // NaN compares false: ignored
    }
    mx[tid] = m;
    __syncthreads();
    // 256 group maxima (each over 4 threads' shares), ranked by the first 4 waves
    __shared__ float gm[256];
    if (tid < 256) {
        m = fmaxf(fmaxf(mx[tid], mx[tid + 256]), fmaxf(mx[tid + 512], mx[tid + 768]));
        gm[tid] = m;
    }
    __syncthreads();
    if (tid >= 256) return;
    int rank = 0;                     // number of maxima ahead of mine (ties: lower thread first)
    for (int j = 0; j < 256; ++j) {
        const float o = gm[j];
        rank += (o > m) || (o == m && j < tid);
    }
    if (rank == r - 1) tau[q] = m;    // exactly one thread has this rank (r <= 256)
}

// step 4: sort each query's candidate list, or flag it for the fix-up
__global__ __launch_bounds__(512) void finalize_kernel(const unsigned long long* cand, const int* cnt, int cap,
                                                       int k, long long nrows, int* fail, float* outD,
                                                       long long* outI, long long pos_offset) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    const int q = blockIdx.x;
    const int c = cnt[q];
    const int need = (int)(nrows < k ? nrows : k);
    if (c < need || c > cap) {
        if (threadIdx.x == 0) { fail[q] = 1; atomicAdd(&fail[gridDim.x], 1); }
        return;
    }
    int P = 2;
    while (P < c) P <<= 1;
    for (int i = threadIdx.x; i < P; i += blockDim.x) keys[i] = (i < c) ? cand[(long long)q * cap + i] : 0ull;
    __syncthreads();
    bitonic_desc(keys, P);
    write_result(keys, c, k, q, outD, outI, pos_offset);
}

// step 5: exact streaming scan of one corpus slice for a failed query; the block that finishes a query's LAST slice (a
// ticket per query, agent-scope release / acquire around it as in cdna_hip_programming.md "in-launch split-K reduction")
// merges the slices and writes the result.  One launch: in the common case - no failed query - it is the only cost of the
// fix-up, and two empty launches cost 8 us of a 140 us single-query search.
__global__ __launch_bounds__(512) void fixup_kernel(const float* X, long long ldx, long long nrows, int d,
                                                    const float* Q, long long ldq, const int* fail, int nq,
                                                    int k, int nslices, unsigned long long* scratch, int* ticket,
                                                    float* outD, long long* outI, long long pos_offset) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];      // merge: k * nslices keys (<= CAND_CAP)
    __shared__ int last_sh;
    __shared__ __attribute__((aligned(16))) unsigned long long buf[FIX_BUF];
    __shared__ __attribute__((aligned(16))) float qv[2048];
    __shared__ unsigned long long thr;
    __shared__ int count;
    // 1-D grid (query-major) of nslices x min(nq, FIX_GRID_Q) blocks, each looping over its queries: failures are
    // rare, and one block per (query, slice) cost 0.07 ms of empty blocks at 4096 queries
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr int NW = 8, U = 8;
    const int s = blockIdx.x % nslices;
    if (fail[nq] == 0) return;                                 // fail[nq] = number of failed queries
    for (int q = blockIdx.x / nslices; q < nq; q += gridDim.x / nslices) {
    if (!fail[q]) continue;                                    // block-uniform
    for (int i = tid; i < d; i += 512) qv[i] = Q[(long long)q * ldq + i];
    if (tid == 0) { thr = 0ull; count = 0; }
    __syncthreads();
    const long long per = (nrows + nslices - 1) / nslices;
    const long long begin = (long long)s * per, end = (begin + per < nrows) ? begin + per : nrows;
    const int d4 = d >> 2;
    auto compact = [&]() {
        int c = count;
        for (int i = tid; i < FIX_BUF; i += 512) if (i >= c) buf[i] = 0ull;
        __syncthreads();
        bitonic_desc(buf, FIX_BUF);
        if (tid == 0) {
            int nc = c < k ? c : k;
            count = nc;
            thr = (nc == k) ? buf[k - 1] : 0ull;
        }
        __syncthreads();
    };
    for (long long row0 = begin; row0 < end; row0 += NW * U) {
        float part[U];
        const f32x4* xr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            long long r = row0 + w * U + u;
            r = r < end ? r : end - 1;                         // clamped: branch-free loads, result dropped below
            xr[u] = reinterpret_cast<const f32x4*>(X + r * ldx);
            part[u] = 0.f;
        }
        // column chunk outermost so that the U row loads of a chunk are in flight together
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
            long long r = row0 + w * U + u;
            if (lane == 0 && r < end && a == a) {
                unsigned long long key = make_key(a, (uint32_t)r);
                if (key > thr) {
                    int pos = atomicAdd(&count, 1);
                    buf[pos] = key;   // pos < FIX_BUF guaranteed by the compaction rule below
                }
            }
        }
        __syncthreads();
        // block-uniform decision: every thread reads `count` before anyone may change it again
        if (__syncthreads_or(count > FIX_BUF - NW * U)) compact();
    }
    compact();
    const int c = count;
    unsigned long long* dst = scratch + ((long long)q * nslices + s) * k;
    for (int i = tid; i < k; i += 512) dst[i] = (i < c) ? buf[i] : 0ull;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's slice stores are out
    __syncthreads();                                           // ... every wave's; also: before count / thr / buf are reset
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the fence's own wait may be dropped: keep this one)
        const int t = __hip_atomic_fetch_add(&ticket[q], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_sh = t == nslices - 1;
        if (t == nslices - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (last_sh) {                                             // block-uniform: merge the query's slices
        const int total = k * nslices;
        int P = 2;
        while (P < total) P <<= 1;
        for (int i = tid; i < P; i += 512) keys[i] = (i < total) ? scratch[(long long)q * total + i] : 0ull;
        __syncthreads();
        bitonic_desc(keys, P);
        write_result(keys, total < k ? total : k, k, q, outD, outI, pos_offset);
    }
    __syncthreads();
    }
}

// Cross-shard merge: n_lists sorted lists of list_k entries per query (one per corpus shard, positions
// already global) -> the global top-k.  One block per query, bitonic sort in LDS.
// Short lists (list_k < k, amdrec_topk_merge_partial): a shard sends only its best list_k rows in the search's own total
// order (score descending, ties -> lower position), so every row it did NOT send is strictly behind its last entry in
// that order.  The merged top-k is therefore the exact global top-k iff no FULL list's last entry lies strictly ahead of
// the merged k-th entry - compared as (score, position) keys, the order the merge itself sorts by.  A tie in SCORE alone
// proves nothing either way and is not counted (duplicate ads with bit-identical embeddings are routine,
// training_pipeline.py:523): the last entry of a list may BE the merged k-th, or tie with it at a higher position, and
// the rows behind it still cannot enter.  A query for which the rule fails - a full list that ends ahead of the merged
// k-th key, or fewer than k merged entries - is counted in *inexact.
__global__ __launch_bounds__(512) void topk_merge_kernel(const char* scores, const char* pos, int n_lists, int list_k,
                                                         long long list_stride_bytes, long long q0, int k,
                                                         float* outD, long long* outI, int* inexact) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    const long long q = blockIdx.x;
    const int total = n_lists * list_k;
    int P = 2;
    while (P < total) P <<= 1;
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        unsigned long long key = 0ull;
        if (i < total) {
            const int g = i / list_k, j = i - g * list_k;
            const float* sg = reinterpret_cast<const float*>(scores + (long long)g * list_stride_bytes);
            const int* pg = reinterpret_cast<const int*>(pos + (long long)g * list_stride_bytes);
            const long long p = pg[(q0 + q) * list_k + j];
            const float sc = sg[(q0 + q) * list_k + j];
            if (p >= 0 && sc == sc) key = make_key(sc, (uint32_t)p);
        }
        keys[i] = key;
    }
    __syncthreads();
    bitonic_desc(keys, P);
    if (inexact != nullptr) {
        const unsigned long long kth = k <= P ? keys[k - 1] : 0ull;
        bool cut = false;
        for (int g = threadIdx.x; g < n_lists; g += blockDim.x) {
            const float* sg = reinterpret_cast<const float*>(scores + (long long)g * list_stride_bytes);
            const int* pg = reinterpret_cast<const int*>(pos + (long long)g * list_stride_bytes);
            const long long p = pg[(q0 + q) * list_k + list_k - 1];
            const float sc = sg[(q0 + q) * list_k + list_k - 1];
            const bool full = p >= 0 && sc == sc;
            cut |= full && (kth == 0ull || make_key(sc, (uint32_t)p) > kth);
        }
        if (__syncthreads_or(cut) && threadIdx.x == 0) atomicAdd(inexact, 1);
    }
    write_result(keys, total < k ? total : k, k, q, outD, outI, 0);
}

// ---- streaming bf16 filter (the mixed search's corpus pass for dim in {32, 64, 128, 256}) ---------------------
// The generic GEMM tiles re-stage the query tile for every corpus tile and prefetch one 128-byte K-step ahead: with
// K = 256 bf16 a tile is only four K-steps of 0.4 us of MFMA each, far less than the HBM latency, and the pass ran
// latency-bound at ~14 % of HBM.  Here the roles are fixed for the whole launch instead:
//  * one 8-wave workgroup per CU; wave w owns queries [64 w, 64 w + 64) of the block's 512-query group and keeps
//    their bf16 fragments for the WHOLE K extent in registers (2 x KS x 4 VGPRs = 128 at dim 256): queries are read
//    once per launch, never staged through LDS (groups of <= 256 queries: the spare waves share the tile's rows);
//  * the corpus streams through a two-deep LDS ring of 128-row full-K tiles (64 KB at dim 256) filled by LDS-DMA
//    (global_load_lds, 16 B per lane, source-side XOR swizzle so the fragment reads are conflict-free); the next tile
//    is in flight during the MFMAs on the current one - one barrier per tile;
//  * every wave multiplies the same 32 corpus rows at a time (one A fragment per k16 step, read from LDS four steps
//    ahead) with its own 64 queries: 2 accumulator tiles, v_mfma_f32_32x32x16_bf16, then compares against tau_q;
//  * hits go to a wave-private LDS list without atomics (ballot + lane prefix; the wave's count lives in an SGPR);
//    one tile later the wave files them in the workgroup's OWN segment of each query's candidate block (slot-major,
//    CAND_CAP / workgroups slots per query; the slot comes from an LDS counter) - no global atomic, and the key stores
//    go out before the next tile's DMA so the in-order vmcnt wait for that DMA never waits for them.  A full segment
//    spills to the query's overflow block through a global counter; finalize_mixed_kernel reads the segments in place.
// Each corpus byte is read from HBM once per 512 queries; waves whose queries lie beyond nq skip their MFMAs, so a
// small batch runs at the HBM rate and a full one at the bf16 MFMA rate.
// tau of ONE query by ONE wave from its row of group maxima gm[0 .. n) (n % 4 == 0, 16-byte aligned): float4 chunk i
// belongs to group i % 256; tau = the r-th largest of the 256 group maxima (r <= 256; fewer than r finite maxima -> -inf;
// NaN never wins a maximum).  The same estimator as sample_threshold_kernel: expected sample rank of the result r + r^2/512,
// i.e. the candidate count errs high, far inside the capacity.  Selection without LDS: a lane holds 4 of the 256 maxima
// as order-preserving 32-bit keys; the r-th largest is built bit by bit (largest T with #{key >= T} >= r), each count a
// ballot + scalar popcount per register - ~500 instructions, where ranking every value against all 256 through LDS cost
// 25 us per query.
__device__ __forceinline__ float wave_tau(const float* row, long long n, int r, int lane) {
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    const long long nch = n >> 2;
    for (long long c0 = 0; c0 < nch; c0 += 512) {           // eight independent 16-byte loads in flight per lane
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long i = c0 + 64 * u + lane;
            v[u] = i < nch ? *reinterpret_cast<const f32x4*>(row + 4 * i) : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) m[u & 3] = v[u][e] > m[u & 3] ? v[u][e] : m[u & 3];     // chunk i -> group i % 256
    }
    uint32_t key[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t b = __float_as_uint(m[u]);
        key[u] = (b & 0x80000000u) ? ~b : (b | 0x80000000u);     // unsigned order == float order (no NaN among the maxima)
    }
    uint32_t T = 0u;
#pragma unroll 1
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t c = T | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) cnt += __builtin_popcountll(__ballot(key[u] >= c));
        if (cnt >= r) T = c;                                  // wave-uniform
    }
    const uint32_t b = (T & 0x80000000u) ? (T & 0x7fffffffu) : ~T;
    return __uint_as_float(b);
}

// Round-4 switches of the corpus pass (bits of AMDREC_SCAN_OPT; same-box A/Bs: tools/scan_ab.sh, profiles/r04_scan_*):
// DEFAULT 32.  Measured and NOT adopted (profiles/r04_scan_ab.log, same box, search-only loop at 512 / 128 queries): 1 ->
// 0.287-0.297 against 0.286-0.287 ms per pass, 17 -> 0.285-0.287 against 0.283-0.287, 2 -> 0.288-0.289 against 0.280-0.282
// (0.151 against 0.118 at 128 queries: a 64-row slot leaves the spare waves of a small group half the row parts), 3 ->
// 0.298-0.301, 96 -> 0.311-0.312 against 0.289-0.290 (13 spilled registers).  Adopted: 32 -> 0.289-0.292 against 0.296-0.297
// at 512 queries, neutral below.  Diagnostic bits: 4 = no
// corpus DMA (the waves compute on whatever the ring holds), 8 = no hit handling (old quarter code only).
//  32  the quarter's LDS-read / MFMA interleave spelled out with sched_group_barrier (the scheduler otherwise hoists all 16
//      fragment reads of a quarter in front of its first MFMA)
//  64  (with 32) two accumulator sets ping-pong: the previous quarter's threshold scan rides between the current quarter's
//      MFMAs and the next quarter's first fragments are requested behind its last ones
//  16  (with 1) the A fragments are carried across the units: a unit's last MFMAs prefetch the next unit's first fragments
//   2  ring of four 64-row slots instead of two 128-row tiles (DESIGN section 7.3's experiment): DMA three slots ahead
//   1  j-major MFMA order with the threshold scan of the PREVIOUS (32 rows x 32 queries) accumulator interleaved between the
//      MFMAs of the current one: the scan (and its hit path) used to run after a quarter's last MFMA with the matrix pipe
//      idle - and both waves of a SIMD reach that point together, the tile barrier keeps them in step.  Costs a second
//      read of every A fragment (once per query tile); no extra registers (the two accumulators ping-pong).
#ifndef AMDREC_SCAN_OPT
#define AMDREC_SCAN_OPT 32
#endif
constexpr int SCAN_OPT = AMDREC_SCAN_OPT;
constexpr int SCAN_ROWS = 128;          // corpus rows per LDS tile
constexpr int SCAN_WHITS = 128;         // hit-list entries per wave and tile (expected ~11; overflow -> direct append)
constexpr int SCAN_QGROUP = 512;        // queries per workgroup (8 waves x 64)
constexpr int SCAN_HIT_BYTES = 2 * 8 * SCAN_WHITS * 12;   // [2 lists][8 waves] x {keys u64[SCAN_WHITS], q int[SCAN_WHITS]}

// LDS stores of the hit list, hidden from the compiler: it orders every LDS write it can see after the LDS-DMA in
// flight (s_waitcnt vmcnt(0)), which would stall each wave on the next tile's loads at its first hit.  The hit lists
// and the tile ring are disjoint and a wave's LDS operations execute in order, so no wait is needed.
__device__ __forceinline__ void lds_store_hit_opaque(uint32_t key_addr, unsigned long long key, uint32_t q_addr, int q) {
    asm volatile("ds_write_b64 %0, %1\n\tds_write_b32 %2, %3" ::"v"(key_addr), "v"(key), "v"(q_addr), "v"(q) : "memory");
}

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): an index walk whose indices are constants in the body
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

template <int KS>                       // KS = dim / 16 in {2, 4, 8, 16}
__global__ __launch_bounds__(512, 1) void scan_filter_kernel(const uint16_t* __restrict__ X16, long long ld16,
                                                             long long nrows, const uint16_t* __restrict__ Q16, int nq,
                                                             float* __restrict__ tau, unsigned long long* cand,
                                                             int* segcnt, int* ocnt, int seg_cap, int nx,
                                                             const float* __restrict__ gm, long long ldm, long long gm_n,
                                                             int rank) {
    constexpr int CPR = 2 * KS;                                  // 16-byte chunks per row
    constexpr int FM = CPR < 16 ? CPR - 1 : 15;                  // swizzle mask
    constexpr int FS = CPR >= 16 ? 0 : (CPR == 8 ? 1 : 2);       // swizzle row shift
    // SCAN_OPT & 2: the ring is FOUR slots of 64 rows instead of two of 128 (same 128 KB): a slot's DMA is issued three slots
    // ahead of its first read instead of one tile ahead, and the workgroup meets at a barrier every 64 rows
    constexpr bool RING4 = (SCAN_OPT & 2) != 0 && KS >= 4;
    constexpr int TR = RING4 ? 64 : SCAN_ROWS;                   // corpus rows per ring slot
    constexpr int NSLOT = RING4 ? 4 : 2;
    constexpr int QPT = TR / 32;                                 // 32-row quarters per slot
    constexpr int TILE_CHUNKS = TR * CPR;
    constexpr int PASSES = TILE_CHUNKS / 512;
    constexpr int AHEAD = KS < 4 ? KS : 4;                       // A fragments read ahead of their MFMA
    static_assert(TILE_CHUNKS % 512 == 0, "tile must fill whole DMA passes");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* tilebuf = lds;                                                  // [NSLOT][TILE_CHUNKS * 16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);                        // scalar: wave-uniform branches below
    unsigned char* hitbase = lds + NSLOT * TILE_CHUNKS * 16 + w * (SCAN_WHITS * 12);   // this wave's list 0; list 1 is
    constexpr int LIST_STRIDE = 8 * SCAN_WHITS * 12;                               // LIST_STRIDE bytes further
    int* lcnt = reinterpret_cast<int*>(lds + NSLOT * TILE_CHUNKS * 16 + SCAN_HIT_BYTES);   // [SCAN_QGROUP] hits per query, this WG
    lcnt[tid] = 0;                                                                 // (before the first DMA is in flight)
    const int frow = lane & 31, fh = lane >> 5;
    const int bx = blockIdx.x % nx, by = blockIdx.x / nx;
    // Work split inside the workgroup.  A full group gives every wave its own 64 queries and all 128 rows of a tile; a
    // group with fewer queries would leave most waves idle and one or two waves with the whole tile's MFMAs, threshold
    // scan and hit handling (measured: 0.205 ms at 128 queries against 0.117 with the split), so the waves without
    // queries of their own take a share of the tile's ROWS instead: 1 / 2 / 4 / 8 query waves x up to four 32-row parts.
    const int nqg = nq - by * SCAN_QGROUP < SCAN_QGROUP ? nq - by * SCAN_QGROUP : SCAN_QGROUP;
    const int qsh = nqg <= 64 ? 0 : (nqg <= 128 ? 1 : (nqg <= 256 ? 2 : 3));       // log2 of the query waves
    const int parts = (8 >> qsh) < QPT ? (8 >> qsh) : QPT;                          // 128-row slots: 4 / 4 / 2 / 1
    const int part = w >> qsh;
    const int q0 = by * SCAN_QGROUP + (w & ((1 << qsh) - 1)) * 64;
    const bool active = q0 < nq && part < parts;
    const int rq_begin = part * (QPT / parts), rq_end = rq_begin + QPT / parts;
    const bool second = q0 + 32 < nq;                                              // second query tile has real queries
    const int ntiles = (int)((nrows + SCAN_ROWS - 1) / SCAN_ROWS);
    const int nrows_i = (int)nrows;                                                // < 2^31 (checked by the entry point)

    // query fragments (clamped rows); the thresholds tq are filled in below, once they are known
    bf16x8 qf[2][KS];
    float tq[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = q0 + j * 32 + frow;
        const int qc = q < nq ? q : nq - 1;
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_)
            qf[j][s_] = *reinterpret_cast<const bf16x8*>(Q16 + (long long)qc * (16 * KS) + (2 * s_ + fh) * 8);
    }

    // DMA lane map: LDS chunk slot L = u * 512 + tid of pass u -> row u * RPP + tid / CPR, slot tid % CPR; the source
    // chunk (slot ^ swizzle(row)) does not depend on u (RPP is a multiple of the swizzle period), so a lane's source
    // address is a per-lane byte offset plus a scalar per (tile, pass)
    constexpr int RPP = 512 / CPR;                                 // rows per DMA pass
    const int drow = tid / CPR;
    const uint32_t lane_off = (uint32_t)(drow * (int)ld16 * 2 + (((tid % CPR) ^ ((drow >> FS) & FM)) * 16));
    auto dma = [&](int row0i, int buf) {                           // TR rows from corpus row `row0i` into ring slot `buf`
        if (SCAN_OPT & 4) return;                                  // (diagnostic build)
        const long long row0 = row0i;
        unsigned char* lbase = tilebuf + (size_t)buf * TILE_CHUNKS * 16 + (size_t)(w * 64) * 16;           // wave-uniform
        if (row0 + TR <= nrows) {
            const unsigned char* gb = reinterpret_cast<const unsigned char*>(X16) + row0 * ld16 * 2;
#pragma unroll
            for (int u = 0; u < PASSES; ++u) {
                const unsigned char* g = gb + (long long)u * RPP * ld16 * 2 + lane_off;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(lbase + (size_t)u * 512 * 16),
                                                 16, 0, 0);
            }
        } else {                                                   // last tile: rows past the end are clamped
#pragma unroll
            for (int u = 0; u < PASSES; ++u) {
                long long r = row0 + u * RPP + drow;
                r = r < nrows ? r : nrows - 1;                     // (a slot wholly past the end reads the last row: never a hit)
                const unsigned char* g = reinterpret_cast<const unsigned char*>(X16) + r * ld16 * 2 +
                                         (((tid % CPR) ^ ((drow >> FS) & FM)) * 16);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(lbase + (size_t)u * 512 * 16),
                                                 16, 0, 0);
            }
        }
    };
    // A hit goes to THIS workgroup's segment of the query's candidate block: the slot comes from an LDS counter, so the
    // corpus pass issues no global atomic (a returning device-scope atomic per hit, and its store waited for at the end
    // of the tile, cost 0.07-0.25 ms per pass: profiles/r02_scan_hit_cost.log).  A full segment spills to the query's
    // overflow block through a global counter (large k, skewed data: rare).
    const int qbase = by * SCAN_QGROUP;
    auto append = [&](int q, unsigned long long key) {
        const int slot = atomicAdd(&lcnt[q - qbase], 1);
        unsigned long long* blk = cand + (long long)q * CAND_STRIDE;
        if (slot < seg_cap) {
            blk[slot * nx + bx] = key;                     // slot-major: the used slots of all segments are the block's head
        } else {
            const int o = atomicAdd(&ocnt[q], 1);
            if (o < CAND_CAP) blk[CAND_CAP + o] = key;
        }
    };
    // 32 corpus rows (LDS image at `lb`, fragment k-offset swizzle G) x NJ query tiles: K loop with the A fragments
    // read AHEAD steps before their MFMA, then the threshold scan.  wcount = this wave's hits so far in this tile.
    auto quarter = [&](auto nj_tag, auto full_tag, const unsigned char* lb, int G, int prow, uint32_t list_addr,
                       int& wcount) {
        constexpr int NJ = decltype(nj_tag)::value;
        constexpr bool FULL = decltype(full_tag)::value;           // every row of the tile is a corpus row
        f32x16 acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        bf16x8 a[AHEAD];
#pragma unroll
        for (int s_ = 0; s_ < AHEAD; ++s_) a[s_] = *reinterpret_cast<const bf16x8*>(lb + ((32 * s_) ^ G));
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
            const bf16x8 cur = a[s_ % AHEAD];
            if (s_ + AHEAD < KS) a[s_ % AHEAD] = *reinterpret_cast<const bf16x8*>(lb + ((32 * (s_ + AHEAD)) ^ G));
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur, qf[j][s_], acc[j], 0, 0, 0);
        }
        if constexpr ((SCAN_OPT & 32) != 0) {
            // The source reads a fragment AHEAD k-steps before its MFMAs, but the machine scheduler hoists ALL KS reads of the
            // quarter to its top (it has the registers) and waits for the last of them before the first MFMA: a quarter then
            // runs [16 LDS reads, one full wait] [32 MFMAs] [threshold scan] with nothing overlapped - 0.21 ms per pass with
            // DMA and hits compiled out, against 0.10 at the MFMA rate (profiles/r04_scan_elim.log, the ISA in DESIGN.md).
            // The schedule is therefore spelled out: AHEAD reads, then per k-step the NJ MFMAs followed by one read.
            __builtin_amdgcn_sched_group_barrier(0x100, AHEAD, 0);                       // DS reads
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_) {
                __builtin_amdgcn_sched_group_barrier(0x008, NJ, 0);                      // MFMAs of k-step s_
                if (s_ + AHEAD < KS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // the read AHEAD steps on
            }
        }
        // Threshold scan.  A hit is rare per element (~1.4e-3) but a taken branch per element costs more than the
        // MFMAs it follows, so four elements share one test (their maximum; NaN never wins) and the per-element code is
        // out of line.  (A single v_max3 tree + one test over all 16 elements of a tile was measured in a same-box A/B,
        // profiles/r02_scan_tree16_ab.log: 25 % SLOWER at 512 queries - 0.383 vs 0.307 ms - the four short independent
        // chains overlap with the MFMAs better than one deep one.)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float m4 = fmaxf(fmaxf(acc[j][4 * g], acc[j][4 * g + 1]), fmaxf(acc[j][4 * g + 2], acc[j][4 * g + 3]));
                if (__builtin_expect(__ballot(m4 >= tq[j]) == 0ull, 1)) continue;
                if (SCAN_OPT & 8) { asm volatile("" ::"v"(m4)); continue; }     // (diagnostic build: the test, never the hit path)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float sc = acc[j][4 * g + e];
                    const int p = prow + e + 8 * g;
                    const bool hit = sc >= tq[j] && (FULL || p < nrows_i);
                    const unsigned long long mask = __ballot(hit);
                    if (mask) {                                    // wave-uniform
                        if (hit) {
                            const int q = q0 + j * 32 + frow;
                            const unsigned long long key = make_key(sc, (uint32_t)p);
                            const int slot = wcount + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                           __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                            if (slot < SCAN_WHITS)
                                lds_store_hit_opaque(list_addr + slot * 8, key, list_addr + SCAN_WHITS * 8 + slot * 4, q);
                            else append(q, key);
                        }
                        wcount += __builtin_popcountll(mask);
                    }
                }
            }
    };

    // SCAN_OPT & 1.  The threshold test of ONE 4-element group of accumulator `p` (rows prow + e + 8 g of query tile j).
    auto scan_group = [&](auto full_tag, const f32x16& p, auto g_tag, float tqj, int qj0, int prow, uint32_t list_addr, int& wcount) {
        constexpr bool FULL = decltype(full_tag)::value;
        constexpr int g = decltype(g_tag)::value;
        const float m4 = fmaxf(fmaxf(p[4 * g], p[4 * g + 1]), fmaxf(p[4 * g + 2], p[4 * g + 3]));
        if (__builtin_expect(__ballot(m4 >= tqj) == 0ull, 1)) return;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sc = p[4 * g + e];
            const int pr = prow + e + 8 * g;
            const bool hit = sc >= tqj && (FULL || pr < nrows_i);
            const unsigned long long mask = __ballot(hit);
            if (mask) {                                            // wave-uniform
                if (hit) {
                    const int q = qj0 + frow;
                    const unsigned long long key = make_key(sc, (uint32_t)pr);
                    const int slot = wcount + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                   __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
                    if (slot < SCAN_WHITS)
                        lds_store_hit_opaque(list_addr + slot * 8, key, list_addr + SCAN_WHITS * 8 + slot * 4, q);
                    else append(q, key);
                }
                wcount += __builtin_popcountll(mask);
            }
        }
    };
    // One unit = 32 corpus rows x query tile J: the K loop into `cur`, with the four group tests of the previous unit's
    // accumulator `prv` placed behind MFMAs KS/4 - 1, 2 KS/4 - 1, ... (each test - three max, a compare, a ballot - runs in
    // the shadow of the MFMA in front of it; the wave is back at the next, dependent MFMA before that one has finished).
    auto chain_scan = [&](auto j_tag, auto scan_tag, auto full_tag, const unsigned char* lb, int G, f32x16& cur,
                          const f32x16& prv, float tprv, int qprv0, int prow_prv, uint32_t list_addr, int& wcount,
                          bf16x8 (&a)[AHEAD], const unsigned char* lb_next) {
        constexpr int J = decltype(j_tag)::value;
        constexpr bool SCAN = decltype(scan_tag)::value;
        constexpr bool CARRY = (SCAN_OPT & 16) != 0;               // fragments carried across units (see below)
#pragma unroll
        for (int r = 0; r < 16; ++r) cur[r] = 0.f;
        if constexpr (!CARRY) {
#pragma unroll
            for (int s_ = 0; s_ < AHEAD; ++s_) a[s_] = *reinterpret_cast<const bf16x8*>(lb + ((32 * s_) ^ G));
        }
        // (a compile-time index walk: with a `#pragma unroll` loop the compiler kept the K loop rolled around the group tests'
        //  branches and indexed the query fragments dynamically - 512 bytes of scratch per lane)
        static_for<KS>([&](auto s_tag) {
            constexpr int s_ = decltype(s_tag)::value;
            const bf16x8 c = a[s_ % AHEAD];
            if constexpr (s_ + AHEAD < KS) a[s_ % AHEAD] = *reinterpret_cast<const bf16x8*>(lb + ((32 * (s_ + AHEAD)) ^ G));
            // SCAN_OPT & 16: the NEXT unit's first fragments are requested behind this unit's last MFMAs (the swizzle term G
            // is the same for every 32-row quarter), so a unit no longer opens with an LDS round trip on an empty matrix pipe
            else if constexpr (CARRY) a[s_ % AHEAD] = *reinterpret_cast<const bf16x8*>(lb_next + ((32 * (s_ + AHEAD - KS)) ^ G));
            cur = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c, qf[J][s_], cur, 0, 0, 0);
            if constexpr (SCAN) {
                static_for<(4 * (s_ + 1)) / KS - (4 * s_) / KS>([&](auto g_tag) {
                    constexpr int g = (4 * s_) / KS + decltype(g_tag)::value;
                    __builtin_amdgcn_sched_barrier(0);             // the test stays BEHIND this MFMA (and in front of the next)
                    scan_group(full_tag, prv, std::integral_constant<int, g>{}, tprv, qprv0, prow_prv, list_addr, wcount);
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
        });
    };

    // SCAN_OPT & 64.  One quarter (32 rows x NJ query tiles, k-step major: a fragment feeds both tiles) into `cur`, with (i) the
    // NEXT quarter's first fragments requested behind its last MFMAs, so that no quarter opens with an LDS round trip on an empty
    // matrix pipe, and (ii) the threshold scan of the PREVIOUS quarter's accumulators `prv` placed between its MFMAs - one
    // group of tests behind every KS / 4 k-steps - instead of behind its last MFMA with the pipe idle.  Two accumulator sets
    // ping-pong (+32 registers, which the spelled-out read schedule of bit 32 had freed).
    auto step = [&](auto nj_tag, auto scan_tag, auto full_tag, const unsigned char* lq, const unsigned char* lnext, int G,
                    auto& cur, const auto& prv, int prow_prv, uint32_t list_addr, int& wcount, bf16x8 (&a)[AHEAD]) {
        constexpr int NJ = decltype(nj_tag)::value;
        constexpr bool SCAN = decltype(scan_tag)::value;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) cur[j][r] = 0.f;
        static_for<4>([&](auto seg_tag) {                          // four segments of KS / 4 k-steps, a test group behind each
            constexpr int SEG = decltype(seg_tag)::value;
            constexpr int S0 = SEG * KS / 4, S1 = (SEG + 1) * KS / 4;
            static_for<S1 - S0>([&](auto d_tag) {
                constexpr int s_ = S0 + decltype(d_tag)::value;
                const bf16x8 c = a[s_ % AHEAD];
                if constexpr (s_ + AHEAD < KS) a[s_ % AHEAD] = *reinterpret_cast<const bf16x8*>(lq + ((32 * (s_ + AHEAD)) ^ G));
                else a[s_ % AHEAD] = *reinterpret_cast<const bf16x8*>(lnext + ((32 * (s_ + AHEAD - KS)) ^ G));
#pragma unroll
                for (int j = 0; j < NJ; ++j) cur[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(c, qf[j][s_], cur[j], 0, 0, 0);
            });
#pragma unroll
            for (int d = 0; d < S1 - S0; ++d) {                    // this segment's schedule: per k-step its MFMAs, then its read
                __builtin_amdgcn_sched_group_barrier(0x008, NJ, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            if constexpr (SCAN) {
                __builtin_amdgcn_sched_barrier(0);
                static_for<NJ>([&](auto j_tag) {
                    constexpr int j = decltype(j_tag)::value;
                    scan_group(full_tag, prv[j], seg_tag, tq[j], q0 + j * 32, prow_prv, list_addr, wcount);
                });
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };

    // slot sequence of this workgroup: tiles bx, bx + nx, ... of 128 rows, each SPT = 128 / TR slots
    constexpr int SPT = SCAN_ROWS / TR;
    const int my_tiles = bx < ntiles ? (ntiles - bx + nx - 1) / nx : 0;
    const int nslots = my_tiles * SPT;
    auto slot_row0 = [&](int i) { return (bx + (i / SPT) * nx) * SCAN_ROWS + (i % SPT) * TR; };   // < 2^31 (entry point)
#pragma unroll
    for (int i = 0; i < NSLOT - 1; ++i)
        if (i < nslots) dma(slot_row0(i), i);
    // Thresholds.  Batches of <= 8 queries (gm != nullptr): no threshold launch - while the first tile is in flight wave w
    // computes tau of query w from the sample's group maxima (wave_tau: a deterministic function of gm, so every workgroup
    // arrives at the same value), workgroup 0 publishes it for the finalize.  Columns >= nq get NaN: no score compares >= it.
    __shared__ float tau_sh[8];
    if (gm != nullptr) {
        if (w < nq) {
            const float tw = wave_tau(gm + (long long)w * ldm, gm_n, rank, lane);
            if (lane == 0) {
                tau_sh[w] = tw;
                if (bx == 0) tau[w] = tw;
            }
        }
    }
    if (NSLOT == 2 || nslots < NSLOT - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSLOT - 2) * PASSES) : "memory");      // slot 0 (the oldest) has landed
    __syncthreads();                                               // first slot landed, tau_sh written
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = q0 + j * 32 + frow;
        tq[j] = (q < nq) ? (gm != nullptr ? tau_sh[q] : tau[q]) : __builtin_nanf("");
    }
    int it = 0, wprev = 0;                                         // wprev: hits of the previous slot awaiting their append
    for (; it < nslots; ++it) {
        const int buf = it % NSLOT, lst = it & 1;
        const int prow0 = slot_row0(it);
        // previous slot's hits: slots from the LDS counters, keys to the segment - issued BEFORE this slot's DMA so that the
        // end-of-slot wait for the DMA (vmcnt counts in order) never waits for these stores
        const unsigned char* lprev = hitbase + (lst ^ 1) * LIST_STRIDE;
        const int npend = wprev < SCAN_WHITS ? wprev : SCAN_WHITS;
        for (int h = lane; h < npend; h += 64)
            append(reinterpret_cast<const int*>(lprev + SCAN_WHITS * 8)[h],
                   reinterpret_cast<const unsigned long long*>(lprev)[h]);
        // (the DMA goes out right after the barrier: issuing it behind the first quarter's MFMAs, the move that gained 4 % in
        // the row-owner kernel, cost 30 % here - 0.355 against 0.268 ms at 512 queries, profiles/r03_scan_late_dma_ab.log: with a
        // two-deep ring the next tile needs the whole of this tile's compute time to land)
        // the slot read NSLOT - 1 iterations from now: its ring slot was read in the previous iteration
        if (it + NSLOT - 1 < nslots) dma(slot_row0(it + NSLOT - 1), (it + NSLOT - 1) % NSLOT);
        int wcount = 0;
        if (active) {
            // fragment address = lane row base + compile-time row offset + ((32 s) ^ G): the swizzle term depends on
            // the lane only through G.  G is made opaque per 32-row step so that the fragment addresses are recomputed
            // (one v_xor each) instead of being hoisted out of the loops into live registers (spills at KS = 16).
            int G = (fh ^ ((frow >> FS) & FM)) << 4;
            const unsigned char* lb = tilebuf + (size_t)buf * TILE_CHUNKS * 16 + frow * CPR * 16;
            const uint32_t list_addr =
                (uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)(hitbase + lst * LIST_STRIDE);
            const bool full = prow0 + TR <= nrows_i;
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>;
            if ((SCAN_OPT & 1) && second) {
                // units (rq, j = 0), (rq, 1), (rq + 1, 0) ...: accumulators acc0 / acc1 ping-pong, each unit's MFMAs carry the
                // threshold scan of the unit before it; the tile's last unit is scanned on its own
                f32x16 acc0, acc1;
                const int qa = q0, qb = q0 + 32;
                auto tile_units = [&](auto full_tag) {
                    asm volatile("" : "+v"(G));
                    bf16x8 afr[AHEAD];
                    const unsigned char* l0 = lb + rq_begin * 32 * CPR * 16;
                    if constexpr ((SCAN_OPT & 16) != 0) {
#pragma unroll
                        for (int s_ = 0; s_ < AHEAD; ++s_) afr[s_] = *reinterpret_cast<const bf16x8*>(l0 + ((32 * s_) ^ G));
                    }
                    chain_scan(I0{}, std::false_type{}, full_tag, l0, G, acc0, acc1, 0.f, 0, 0, list_addr, wcount, afr, l0);
#pragma unroll 1
                    for (int rq = rq_begin; rq < rq_end; ++rq) {
                        asm volatile("" : "+v"(G));
                        const int prow = prow0 + rq * 32 + 4 * fh;
                        const unsigned char* lq = lb + rq * 32 * CPR * 16;
                        // (the last unit of the slot prefetches its own quarter again: four harmless reads instead of a branch)
                        const unsigned char* ln = rq + 1 < rq_end ? lq + 32 * CPR * 16 : lq;
                        chain_scan(I1{}, std::true_type{}, full_tag, lq, G, acc1, acc0, tq[0], qa, prow, list_addr, wcount, afr, ln);
                        if (rq + 1 < rq_end) {
                            chain_scan(I0{}, std::true_type{}, full_tag, ln, G, acc0, acc1, tq[1], qb, prow, list_addr, wcount,
                                       afr, ln);
                        } else {
                            static_for<4>([&](auto g_tag) { scan_group(full_tag, acc1, g_tag, tq[1], qb, prow, list_addr, wcount); });
                        }
                    }
                };
                if (full) tile_units(std::true_type{});
                else tile_units(std::false_type{});
            } else if constexpr ((SCAN_OPT & 32) != 0) {
                // one quarter LOOP per (query tiles, full slot) variant: with the variant chosen inside the loop the compiler
                // merged the four variants' identical fragment reads in front of the branch - in a different basic block than
                // their MFMAs, where no schedule can interleave them
                auto quarters = [&](auto nj_tag, auto full_tag) {
                    if constexpr ((SCAN_OPT & 64) != 0) {
                        constexpr int NJ = decltype(nj_tag)::value;
                        constexpr int QB = 32 * CPR * 16;                      // LDS bytes of a 32-row quarter
                        f32x16 accA[NJ], accB[NJ];
                        bf16x8 afr[AHEAD];
                        asm volatile("" : "+v"(G));
                        const unsigned char* l0 = lb + rq_begin * QB;
                        const int pr0 = prow0 + 4 * fh;
#pragma unroll
                        for (int s_ = 0; s_ < AHEAD; ++s_) afr[s_] = *reinterpret_cast<const bf16x8*>(l0 + ((32 * s_) ^ G));
                        // (the slot's last quarter prefetches its own first fragments again: four harmless reads, no branch)
                        step(nj_tag, std::false_type{}, full_tag, l0, rq_begin + 1 < rq_end ? l0 + QB : l0, G, accA, accB, 0,
                             list_addr, wcount, afr);
                        int rq = rq_begin + 1;
#pragma unroll 1
                        for (; rq + 1 < rq_end; rq += 2) {                     // quarters rq (into B) and rq + 1 (into A)
                            asm volatile("" : "+v"(G));
                            const unsigned char* lq = lb + rq * QB;
                            step(nj_tag, std::true_type{}, full_tag, lq, lq + QB, G, accB, accA, pr0 + (rq - 1) * 32, list_addr,
                                 wcount, afr);
                            step(nj_tag, std::true_type{}, full_tag, lq + QB, rq + 2 < rq_end ? lq + 2 * QB : lq + QB, G, accA, accB,
                                 pr0 + rq * 32, list_addr, wcount, afr);
                        }
                        if (rq < rq_end) {
                            asm volatile("" : "+v"(G));
                            const unsigned char* lq = lb + rq * QB;
                            step(nj_tag, std::true_type{}, full_tag, lq, lq, G, accB, accA, pr0 + (rq - 1) * 32, list_addr, wcount, afr);
                            static_for<NJ>([&](auto j_tag) {
                                constexpr int j = decltype(j_tag)::value;
                                static_for<4>([&](auto g_tag) {
                                    scan_group(full_tag, accB[j], g_tag, tq[j], q0 + j * 32, pr0 + rq * 32, list_addr, wcount);
                                });
                            });
                        } else {
                            static_for<NJ>([&](auto j_tag) {
                                constexpr int j = decltype(j_tag)::value;
                                static_for<4>([&](auto g_tag) {
                                    scan_group(full_tag, accA[j], g_tag, tq[j], q0 + j * 32, pr0 + (rq - 1) * 32, list_addr, wcount);
                                });
                            });
                        }
                        return;
                    }
#pragma unroll 1
                    for (int rq = rq_begin; rq < rq_end; ++rq) {
                        asm volatile("" : "+v"(G));
                        quarter(nj_tag, full_tag, lb + rq * 32 * CPR * 16, G, prow0 + rq * 32 + 4 * fh, list_addr, wcount);
                    }
                };
                if (second) {
                    if (full) quarters(I2{}, std::true_type{});
                    else      quarters(I2{}, std::false_type{});
                } else {
                    if (full) quarters(I1{}, std::true_type{});
                    else      quarters(I1{}, std::false_type{});
                }
            } else
#pragma unroll 1
            for (int rq = rq_begin; rq < rq_end; ++rq) {
                asm volatile("" : "+v"(G));
                const int prow = prow0 + rq * 32 + 4 * fh;
                const unsigned char* lq = lb + rq * 32 * CPR * 16;
                if (second) {
                    if (full) quarter(I2{}, std::true_type{}, lq, G, prow, list_addr, wcount);
                    else      quarter(I2{}, std::false_type{}, lq, G, prow, list_addr, wcount);
                } else {
                    if (full) quarter(I1{}, std::true_type{}, lq, G, prow, list_addr, wcount);
                    else      quarter(I1{}, std::false_type{}, lq, G, prow, list_addr, wcount);
                }
            }
        }
        wprev = wcount;
        // my share of the NEXT slot has landed, my hit stores are done.  Two slots: the DMA issued above is the youngest
        // operation and is waited for.  Four slots: only that DMA (PASSES instructions) may stay in flight - everything older,
        // the next slot's DMA and the appends included, is complete.
        if (NSLOT == 2 || !(it + NSLOT - 1 < nslots)) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PASSES) : "memory");
        __syncthreads();                 // this slot consumed by every wave, the next one visible
    }
    // the last slot's hits
    const unsigned char* llast = hitbase + ((it & 1) ^ 1) * LIST_STRIDE;
    const int nlast = wprev < SCAN_WHITS ? wprev : SCAN_WHITS;
    for (int h = lane; h < nlast; h += 64)
        append(reinterpret_cast<const int*>(llast + SCAN_WHITS * 8)[h], reinterpret_cast<const unsigned long long*>(llast)[h]);
    __syncthreads();
    if (tid < nqg) segcnt[(long long)(qbase + tid) * nx + bx] = lcnt[tid];      // every (query, segment) count is written: no memset
}

// ---- fused query conversion + sample pass of the mixed search (dims 32 / 64 / 128 / 256) ---------------------------
// One launch instead of three (bf16_rows_kernel for the queries, a generic-tile GEMM that stored the DENSE sample score
// matrix - 94 MB at 512 queries x 46k sample rows - and the threshold kernel that read it back): the same streaming
// structure as scan_filter_kernel over an evenly spaced subset of 128-row corpus tiles, with
//  * the queries converted fp32 -> bf16 in the prologue (same rounding as bf16_rows_kernel; the workgroups with bx == 0
//    also write the bf16 rows the corpus pass reads, and every workgroup clears its share of the search's counters);
//  * the epilogue keeping only MAXIMA: per (query, 16- / 4- / 1-row group) one float in gm[nq][ldm] - the threshold only
//    needs the r-th largest of ~256 group maxima of the sample (see wave_tau), and a maximum of maxima is the maximum.
// SUB = values per lane, query tile and 32-row quarter: 1 (16-row groups), 4 (4-row groups), 16 (every score: corpora so
// small that coarser groups would leave fewer than ~1000 columns).  Column of a value: ((tile * 4 + quarter) * 2 + fh) * SUB + i.
template <int KS, int SUB>
__global__ __launch_bounds__(512, 1) void sample_max_kernel(const uint16_t* __restrict__ X16, long long ld16, int tstride,
                                                            int n_tiles, const float* __restrict__ Q, long long ldq, int nq,
                                                            uint16_t* __restrict__ Q16, float* __restrict__ gm, long long ldm,
                                                            int nx, int* zero, long long n_zero) {
    constexpr int CPR = 2 * KS, FM = CPR < 16 ? CPR - 1 : 15, FS = CPR >= 16 ? 0 : (CPR == 8 ? 1 : 2);
    constexpr int TILE_CHUNKS = SCAN_ROWS * CPR, PASSES = TILE_CHUNKS / 512, AHEAD = KS < 4 ? KS : 4, RPP = 512 / CPR;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];                      // [2][TILE_CHUNKS * 16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 31, fh = lane >> 5;
    const int bx = blockIdx.x % nx, by = blockIdx.x / nx;
    for (long long i = (long long)blockIdx.x * 512 + tid; i < n_zero; i += (long long)gridDim.x * 512) zero[i] = 0;
    // the same work split as the corpus pass: waves without queries of their own take a share of the tile's rows
    const int nqg = nq - by * SCAN_QGROUP < SCAN_QGROUP ? nq - by * SCAN_QGROUP : SCAN_QGROUP;
    const int qsh = nqg <= 64 ? 0 : (nqg <= 128 ? 1 : (nqg <= 256 ? 2 : 3));
    const int parts = qsh <= 1 ? 4 : (qsh == 2 ? 2 : 1);
    const int part = w >> qsh;
    const int q0 = by * SCAN_QGROUP + (w & ((1 << qsh) - 1)) * 64;
    const bool active = q0 < nq && part < parts;
    const int rq_begin = part * (4 / parts), rq_end = rq_begin + 4 / parts;
    const bool second = q0 + 32 < nq;

    // query fragments: 8 consecutive floats -> 8 bf16 (round to nearest even, NaN kept: bf16_rows_kernel's rounding)
    bf16x8 qf[2][KS];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = q0 + j * 32 + frow;
        const int qc = q < nq ? q : nq - 1;
        const bool wr = bx == 0 && part == 0 && q < nq;              // exactly one lane pair per (query, k-step) writes
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_) {
            const float* src = Q + (long long)qc * ldq + (2 * s_ + fh) * 8;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
            uint32_t h[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = e < 4 ? v0[e & 3] : v1[e & 3];
                const uint32_t u = __float_as_uint(v);
                h[e] = (v != v) ? 0x7fc0u : ((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
            }
            const u32x4 pk{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)};
            qf[j][s_] = __builtin_bit_cast(bf16x8, pk);
            if (wr && q0 + j * 32 < nq) *reinterpret_cast<u32x4*>(Q16 + (long long)q * (16 * KS) + (2 * s_ + fh) * 8) = pk;
        }
    }
    // tile DMA: the lane map of scan_filter_kernel (sample tiles are whole tiles: no clamped path)
    const int drow = tid / CPR;
    const uint32_t lane_off = (uint32_t)(drow * (int)ld16 * 2 + (((tid % CPR) ^ ((drow >> FS) & FM)) * 16));
    auto dma = [&](int st_, int buf) {
        const long long row0 = (long long)st_ * tstride * SCAN_ROWS;
        unsigned char* lbase = lds + (size_t)buf * TILE_CHUNKS * 16 + (size_t)(w * 64) * 16;
        const unsigned char* gb = reinterpret_cast<const unsigned char*>(X16) + row0 * ld16 * 2;
#pragma unroll
        for (int u = 0; u < PASSES; ++u) {
            const unsigned char* g = gb + (long long)u * RPP * ld16 * 2 + lane_off;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(lbase + (size_t)u * 512 * 16), 16, 0, 0);
        }
    };
    // both buffers are requested up front (a workgroup of a 1M-row search has one or two tiles: their latencies overlap);
    // from the third tile on a buffer is re-filled as soon as every wave has left it
    int t = bx;
    if (t < n_tiles) dma(t, 0);
    if (t + nx < n_tiles) {
        dma(t + nx, 1);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PASSES) : "memory");     // the first tile's pieces (issued first) landed
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    for (int it = 0; t < n_tiles; t += nx, ++it) {
        const int buf = it & 1;
        if (it > 0 && t + nx < n_tiles) dma(t + nx, buf ^ 1);
        if (active) {
            int G = (fh ^ ((frow >> FS) & FM)) << 4;
            const unsigned char* lb = lds + (size_t)buf * TILE_CHUNKS * 16 + frow * CPR * 16;
            // one quarter loop per number of live query tiles, the LDS-read / MFMA interleave spelled out (scan_filter_kernel,
            // SCAN_OPT 32: with `if (second)` inside the K loop every k-step ended a basic block)
            auto quarters = [&](auto nj_tag) {
                constexpr int NJ = decltype(nj_tag)::value;
#pragma unroll 1
                for (int rq = rq_begin; rq < rq_end; ++rq) {
                    asm volatile("" : "+v"(G));
                    const unsigned char* lq = lb + rq * 32 * CPR * 16;
                    f32x16 acc[NJ];
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
                    bf16x8 a[AHEAD];
#pragma unroll
                    for (int s_ = 0; s_ < AHEAD; ++s_) a[s_] = *reinterpret_cast<const bf16x8*>(lq + ((32 * s_) ^ G));
#pragma unroll
                    for (int s_ = 0; s_ < KS; ++s_) {
                        const bf16x8 cur = a[s_ % AHEAD];
                        if (s_ + AHEAD < KS) a[s_ % AHEAD] = *reinterpret_cast<const bf16x8*>(lq + ((32 * (s_ + AHEAD)) ^ G));
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur, qf[j][s_], acc[j], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, AHEAD, 0);
#pragma unroll
                    for (int s_ = 0; s_ < KS; ++s_) {
                        __builtin_amdgcn_sched_group_barrier(0x008, NJ, 0);
                        if (s_ + AHEAD < KS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    const long long col0 = (((long long)t * 4 + rq) * 2 + fh) * SUB;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int q = q0 + j * 32 + frow;
                        if (q >= nq) continue;
                        float* dst = gm + (long long)q * ldm + col0;
                        if (SUB == 16) {
#pragma unroll
                            for (int g = 0; g < 4; ++g)
                                *reinterpret_cast<f32x4*>(dst + 4 * g) = f32x4{acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
                        } else {
                            float m4[4];
#pragma unroll
                            for (int g = 0; g < 4; ++g) {       // v > m ? v : m: a NaN score never wins (sample_threshold_kernel's rule)
                                float m = acc[j][4 * g];
#pragma unroll
                                for (int e = 1; e < 4; ++e) m = acc[j][4 * g + e] > m ? acc[j][4 * g + e] : m;
                                m4[g] = m;
                            }
                            if (SUB == 4) {
                                *reinterpret_cast<f32x4*>(dst) = f32x4{m4[0], m4[1], m4[2], m4[3]};
                            } else {
                                float m = m4[0];
#pragma unroll
                                for (int g = 1; g < 4; ++g) m = m4[g] > m ? m4[g] : m;
                                *dst = m;
                            }
                        }
                    }
                }
            };
            if (second) quarters(std::integral_constant<int, 2>{});
            else quarters(std::integral_constant<int, 1>{});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

// tau[q] for every query from the group maxima: one wave per query, eight queries per workgroup
__global__ __launch_bounds__(512) void tau_from_maxima_kernel(const float* gm, long long ldm, long long n, int r, int nq,
                                                              float* tau) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = blockIdx.x * 8 + w;
    if (q >= nq) return;
    const float t = wave_tau(gm + (long long)q * ldm, n, r, lane);
    if (lane == 0) tau[q] = t;
}

// segments per query = workgroups per 512-query group of the streaming pass; each owns CAND_CAP / nseg candidate slots
static inline int scan_segments(long long nrows, long long nq) {
    const long long ntiles = (nrows + SCAN_ROWS - 1) / SCAN_ROWS;
    const long long ny = (nq + SCAN_QGROUP - 1) / SCAN_QGROUP;
    long long nx = 256 / ny;                       // one workgroup per CU
    if (nx < 1) nx = 1;
    if (nx > ntiles) nx = ntiles;
    return (int)(nx < 1 ? 1 : nx);
}

template <int KS>
static hipError_t launch_scan(const uint16_t* X16, long long ld16, long long nrows, const uint16_t* Q16, int nq,
                              float* tau, unsigned long long* cand, int* segcnt, int* ocnt, hipStream_t st,
                              const float* gm = nullptr, long long ldm = 0, long long gm_n = 0, int rank = 0) {
    auto kern = scan_filter_kernel<KS>;
    constexpr size_t lds_bytes = 2ull * SCAN_ROWS * 2 * KS * 16 + SCAN_HIT_BYTES + SCAN_QGROUP * 4;
    static_assert(lds_bytes <= 160 * 1024, "LDS budget");
    static PerDeviceOnce attr_done;
    if (attr_done.pending()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        attr_done.mark();
    }
    const int ny = (nq + SCAN_QGROUP - 1) / SCAN_QGROUP;
    const int nx = scan_segments(nrows, nq);
    const int d = 16 * KS;
    ProfScope prof("search_filter_stream128x512_bf16", 2.0 * (double)nrows * (double)nq * d,
                   2.0 * ((double)nrows * d * ny + (double)nq * d), st);
    hipLaunchKernelGGL(kern, dim3((unsigned)(nx * ny)), dim3(512), lds_bytes, st, X16, ld16, nrows, Q16, nq, tau, cand, segcnt,
                       ocnt, CAND_CAP / nx, nx, gm, ldm, gm_n, rank);
    return hipGetLastError();
}

// the fused conversion + sample launch: n_tiles evenly spaced 128-row tiles -> gm[nq][ldm], ldm = columns = n_tiles * 8 * SUB
struct SamplePlan {
    int n_tiles, tstride, sub;
    long long cols;
};
static inline SamplePlan sample_plan(long long nrows, long long n_sample_rows) {
    SamplePlan sp;
    const long long full = nrows / SCAN_ROWS;                          // whole tiles only
    long long nt = (n_sample_rows + SCAN_ROWS - 1) / SCAN_ROWS;
    if (nt > full) nt = full;
    if (nt < 1) nt = 1;
    sp.n_tiles = (int)nt;
    sp.tstride = (int)(full / nt);
    const long long rows = nt * SCAN_ROWS;
    sp.sub = rows / 16 >= 1024 ? 1 : (rows / 4 >= 1024 ? 4 : 16);      // keep >= ~1000 columns where the sample allows
    sp.cols = nt * 8 * sp.sub;
    return sp;
}
template <int KS>
static hipError_t launch_sample(const uint16_t* X16, long long ld16, const SamplePlan& sp, const float* Q, long long ldq,
                                int nq, uint16_t* Q16, float* gm, int* zero, long long n_zero, hipStream_t st) {
    constexpr size_t lds_bytes = 2ull * SCAN_ROWS * 2 * KS * 16;
    static PerDeviceOnce attr_done;
    if (attr_done.pending()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sample_max_kernel<KS, 1>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(sample_max_kernel<KS, 4>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(sample_max_kernel<KS, 16>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        attr_done.mark();
    }
    const int ny = (nq + SCAN_QGROUP - 1) / SCAN_QGROUP;
    int nx = 256 / ny;
    nx = nx < 1 ? 1 : (nx > sp.n_tiles ? sp.n_tiles : nx);
    const int d = 16 * KS;
    ProfScope prof("search_sample_max128x512_bf16", 2.0 * (double)sp.n_tiles * SCAN_ROWS * (double)nq * d,
                   2.0 * (double)sp.n_tiles * SCAN_ROWS * d * ny + 4.0 * (double)nq * d + 4.0 * (double)nq * sp.cols, st);
    const dim3 grid((unsigned)(nx * ny)), block(512);
    if (sp.sub == 1)
        hipLaunchKernelGGL((sample_max_kernel<KS, 1>), grid, block, lds_bytes, st, X16, ld16, sp.tstride, sp.n_tiles, Q, ldq, nq,
                           Q16, gm, sp.cols, nx, zero, n_zero);
    else if (sp.sub == 4)
        hipLaunchKernelGGL((sample_max_kernel<KS, 4>), grid, block, lds_bytes, st, X16, ld16, sp.tstride, sp.n_tiles, Q, ldq, nq,
                           Q16, gm, sp.cols, nx, zero, n_zero);
    else
        hipLaunchKernelGGL((sample_max_kernel<KS, 16>), grid, block, lds_bytes, st, X16, ld16, sp.tstride, sp.n_tiles, Q, ldq, nq,
                           Q16, gm, sp.cols, nx, zero, n_zero);
    return hipGetLastError();
}

// fp32 rows -> bf16 rows (round to nearest even, NaN kept), one wave per row; optionally max_norm[0] = the maximum row norm
// and max_norm[1] = the maximum ROUNDING-ERROR norm max_r ||x_r - bf16(x_r)||_2, the two corpus-side terms of the search's
// error bound (atomic max over the float bits: norms are >= 0 so the unsigned order is the float order; NaN/inf propagate
// and make every certificate fail -> exact fix-up path).
__global__ __launch_bounds__(256) void bf16_rows_kernel(const float* x, long long rows, long long ld, int d,
                                                        uint16_t* out, long long ld_out, float* max_norm,
                                                        int* zero = nullptr, long long n_zero = 0) {
    // optional: clear n_zero ints (the search's counters, so that a search call needs no separate memset launch)
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_zero; i += (long long)gridDim.x * 256) zero[i] = 0;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    float ss = 0.f, ds = 0.f;
    for (int c = lane; c < (d >> 2); c += 64) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ld + 4 * c);
        uint32_t h[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t u = __float_as_uint(v[e]);
            h[e] = (v[e] != v[e]) ? 0x7fc0u : ((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
            ss += v[e] * v[e];
            const float dv = v[e] - __uint_as_float(h[e] << 16);       // exact: the rounding error of this component
            ds += dv * dv;
        }
        uint2 pk{h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
        *reinterpret_cast<uint2*>(out + r * ld_out + 4 * c) = pk;
    }
    if (max_norm) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            ss += __shfl_xor(ss, o, 64);
            ds += __shfl_xor(ds, o, 64);
        }
        if (lane == 0) {
            // the maxima only grow: look first, update only when this row raises one.  An unconditional atomicMax per row
            // serialised 2 x 10M device-scope atomics on two addresses: 227 ms for a 10M-row shadow that moves 15 GB
            // (profiles/r04_ivf10m_kernel_stats.csv), 68 GB/s
            unsigned int* m = reinterpret_cast<unsigned int*>(max_norm);
            const unsigned int a = __float_as_uint(sqrtf(ss));
            // rounded UP a little: the sum above is itself an fp32 evaluation
            const unsigned int b = __float_as_uint(sqrtf(ds) * 1.0001f);
            if (__hip_atomic_load(m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a) atomicMax(m, a);
            if (__hip_atomic_load(m + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < b) atomicMax(m + 1, b);
        }
    }
}

// Error bound of the bf16 pass, from MEASURED rounding errors (rigorous, and ~4x tighter than the worst case on real data):
// with q~ = bf16(q), x~ = bf16(x), dq = q~ - q, dx = x~ - x:   q~.x~ - q.x = dq.x~ + q.dx   (an identity), hence
//   |approx - exact| <= ||dq|| ||x~|| + ||q|| ||dx|| <= ||dq|| (M + D) + ||q|| D,    M = max_r ||x_r||, D = max_r ||dx_r||
// (M, D accumulated by amdrec_bf16_rows when the shadow is written; ||dq|| computed here from the query itself), plus
// 2 d 2^-24 ||q|| (M + D) for the two fp32 accumulations being compared.  The worst case of this bound is the classical
// (2u + u^2) ||q|| M with u = 2^-8 (every component half an ulp off: tests/test_search_gpu.py::
// test_mixed_worst_case_rounding, where round 1's u = 2^-9 constant returned a wrong top-k); on random unit vectors
// ||dq|| ~ ||dx|| ~ 0.0008, i.e. eps ~ 0.0018 instead of 0.0078: half as many candidates survive the prune and are re-scored.
// (eps_bound itself lives in topk_utils.hpp: the IVF scan's bf16 prefilter uses the same bound)

// step 4 of the mixed-precision search: select a_k -> prune -> fp32 re-score -> exact sort -> certificate, one workgroup per
// query.  The candidate keys (<= CAPK / NT + overflow share per thread) stay in registers for the selection: a 3-pass radix
// select of the k-th largest approximate score (LDS histograms), then only the survivors of the pruning rule go to LDS, are
// re-scored and sorted (typically ~1.25 k keys instead of the whole list at k = 500).
// Small batches (<= FUSED_MAX_NQ queries) take finalize_fused_kernel below instead: several workgroups per query.
// Shapes <NT threads, CAPK candidates>: <512, 8192> is the general one (two workgroups per CU: 64 KB of keys each).  The
// sharded search asks for SHORT lists of MANY queries (8 ranks: 4096 queries x k = 128, ~400 candidates each), where a
// workgroup's time is a chain of dependent steps (loads, three histogram passes, re-score rounds, sort), not work: 33 us per
// query and only two chains in flight per CU - 0.26 ms of a rank's 0.68 ms search.  <128, 1024> and <256, 2048> keep 8 / 4
// workgroups per CU in flight (16 waves either way); a query with more candidates than CAPK goes to the exact fix-up scan
// (the host picks a shape with twice the expected count, 10 sigma of the threshold estimate).
constexpr int FUSED_MAX_NQ = 128;
// Candidate input: block q of `cand` (cstride keys) = nseg segments of seg_cap slots, slot-major (slot s of segment g at
// s * nseg + g) [+ an overflow block at CAND_CAP when the streaming pass produced it]; segcnt[q][nseg] = hits each segment
// saw (may exceed seg_cap: the excess went to the overflow block, ocnt[q] entries).  The generic GEMM pass writes one
// segment (nseg = 1, seg_cap = CAND_CAP, no overflow).
template <int NT, int CAPK>
__global__ __launch_bounds__(NT, NT == 512 ? 2 : 4) void finalize_mixed_kernel(const unsigned long long* cand, long long cstride,
                                                                       const int* segcnt, int nseg, int seg_cap, const int* ocnt,
                                                                       int k, long long nrows, const float* tau,
                                                                       const float* max_norm, const float* X, long long ldx,
                                                                       int d, const float* Q, long long ldq, int* fail,
                                                                       float* outD, long long* outI, long long pos_offset) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];      // [CAPK] then qv[d]
    float* qv = reinterpret_cast<float*>(keys + CAPK);
    __shared__ int hist[2048];
    __shared__ int scratch[NT + 2];
    __shared__ float red[16];
    __shared__ int m_sh;
    __shared__ int seg_n[256], tot_sh, lost_sh, maxn_sh, fill_sh;
    constexpr bool GENERAL = CAPK == CAND_CAP;                // threads read the slot-major candidate area directly
    constexpr int NW = NT / 64;
    constexpr int PER = CAPK / NT;                            // candidate keys per thread
    constexpr int OVP = GENERAL ? 2048 / NT : 0;              // + overflow keys per thread (general shape: up to 2048 per query)
    static_assert(!GENERAL || NT == 512, "the general shape is 512 threads");
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int need = (int)(nrows < k ? nrows : k);
    auto give_up = [&]() {
        if (tid == 0) {
            fail[q] = 1;
            atomicAdd(&fail[gridDim.x], 1);
        }
    };
    // segment counts: valid slots per segment, their total, and the hits that did not fit
    if (tid == 0) { tot_sh = 0; lost_sh = 0; maxn_sh = 0; fill_sh = 0; }
    __syncthreads();
    for (int g0 = 0; g0 < nseg; g0 += NT) {                   // (nseg <= 256)
        int sv = 0, lost = 0;
        if (g0 + tid < nseg) {
            const int v = segcnt[(long long)q * nseg + g0 + tid];
            sv = v < seg_cap ? v : seg_cap;
            lost = v - sv;
            seg_n[g0 + tid] = sv;
            if (!GENERAL && sv) atomicMax(&maxn_sh, sv);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sv += __shfl_xor(sv, o, 64);
            lost += __shfl_xor(lost, o, 64);
        }
        if (lane == 0) {
            if (sv) atomicAdd(&tot_sh, sv);
            if (lost) atomicAdd(&lost_sh, lost);
        }
    }
    const int oc = ocnt[q];
    __syncthreads();
    const int c = tot_sh + oc;
    // every hit must be in a segment or in the overflow block, and the list must be usable (block-uniform tests)
    if (lost_sh != oc || oc > 2048 || c < need || c > CAPK) { give_up(); return; }
    float ss = 0.f, ds = 0.f;
    for (int i = tid; i < d; i += NT) {
        const float v = Q[(long long)q * ldq + i];
        qv[i] = v;
        ss += v * v;
        const uint32_t u = __float_as_uint(v);                    // the same rounding as bf16_rows_kernel applied to the query
        const float dv = v - __uint_as_float(((u + 0x7fffu + ((u >> 16) & 1u)) >> 16) << 16);
        ds += dv * dv;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ss += __shfl_xor(ss, o, 64);
        ds += __shfl_xor(ds, o, 64);
    }
    if (lane == 0) { red[w] = ss; red[8 + w] = ds; }
    if (tid == 0) m_sh = 0;
    const unsigned long long* blk = cand + (long long)q * cstride;
    const int nslots = nseg * seg_cap;
    const int seg_shift = (nseg & (nseg - 1)) == 0 ? 31 - __builtin_clz((unsigned)nseg) : -1;      // nseg a power of two
    unsigned long long mine[PER + OVP];
    if constexpr (GENERAL) {
        // the candidates stay where the pass put them: thread <- slots tid + NT j of the segment area, validity from the
        // segment's count (an empty slot is key 0, which no hit can be: its score would have to be NaN)
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int s_ = tid + NT * j;                       // slot-major: element (slot, segment) at slot * nseg + segment
            bool ok = false;
            if (s_ < nslots) {
                const int slot = seg_shift >= 0 ? s_ >> seg_shift : s_ / nseg;
                ok = slot < seg_n[s_ - slot * nseg];
            }
            mine[j] = ok ? blk[s_] : 0ull;
        }
#pragma unroll
        for (int j = 0; j < OVP; ++j) {
            const int i = tid + NT * j;
            mine[PER + j] = (i < oc) ? blk[CAND_CAP + i] : 0ull;
        }
        __syncthreads();
    } else {
        // short lists: the occupied prefix of the slot-major area (slots below the fullest segment's count) and the overflow
        // block are compacted through LDS, then dealt to the threads' registers
        const int lim = maxn_sh * nseg;
        for (int s_ = tid; s_ < lim; s_ += NT) {
            const int slot = seg_shift >= 0 ? s_ >> seg_shift : s_ / nseg;
            if (slot < seg_n[s_ - slot * nseg]) keys[atomicAdd(&fill_sh, 1)] = blk[s_];
        }
        for (int i = tid; i < oc; i += NT) keys[atomicAdd(&fill_sh, 1)] = blk[CAND_CAP + i];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PER; ++j) mine[j] = (tid + NT * j < c) ? keys[tid + NT * j] : 0ull;
        __syncthreads();                                      // keys[] is reused for the survivors
    }
    float qn = 0.f, dqn = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) { qn += red[i]; dqn += red[8 + i]; }
    const float eps = eps_bound(sqrtf(qn), sqrtf(dqn) * 1.0001f, max_norm[0], max_norm[1], d);
    if (!(eps < INFINITY)) { give_up(); return; }            // NaN / inf norms: exact path (block-uniform)
    if (need == 0) {
        write_result(keys, 0, k, q, outD, outI, pos_offset);
        return;
    }
    // a_k = need-th largest approximate score (orderable 32-bit image), radix select 11 + 11 + 10 bits
    uint32_t prefix = 0u, pmask = 0u;
    int rr = need;
    const int shifts[3] = {21, 10, 0};
    const int nbits[3] = {11, 11, 10};
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
        for (int i = tid; i < 2048; i += NT) hist[i] = 0;
        __syncthreads();
        const uint32_t bm = (1u << nbits[pass]) - 1u;
#pragma unroll
        for (int j = 0; j < PER + OVP; ++j) {
            const uint32_t u = (uint32_t)(mine[j] >> 32);
            if (mine[j] != 0ull && (u & pmask) == prefix) atomicAdd(&hist[(u >> shifts[pass]) & bm], 1);
        }
        __syncthreads();
        const int bin = find_bin_desc<2048, NT>(hist, rr, scratch);       // always found: c >= need >= rr
        prefix |= (uint32_t)bin << shifts[pass];
        pmask |= bm << shifts[pass];
    }
    // prune: a row with approx < a_k - 2 eps has exact < a_k - eps <= exact of each of the k best-by-approx rows
    const float cut = f32_from_orderable(prefix) - 2.f * eps;
#pragma unroll
    for (int j = 0; j < PER + OVP; ++j)
        if (mine[j] != 0ull && key_score(mine[j]) >= cut) keys[atomicAdd(&m_sh, 1)] = mine[j];
    __syncthreads();
    const int m = m_sh;                                       // need <= m <= c
    // fp32 re-score, one wave per candidate, 8 candidates (random 1 KB rows: latency-bound) in flight per wave; two
    // workgroups per CU (<= 128 VGPRs): with 16 in flight the kernel needed 169 and ran one workgroup per CU, its select
    // and sort phases - barriers and LDS latency - covered by nothing
    const int d4 = d >> 2;
    constexpr int RU = 8;
    for (int i0 = w; i0 < m; i0 += NW * RU) {
        float a[RU];
        uint32_t pos[RU];
        const f32x4* xr[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int i = i0 + NW * u;
            a[u] = 0.f;
            pos[u] = key_pos(keys[i < m ? i : i0]);
            xr[u] = reinterpret_cast<const f32x4*>(X + (long long)pos[u] * ldx);
        }
        // column chunk outermost: the RU row loads of one chunk are independent and go out back to back (with the
        // row loop outside, each row's load had to return before the next row's was issued: measured 100 ns per row)
        for (int cc = lane; cc < d4; cc += 64) {
            const f32x4 y = *reinterpret_cast<const f32x4*>(&qv[4 * cc]);
            f32x4 x[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) x[u] = xr[u][cc];
#pragma unroll
            for (int u = 0; u < RU; ++u)       // explicit fma chain: the same rounding sequence in every slot u
                a[u] = __builtin_fmaf(x[u][3], y[3], __builtin_fmaf(x[u][2], y[2], __builtin_fmaf(x[u][1], y[1],
                                      __builtin_fmaf(x[u][0], y[0], a[u]))));
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            float v = a[u];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            const int i = i0 + NW * u;
            // a NaN re-score (inf - inf in fp32 that the bf16 pass did not produce) ranks last, as in the fix-up scan
            if (lane == 0 && i < m) keys[i] = (v == v) ? make_key(v, pos[u]) : 0ull;
        }
    }
    __syncthreads();
    const unsigned long long* sorted = keys;
    if (NT == 512 && m <= 2048) {                             // (8192 keys of LDS: source and destination both fit)
        const int N = m <= 1024 ? 1024 : 2048;
        for (int i = tid; i < N; i += NT) keys[N + i] = 0ull;    // NaN re-scores (key 0) are not placed by the run sort
        __syncthreads();
        if constexpr (NT == 512) {
            if (m <= 1024) sort_desc_runs<2>(keys, keys + N, m);
            else sort_desc_runs<4>(keys, keys + N, m);
        }
        sorted = keys + N;
    } else {
        int P2 = 2;
        while (P2 < m) P2 <<= 1;
        for (int i = m + tid; i < P2; i += NT) keys[i] = 0ull;
        __syncthreads();
        bitonic_desc(keys, P2);
    }
    // certificate (block-uniform): rows outside the list have exact < tau + eps
    const unsigned long long kth = sorted[need - 1];
    const bool all_rows = (long long)c >= nrows;
    if (!all_rows && !(kth != 0ull && key_score(kth) >= tau[q] + eps)) { give_up(); return; }
    write_result(sorted, m, k, q, outD, outI, pos_offset);
}

// Small batches (<= FUSED_MAX_NQ queries), round 3: select -> prune -> re-score -> sort -> certificate in ONE launch of
// gridDim.x workgroups per query (the three kernels above cost 38 us at one query: 12.5 + 8 + 17, each of the small ones
// mostly launch-to-drain latency; they are gone).  Every workgroup of a query repeats the select + prune on the query's candidates (the same
// deterministic result in each: redundant, but in parallel), re-scores the survivors whose corpus position falls to it
// (pos % gridDim.x: a share that does not depend on the order in which a block's threads compacted the survivors), appends
// the exact keys to the query's list in the workspace, and takes a ticket; the workgroup that draws the LAST ticket (agent-scope
// release / acquire around it, as in fixup_kernel) sorts the list - sort_desc_runs for <= 2048 survivors -, certifies and
// writes the result.  Same per-row re-score arithmetic, same prune rule, same certificate as finalize_mixed_kernel.
__global__ __launch_bounds__(512) void finalize_fused_kernel(const unsigned long long* cand, long long cstride, const int* segcnt,
                                                             int nseg, int seg_cap, const int* ocnt, int cap, int k,
                                                             long long nrows, const float* tau, const float* max_norm,
                                                             const float* X, long long ldx, int d, const float* Q, long long ldq,
                                                             int* fail, float* outD, long long* outI, long long pos_offset,
                                                             unsigned long long* exact, int* fcount, int* fticket) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];      // [cap] then qv[d]
    float* qv = reinterpret_cast<float*>(keys + cap);
    __shared__ int hist[2048];
    __shared__ int scratch[514];
    __shared__ float red[16];
    __shared__ int m_sh, own_sh, base_sh, last_sh;
    __shared__ int seg_n[256], tot_sh, lost_sh;
    constexpr int PER = CAND_CAP / 512;
    constexpr int OVP = 4;
    const int q = blockIdx.y, nq = gridDim.y, s = blockIdx.x, S = gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int need = (int)(nrows < k ? nrows : k);
    auto give_up = [&]() {                                      // every workgroup of the query reaches the same verdict: one reports
        if (tid == 0 && s == 0) {
            fail[q] = 1;
            atomicAdd(&fail[nq], 1);
        }
    };
    // the candidate slots are requested before the segment counts that say which of them are valid have arrived (one
    // global round trip instead of two); validity is applied below
    const unsigned long long* blk = cand + (long long)q * cstride;
    const int nslots = nseg * seg_cap;
    unsigned long long mine[PER + OVP];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int s_ = tid + 512 * j;
        mine[j] = s_ < nslots ? blk[s_] : 0ull;
    }
#pragma unroll
    for (int j = 0; j < OVP; ++j) mine[PER + j] = blk[CAND_CAP + tid + 512 * j];
    if (tid == 0) { tot_sh = 0; lost_sh = 0; }
    __syncthreads();
    int sv = 0, lost = 0;
    if (tid < nseg) {
        const int v = segcnt[(long long)q * nseg + tid];
        sv = v < seg_cap ? v : seg_cap;
        lost = v - sv;
        seg_n[tid] = sv;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sv += __shfl_xor(sv, o, 64);
        lost += __shfl_xor(lost, o, 64);
    }
    if (lane == 0 && w * 64 < nseg) {
        atomicAdd(&tot_sh, sv);
        if (lost) atomicAdd(&lost_sh, lost);
    }
    const int oc = ocnt[q];
    __syncthreads();
    const int c = tot_sh + oc;
    if (lost_sh != oc || oc > 512 * OVP || c < need || c > cap) { give_up(); return; }
    float ss = 0.f, ds = 0.f;
    for (int i = tid; i < d; i += 512) {
        const float v = Q[(long long)q * ldq + i];
        qv[i] = v;
        ss += v * v;
        const uint32_t u = __float_as_uint(v);                    // the same rounding as bf16_rows_kernel applied to the query
        const float dv = v - __uint_as_float(((u + 0x7fffu + ((u >> 16) & 1u)) >> 16) << 16);
        ds += dv * dv;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ss += __shfl_xor(ss, o, 64);
        ds += __shfl_xor(ds, o, 64);
    }
    if (lane == 0) { red[w] = ss; red[8 + w] = ds; }
    if (tid == 0) { m_sh = 0; own_sh = 0; }
    const int seg_shift = (nseg & (nseg - 1)) == 0 ? 31 - __builtin_clz((unsigned)nseg) : -1;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int s_ = tid + 512 * j;
        bool ok = false;
        if (s_ < nslots) {
            const int slot = seg_shift >= 0 ? s_ >> seg_shift : s_ / nseg;
            ok = slot < seg_n[s_ - slot * nseg];
        }
        if (!ok) mine[j] = 0ull;
    }
#pragma unroll
    for (int j = 0; j < OVP; ++j)
        if (tid + 512 * j >= oc) mine[PER + j] = 0ull;
    __syncthreads();
    float qn = 0.f, dqn = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { qn += red[i]; dqn += red[8 + i]; }
    const float eps = eps_bound(sqrtf(qn), sqrtf(dqn) * 1.0001f, max_norm[0], max_norm[1], d);
    if (!(eps < INFINITY)) { give_up(); return; }
    if (need == 0) {
        if (s == 0) write_result(keys, 0, k, q, outD, outI, pos_offset);
        return;
    }
    // a_k to its top 22 bits (two radix passes): the remaining 10 bits cleared give a value <= a_k in the orderable image,
    // i.e. a slightly lower cut - a superset of the survivors of the exact a_k (a relative 2^-13 of the score: no
    // measurable number of extra survivors), never a missing one
    uint32_t prefix = 0u, pmask = 0u;
    int rr = need;
    const int shifts[2] = {21, 10};
    const int nbits[2] = {11, 11};
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = tid; i < 2048; i += 512) hist[i] = 0;
        __syncthreads();
        const uint32_t bm = (1u << nbits[pass]) - 1u;
#pragma unroll
        for (int j = 0; j < PER + OVP; ++j) {
            const uint32_t u = (uint32_t)(mine[j] >> 32);
            if (mine[j] != 0ull && (u & pmask) == prefix) atomicAdd(&hist[(u >> shifts[pass]) & bm], 1);
        }
        __syncthreads();
        const int bin = find_bin_desc<2048, 512>(hist, rr, scratch);
        prefix |= (uint32_t)bin << shifts[pass];
        pmask |= bm << shifts[pass];
    }
    // prune (as in finalize_mixed_kernel); this workgroup keeps the survivors at positions pos % S == s
    const float cut = f32_from_orderable(prefix) - 2.f * eps;
    int n_surv = 0;
#pragma unroll
    for (int j = 0; j < PER + OVP; ++j)
        if (mine[j] != 0ull && key_score(mine[j]) >= cut) {
            ++n_surv;
            if ((int)(key_pos(mine[j]) % (uint32_t)S) == s) keys[atomicAdd(&own_sh, 1)] = mine[j];
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n_surv += __shfl_xor(n_surv, o, 64);
    if (lane == 0 && n_surv) atomicAdd(&m_sh, n_surv);
    __syncthreads();
    const int m = m_sh, own = own_sh;                         // need <= m <= c
    // fp32 re-score of the own share, one wave per candidate, 8 in flight per wave (finalize_mixed_kernel's arithmetic)
    const int d4 = d >> 2;
    constexpr int RU = 8;
    for (int i0 = w; i0 < own; i0 += 8 * RU) {
        float a[RU];
        uint32_t pos[RU];
        const f32x4* xr[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int i = i0 + 8 * u;
            a[u] = 0.f;
            pos[u] = key_pos(keys[i < own ? i : i0]);
            xr[u] = reinterpret_cast<const f32x4*>(X + (long long)pos[u] * ldx);
        }
        for (int cc = lane; cc < d4; cc += 64) {
            const f32x4 y = *reinterpret_cast<const f32x4*>(&qv[4 * cc]);
            f32x4 x[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) x[u] = xr[u][cc];
#pragma unroll
            for (int u = 0; u < RU; ++u)
                a[u] = __builtin_fmaf(x[u][3], y[3], __builtin_fmaf(x[u][2], y[2], __builtin_fmaf(x[u][1], y[1],
                                      __builtin_fmaf(x[u][0], y[0], a[u]))));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");    // keys[i] of this round were read (pos) before they are rewritten
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            float v = a[u];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            const int i = i0 + 8 * u;
            if (lane == 0 && i < own) keys[i] = (v == v) ? make_key(v, pos[u]) : 0ull;   // NaN re-score: ranks last (key 0)
        }
    }
    __syncthreads();
    if (tid == 0) base_sh = __hip_atomic_fetch_add(&fcount[q], own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    unsigned long long* list = exact + (long long)q * cap;
    for (int i = tid; i < own; i += 512) list[base_sh + i] = keys[i];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's stores are out
    __syncthreads();                                           // ... every wave's
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int t = __hip_atomic_fetch_add(&fticket[q], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_sh = t == S - 1;
        if (t == S - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (!last_sh) return;                                      // block-uniform
    // the query's last workgroup: all m exact keys are in the list (NaN re-scores as key 0: they sort last, as before)
    if (m <= 2048) {
        const int N = m <= 1024 ? 1024 : 2048;
        // zero keys (NaN re-scores) are not placed by the run sort: the destination starts cleared
        for (int i = tid; i < N; i += 512) { keys[i] = i < m ? list[i] : 0ull; keys[N + i] = 0ull; }
        __syncthreads();
        if (m <= 1024) sort_desc_runs<2>(keys, keys + N, m);
        else sort_desc_runs<4>(keys, keys + N, m);
        const unsigned long long* sorted = keys + N;
        const unsigned long long kth = sorted[need - 1];
        const bool all_rows = (long long)c >= nrows;
        if (!all_rows && !(kth != 0ull && key_score(kth) >= tau[q] + eps)) {
            if (tid == 0) { fail[q] = 1; atomicAdd(&fail[nq], 1); }
            return;
        }
        write_result(sorted, m, k, q, outD, outI, pos_offset);
        return;
    }
    int P2 = 2;
    while (P2 < m) P2 <<= 1;
    for (int i = tid; i < P2; i += 512) keys[i] = i < m ? list[i] : 0ull;
    __syncthreads();
    bitonic_desc(keys, P2);
    const unsigned long long kth = keys[need - 1];
    const bool all_rows = (long long)c >= nrows;
    if (!all_rows && !(kth != 0ull && key_score(kth) >= tau[q] + eps)) {
        if (tid == 0) { fail[q] = 1; atomicAdd(&fail[nq], 1); }
        return;
    }
    write_result(keys, m, k, q, outD, outI, pos_offset);
}

__global__ void fill_f32_kernel(float* p, long long n, float v) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// ---- plan ---------------------------------------------------------------------------
struct SearchPlan {
    long long n_sample;      // sample rows (multiple of SAMPLE_G), 0 => no sampling (tau = -inf)
    long long gstride;       // rows between sample blocks
    int rank;                // r
    int nslices;             // fix-up slices
    long long target;        // expected candidates per query (0: every row is a candidate)
    size_t off_tau, off_cnt, off_ocnt, off_ticket, off_fcount, off_fticket, off_fail, off_segcnt, off_cand, off_sample, off_fix, off_q16, off_exact, bytes;
};

// dim16 > 0: the mixed-precision search, which also keeps a bf16 copy of the queries in the workspace
static int make_plan(long long nq, long long nrows, int k, SearchPlan& pl, int dim16 = 0) {
    long long nt = (nrows + SAMPLE_G - 1) / SAMPLE_G;
    // expected candidates per query: far enough above k that an unlucky sample cannot undershoot it (the estimate's
    // sigma is ~target/8: k = 500 -> 1400 = k + 5.1 sigma), small enough that the finalize sort stays at 2048 keys for
    // k = 500.  Round 2 gave every k at least k + 900 - for the SHORT per-shard lists of the sharded search (128 of 500 at 8
    // ranks) that is 7 sigma and, on a shard an eighth the size, 8x the hit density of the single-GPU pass.  Now, for
    // batches whose corpus pass is compute-side bound (>= 256 queries; a small batch is HBM-bound and pays more for a larger
    // sample than its candidates cost): between the 5.1-sigma floor max(2.8 k, k + 256) and k + 900, the value that balances
    // the sample (rows ~ 64 nrows / target) against the per-hit work of the pass (~ target): ~0.6 sqrt(nrows), fitted to
    // same-box runs at 125k-row shards and 1M rows (profiles/r03_shard_search_target.log: per-rank search at the 8-rank
    // shape 0.766 -> 0.660 ms, k = 500 at 1M unchanged).
    long long t5 = k + 900ll;
    if (nq >= 256) {
        long long lo = (28ll * k + 9) / 10;
        if (lo < k + 256ll) lo = k + 256ll;
        long long opt = (long long)(0.6 * sqrt((double)(nrows > 0 ? nrows : 1)));
        if (opt < lo) opt = lo;
        if (opt < t5) t5 = opt;
    }
    long long target = (2ll * k > t5) ? 2ll * k : t5;
    pl.target = nrows <= CAND_CAP ? 0 : target;
    if (nrows <= CAND_CAP) {
        pl.n_sample = 0;                  // every row becomes a candidate
        pl.gstride = SAMPLE_G;
        pl.rank = 0;
    } else {
        long long n = (SAMPLE_RANK * nrows + target - 1) / target;
        long long st = (n + SAMPLE_G - 1) / SAMPLE_G;       // sample tiles
        if (st < 1) st = 1;
        if (st > nt) st = nt;
        long long stride = nt / st;
        pl.gstride = stride * SAMPLE_G;
        pl.n_sample = st * SAMPLE_G;
        pl.rank = SAMPLE_RANK;
    }
    int ns = CAND_CAP / (k > 0 ? k : 1);
    pl.nslices = ns < 1 ? 1 : (ns > 16 ? 16 : ns);
    size_t o = 0;
    pl.off_tau = o;    o = align_up(o + (size_t)nq * 4, 256);
    pl.off_cnt = o;    o = align_up(o + (size_t)nq * 4, 256);                   // cnt | ocnt | ticket | fail: cleared together
    pl.off_ocnt = o;   o = align_up(o + (size_t)nq * 4, 256);
    pl.off_ticket = o; o = align_up(o + (size_t)nq * 4, 256);                   // fix-up: slices finished per query
    pl.off_fcount = o; o = align_up(o + (size_t)nq * 4, 256);                   // fused finalize (small batches): exact keys appended,
    pl.off_fticket = o; o = align_up(o + (size_t)nq * 4, 256);                  // blocks finished per query
    pl.off_fail = o;   o = align_up(o + (size_t)(nq + 1) * 4, 256);
    pl.off_segcnt = o; o = align_up(o + (size_t)(dim16 ? nq : 0) * 256 * 4, 256);  // streaming pass: hits per (query, segment)
    pl.off_cand = o;   o = align_up(o + (size_t)nq * CAND_CAP * 8 * (dim16 ? 2 : 1), 256);   // mixed: + overflow block
    pl.off_sample = o; o = align_up(o + (size_t)nq * (size_t)pl.n_sample * 4, 256);
    pl.off_fix = o;    o = align_up(o + (size_t)nq * pl.nslices * k * 8, 256);
    pl.off_q16 = o;    o = align_up(o + (size_t)nq * (size_t)dim16 * 2, 256);
    pl.off_exact = o;  o = align_up(o + (size_t)(dim16 && nq <= FUSED_MAX_NQ ? nq : 0) * CAND_CAP * 8, 256);   // fused finalize: re-scored keys
    pl.bytes = o;
    return 0;
}

// X / Q / ldx / ldq / d are in staged floats: for a BF16 shape the bf16 matrices viewed as float matrices of
// half the columns (d_alg = the un-halved dimension, for the profiling hook's FLOP count)
template <class S>
static hipError_t run_passes(const float* X, long long ldx, long long nrows, int d, const float* Q, long long ldq,
                             int nq, const SearchPlan& pl, char* ws, hipStream_t st, int d_alg = 0,
                             bool filter = true, long long cand_stride = CAND_CAP) {
    DenseRows lq{Q, nq, (int)ldq, d, 30, 1ll << 30};
    float* tau = reinterpret_cast<float*>(ws + pl.off_tau);
    int* cnt = reinterpret_cast<int*>(ws + pl.off_cnt);
    if (pl.n_sample > 0) {
        int gshift = 8;   // SAMPLE_G == 256
        DenseRows lps{X, nrows, (int)ldx, d, gshift, pl.gstride};
        float* S_ = reinterpret_cast<float*>(ws + pl.off_sample);
        EpiStoreScores es{S_, pl.n_sample, nq, pl.n_sample, lps};
        hipError_t e = launch_gemm<S, false>(lps, lq, es, d, pl.n_sample, nq, st, d_alg);
        if (e != hipSuccess) return e;
        ProfScope prof("search_threshold", 0.0, 4.0 * (double)nq * (double)pl.n_sample, st);
        hipLaunchKernelGGL(sample_threshold_kernel, dim3(nq), dim3(THR_NT), 0, st, S_, pl.n_sample, pl.n_sample,
                           pl.rank, tau);
    } else {
        hipLaunchKernelGGL(fill_f32_kernel, dim3((nq + 255) / 256), dim3(256), 0, st, tau, (long long)nq,
                           -INFINITY);
    }
    if (!filter) return hipGetLastError();
    DenseRows lp{X, nrows, (int)ldx, d, 30, 1ll << 30};
    EpiFilter ef{tau, reinterpret_cast<unsigned long long*>(ws + pl.off_cand), cnt, CAND_CAP, nq, nrows, cand_stride};
    return launch_gemm<S, false>(lp, lq, ef, d, nrows, nq, st, d_alg);
}

}  // namespace amdrec

using namespace amdrec;

static int topk_merge_impl(const float* scores, const int32_t* pos, int n_lists, int list_k, int64_t list_stride_bytes,
                           int64_t q0, int64_t nq, int k, float* out_scores, int64_t* out_pos, int32_t* n_inexact,
                           void* stream) {
    REQUIRE(n_lists >= 1 && list_k >= 1 && k >= 1 && (long long)n_lists * list_k <= 16384 && k <= 16384,
            "n_lists*list_k and k must be in [1,16384]");
    REQUIRE(list_stride_bytes % 4 == 0 && q0 >= 0, "bad stride/offset");
    if (nq <= 0) return AMDREC_OK;
    REQUIRE(scores && pos && out_scores && out_pos, "null pointer");
    int P = 2;
    while (P < n_lists * list_k) P <<= 1;
    static PerDeviceOnce attr_done;
    if (attr_done.pending()) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(topk_merge_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8));
        attr_done.mark();
    }
    hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)nq), dim3(512), (size_t)P * 8,
                       reinterpret_cast<hipStream_t>(stream), (const char*)scores, (const char*)pos, n_lists, list_k,
                       (long long)list_stride_bytes, (long long)q0, k, out_scores, (long long*)out_pos, (int*)n_inexact);
    HIP_TRY(hipGetLastError());
    return AMDREC_OK;
}

extern "C" int amdrec_topk_merge(const float* scores, const int32_t* pos, int n_lists, int64_t list_stride_bytes,
                                 int64_t q0, int64_t nq, int k, float* out_scores, int64_t* out_pos,
                                 void* stream) {
    return topk_merge_impl(scores, pos, n_lists, k, list_stride_bytes, q0, nq, k, out_scores, out_pos, nullptr, stream);
}

extern "C" int amdrec_topk_merge_partial(const float* scores, const int32_t* pos, int n_lists, int list_k,
                                         int64_t list_stride_bytes, int64_t q0, int64_t nq, int k, float* out_scores,
                                         int64_t* out_pos, int32_t* n_inexact, void* stream) {
    REQUIRE(n_inexact != nullptr, "n_inexact is null");
    return topk_merge_impl(scores, pos, n_lists, list_k, list_stride_bytes, q0, nq, k, out_scores, out_pos, n_inexact,
                           stream);
}

extern "C" int amdrec_flat_search_workspace(int64_t nq, int64_t nrows, int k, size_t* bytes) {
    REQUIRE(bytes != nullptr, "bytes is null");
    REQUIRE(nq >= 0 && nrows >= 0, "negative size");
    REQUIRE(k >= 1 && k <= KMAX, "k=%d outside [1,%d]", k, KMAX);
    SearchPlan pl;
    make_plan(nq, nrows, k, pl);
    *bytes = pl.bytes;
    return AMDREC_OK;
}

extern "C" int amdrec_flat_search(const float* corpus, int64_t nrows, int64_t ld_corpus, int dim,
                                  const float* queries, int64_t nq, int64_t ld_queries, int k,
                                  int64_t pos_offset, float* out_scores, int64_t* out_pos, void* workspace,
                                  size_t workspace_bytes, int* n_fixup, void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    REQUIRE(k >= 1 && k <= KMAX, "k=%d outside [1,%d]", k, KMAX);
    REQUIRE(dim >= 4 && dim % 4 == 0 && dim <= 2048, "dim=%d must be a multiple of 4 in [4,2048]", dim);
    REQUIRE(nrows >= 0 && nrows < (1ll << 31) - 1024, "nrows out of range");
    REQUIRE(nq >= 0 && nq < (1ll << 24), "nq out of range");
    if (nq == 0) return AMDREC_OK;
    REQUIRE((nrows == 0 || (ld_corpus >= dim && ld_corpus % 4 == 0)) && ld_queries >= dim && ld_queries % 4 == 0,
            "leading dimensions must be >= dim and multiples of 4");
    REQUIRE(corpus || nrows == 0, "corpus is null");
    REQUIRE(queries && out_scores && out_pos, "null pointer");
    REQUIRE(((uintptr_t)corpus % 16) == 0 && ((uintptr_t)queries % 16) == 0, "corpus/queries must be 16-byte aligned");
    SearchPlan pl;
    make_plan(nq, nrows, k, pl);
    if (workspace_bytes < pl.bytes || workspace == nullptr)
        return set_error(AMDREC_EWORKSPACE, "workspace too small: need %zu bytes, got %zu", pl.bytes, workspace_bytes);
    REQUIRE(((uintptr_t)workspace % 256) == 0, "workspace must be 256-byte aligned");
    char* ws = reinterpret_cast<char*>(workspace);
    int* cnt = reinterpret_cast<int*>(ws + pl.off_cnt);
    int* fail = reinterpret_cast<int*>(ws + pl.off_fail);
    // cnt[nq] and fail[nq + 1] are adjacent in the plan: one fill
    HIP_TRY(hipMemsetAsync(cnt, 0, (pl.off_fail - pl.off_cnt) + (size_t)(nq + 1) * 4, st));

    if (nrows > 0) {
        hipError_t e;
        // corpus tile 256 rows x query tile 128 / 64 / 32 (4 waves each)
        if (nq > 64)       e = run_passes<Shape<2, 2, 4, 2>>(corpus, ld_corpus, nrows, dim, queries, ld_queries, (int)nq, pl, ws, st);
        else if (nq > 32)  e = run_passes<Shape<4, 1, 2, 2>>(corpus, ld_corpus, nrows, dim, queries, ld_queries, (int)nq, pl, ws, st);
        else               e = run_passes<Shape<4, 1, 2, 1>>(corpus, ld_corpus, nrows, dim, queries, ld_queries, (int)nq, pl, ws, st);
        HIP_TRY(e);
    }
    static PerDeviceOnce attr_done;
    if (attr_done.pending()) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(finalize_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, CAND_CAP * 8));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fixup_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, CAND_CAP * 8));
        attr_done.mark();
    }
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(ws + pl.off_cand);
    unsigned long long* fix = reinterpret_cast<unsigned long long*>(ws + pl.off_fix);
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)nq), dim3(512), CAND_CAP * 8, st, cand, cnt, CAND_CAP, k,
                       (long long)nrows, fail, out_scores, (long long*)out_pos, (long long)pos_offset);
    hipLaunchKernelGGL(fixup_kernel, dim3((unsigned)(pl.nslices * (nq < FIX_GRID_Q ? nq : FIX_GRID_Q))), dim3(512), CAND_CAP * 8, st,
                       corpus, (long long)ld_corpus, (long long)nrows, dim, queries, (long long)ld_queries, fail, (int)nq, k,
                       pl.nslices, fix, reinterpret_cast<int*>(ws + pl.off_ticket), out_scores, (long long*)out_pos,
                       (long long)pos_offset);
    HIP_TRY(hipGetLastError());
    if (n_fixup) HIP_TRY(hipMemcpyAsync(n_fixup, fail + nq, sizeof(int), hipMemcpyDeviceToDevice, st));
    return AMDREC_OK;
}

extern "C" int amdrec_bf16_rows(const float* x, int64_t rows, int64_t ld, int dim, uint16_t* out, int64_t ld_out,
                                float* max_norm, void* stream) {
    REQUIRE(dim >= 4 && dim % 4 == 0 && dim <= 2048, "dim=%d must be a multiple of 4 in [4,2048]", dim);
    if (rows <= 0) return AMDREC_OK;
    REQUIRE(x && out, "null pointer");
    REQUIRE(ld >= dim && ld % 4 == 0 && ld_out >= dim && ld_out % 4 == 0, "leading dimensions must be >= dim and multiples of 4");
    REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)out % 8) == 0, "x must be 16-byte and out 8-byte aligned");
    hipLaunchKernelGGL(bf16_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), x, (long long)rows, (long long)ld, dim, out,
                       (long long)ld_out, max_norm, (int*)nullptr, 0ll);
    HIP_TRY(hipGetLastError());
    return AMDREC_OK;
}

extern "C" int amdrec_flat_search_mixed_workspace(int64_t nq, int64_t nrows, int k, int dim, size_t* bytes) {
    REQUIRE(bytes != nullptr, "bytes is null");
    REQUIRE(nq >= 0 && nrows >= 0, "negative size");
    REQUIRE(k >= 1 && k <= KMAX, "k=%d outside [1,%d]", k, KMAX);
    REQUIRE(dim >= 8 && dim % 8 == 0 && dim <= 2048, "dim=%d must be a multiple of 8 in [8,2048]", dim);
    SearchPlan pl;
    make_plan(nq, nrows, k, pl, dim);
    *bytes = pl.bytes;
    return AMDREC_OK;
}

extern "C" int amdrec_flat_search_mixed(const float* corpus, int64_t nrows, int64_t ld_corpus, int dim,
                                        const uint16_t* corpus_bf16, int64_t ld_bf16, const float* max_norm,
                                        const float* queries, int64_t nq, int64_t ld_queries, int k,
                                        int64_t pos_offset, float* out_scores, int64_t* out_pos, void* workspace,
                                        size_t workspace_bytes, int* n_fixup, void* stream) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    REQUIRE(k >= 1 && k <= KMAX, "k=%d outside [1,%d]", k, KMAX);
    REQUIRE(dim >= 8 && dim % 8 == 0 && dim <= 2048, "dim=%d must be a multiple of 8 in [8,2048]", dim);
    REQUIRE(nrows >= 0 && nrows < (1ll << 31) - 1024, "nrows out of range");
    REQUIRE(nq >= 0 && nq < (1ll << 24), "nq out of range");
    if (nq == 0) return AMDREC_OK;
    REQUIRE((nrows == 0 || (ld_corpus >= dim && ld_corpus % 4 == 0 && ld_bf16 >= dim && ld_bf16 % 8 == 0 &&
                            ld_bf16 < (1 << 20))) &&
                ld_queries >= dim && ld_queries % 4 == 0,
            "leading dimensions must be >= dim; fp32 multiples of 4, bf16 multiples of 8");
    REQUIRE((corpus && corpus_bf16) || nrows == 0, "corpus is null");
    REQUIRE(queries && out_scores && out_pos && max_norm, "null pointer");
    REQUIRE(((uintptr_t)corpus % 16) == 0 && ((uintptr_t)corpus_bf16 % 16) == 0 && ((uintptr_t)queries % 16) == 0,
            "corpus/queries must be 16-byte aligned");
    SearchPlan pl;
    make_plan(nq, nrows, k, pl, dim);
    if (workspace_bytes < pl.bytes || workspace == nullptr)
        return set_error(AMDREC_EWORKSPACE, "workspace too small: need %zu bytes, got %zu", pl.bytes, workspace_bytes);
    REQUIRE(((uintptr_t)workspace % 256) == 0, "workspace must be 256-byte aligned");
    char* ws = reinterpret_cast<char*>(workspace);
    int* cnt = reinterpret_cast<int*>(ws + pl.off_cnt);
    int* ocnt = reinterpret_cast<int*>(ws + pl.off_ocnt);
    int* fail = reinterpret_cast<int*>(ws + pl.off_fail);
    // cnt[nq], ocnt[nq] and fail[nq + 1] are adjacent in the plan: cleared by the query-conversion kernel below (no
    // separate fill launch: a launch costs ~5 us, a 32-query search 170)
    const long long n_zero = (long long)((pl.off_fail - pl.off_cnt) / 4) + nq + 1;
    if (nrows <= 0) HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)n_zero * 4, st));
    constexpr long long CSTRIDE = CAND_STRIDE;
    // corpus pass: the streaming kernel for the power-of-two dims it is instantiated for, else the generic tiles
    const bool streaming = nrows > 0 && (dim == 32 || dim == 64 || dim == 128 || dim == 256);
    const int nseg = streaming ? scan_segments(nrows, nq) : 1;
    const int seg_cap = CAND_CAP / nseg;
    const int* segcnt = streaming ? reinterpret_cast<int*>(ws + pl.off_segcnt) : cnt;

    if (streaming && pl.n_sample > 0) {
        // conversion + sample in ONE launch (group maxima only), tau from the maxima: inside the corpus pass for <= 8
        // queries (no threshold launch), one wave per query otherwise
        uint16_t* q16 = reinterpret_cast<uint16_t*>(ws + pl.off_q16);
        float* gm = reinterpret_cast<float*>(ws + pl.off_sample);
        float* tau_ = reinterpret_cast<float*>(ws + pl.off_tau);
        unsigned long long* cand_ = reinterpret_cast<unsigned long long*>(ws + pl.off_cand);
        int* sc_ = reinterpret_cast<int*>(ws + pl.off_segcnt);
        const SamplePlan sp = sample_plan(nrows, pl.n_sample);
        REQUIRE(sp.cols <= pl.n_sample, "internal: sample maxima do not fit their workspace");
        const bool tau_in_scan = nq <= 8;
        hipError_t e;
        auto run = [&](auto ks_tag) -> hipError_t {
            constexpr int KS = decltype(ks_tag)::value;
            hipError_t e2 = launch_sample<KS>(corpus_bf16, ld_bf16, sp, queries, ld_queries, (int)nq, q16, gm, cnt, n_zero, st);
            if (e2 != hipSuccess) return e2;
            if (!tau_in_scan) {
                ProfScope prof("search_threshold", 0.0, 4.0 * (double)nq * (double)sp.cols, st);
                hipLaunchKernelGGL(tau_from_maxima_kernel, dim3((unsigned)((nq + 7) / 8)), dim3(512), 0, st, (const float*)gm,
                                   sp.cols, sp.cols, pl.rank, (int)nq, tau_);
            }
            return launch_scan<KS>(corpus_bf16, ld_bf16, nrows, q16, (int)nq, tau_, cand_, sc_, ocnt, st,
                                   tau_in_scan ? (const float*)gm : nullptr, sp.cols, sp.cols, pl.rank);
        };
        if (dim == 256)      e = run(std::integral_constant<int, 16>{});
        else if (dim == 128) e = run(std::integral_constant<int, 8>{});
        else if (dim == 64)  e = run(std::integral_constant<int, 4>{});
        else                 e = run(std::integral_constant<int, 2>{});
        HIP_TRY(e);
    } else if (nrows > 0) {
        uint16_t* q16 = reinterpret_cast<uint16_t*>(ws + pl.off_q16);
        hipLaunchKernelGGL(bf16_rows_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, st, queries, (long long)nq,
                           (long long)ld_queries, dim, q16, (long long)dim, (float*)nullptr, cnt, n_zero);
        // the bf16 matrices as float matrices of dim/2 columns (gemm_core.hpp, Shape::BF16)
        const float* X = reinterpret_cast<const float*>(corpus_bf16);
        const float* Q = reinterpret_cast<const float*>(q16);
        const long long ldx = ld_bf16 / 2, ldq = dim / 2;
        const int dh = dim / 2;
        hipError_t e;
        if (nq > 64)       e = run_passes<Shape<2, 2, 4, 2, false, true>>(X, ldx, nrows, dh, Q, ldq, (int)nq, pl, ws, st, dim, !streaming, CSTRIDE);
        else if (nq > 32)  e = run_passes<Shape<4, 1, 2, 2, false, true>>(X, ldx, nrows, dh, Q, ldq, (int)nq, pl, ws, st, dim, !streaming, CSTRIDE);
        else               e = run_passes<Shape<4, 1, 2, 1, false, true>>(X, ldx, nrows, dh, Q, ldq, (int)nq, pl, ws, st, dim, !streaming, CSTRIDE);
        HIP_TRY(e);
        if (streaming) {
            float* tau_ = reinterpret_cast<float*>(ws + pl.off_tau);
            unsigned long long* cand_ = reinterpret_cast<unsigned long long*>(ws + pl.off_cand);
            int* sc_ = reinterpret_cast<int*>(ws + pl.off_segcnt);
            if (dim == 256)      e = launch_scan<16>(corpus_bf16, ld_bf16, nrows, q16, (int)nq, tau_, cand_, sc_, ocnt, st);
            else if (dim == 128) e = launch_scan<8>(corpus_bf16, ld_bf16, nrows, q16, (int)nq, tau_, cand_, sc_, ocnt, st);
            else if (dim == 64)  e = launch_scan<4>(corpus_bf16, ld_bf16, nrows, q16, (int)nq, tau_, cand_, sc_, ocnt, st);
            else                 e = launch_scan<2>(corpus_bf16, ld_bf16, nrows, q16, (int)nq, tau_, cand_, sc_, ocnt, st);
            HIP_TRY(e);
        }
    }
    const size_t fin_lds = (size_t)CAND_CAP * 8 + (size_t)dim * 4;
    static PerDeviceOnce attr_done;
    if (attr_done.pending()) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(finalize_mixed_kernel<512, CAND_CAP>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, CAND_CAP * 8 + 2048 * 4));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(finalize_fused_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, CAND_CAP * 8 + 2048 * 4));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fixup_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, CAND_CAP * 8));
        attr_done.mark();
    }
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(ws + pl.off_cand);
    unsigned long long* fix = reinterpret_cast<unsigned long long*>(ws + pl.off_fix);
    const float* tau = reinterpret_cast<const float*>(ws + pl.off_tau);
    {
        ProfScope prof("search_finalize_mixed", 0.0, 0.0, st);
        if (nq <= FUSED_MAX_NQ) {
            int slices = (int)(512 / nq);                           // ~512 workgroups in all: the chip once
            slices = slices < 1 ? 1 : (slices > 16 ? 16 : slices);
            hipLaunchKernelGGL(finalize_fused_kernel, dim3((unsigned)slices, (unsigned)nq), dim3(512), fin_lds, st,
                               (const unsigned long long*)cand, CSTRIDE, segcnt, nseg, seg_cap, (const int*)ocnt, CAND_CAP, k,
                               (long long)nrows, tau, max_norm, corpus, (long long)ld_corpus, dim, queries,
                               (long long)ld_queries, fail, out_scores, (long long*)out_pos, (long long)pos_offset,
                               reinterpret_cast<unsigned long long*>(ws + pl.off_exact),
                               reinterpret_cast<int*>(ws + pl.off_fcount), reinterpret_cast<int*>(ws + pl.off_fticket));
        } else {
            // one workgroup per query; short lists of many queries (the sharded search) on the small shapes: twice the
            // expected candidate count must fit (10 sigma of the threshold estimate; beyond it the query takes the fix-up scan)
#define AMDREC_FINALIZE(NT_, CAPK_)                                                                                          \
    hipLaunchKernelGGL((finalize_mixed_kernel<NT_, CAPK_>), dim3((unsigned)nq), dim3(NT_), (size_t)(CAPK_) * 8 + (size_t)dim * 4, \
                       st, (const unsigned long long*)cand, CSTRIDE, segcnt, nseg, seg_cap, (const int*)ocnt, k,             \
                       (long long)nrows, tau, max_norm, corpus, (long long)ld_corpus, dim, queries, (long long)ld_queries,    \
                       fail, out_scores, (long long*)out_pos, (long long)pos_offset)
            if (pl.target > 0 && 2 * pl.target <= 1024 && dim <= 2048) AMDREC_FINALIZE(128, 1024);
            else if (pl.target > 0 && 2 * pl.target <= 2048 && dim <= 2048) AMDREC_FINALIZE(256, 2048);
            else AMDREC_FINALIZE(512, CAND_CAP);
#undef AMDREC_FINALIZE
        }
    }
    ProfScope prof_fix("search_fixup", 0.0, 0.0, st);
    hipLaunchKernelGGL(fixup_kernel, dim3((unsigned)(pl.nslices * (nq < FIX_GRID_Q ? nq : FIX_GRID_Q))), dim3(512), CAND_CAP * 8, st,
                       corpus, (long long)ld_corpus, (long long)nrows, dim, queries, (long long)ld_queries, fail, (int)nq, k,
                       pl.nslices, fix, reinterpret_cast<int*>(ws + pl.off_ticket), out_scores, (long long*)out_pos,
                       (long long)pos_offset);
    HIP_TRY(hipGetLastError());
    if (n_fixup) HIP_TRY(hipMemcpyAsync(n_fixup, fail + nq, sizeof(int), hipMemcpyDeviceToDevice, st));
    return AMDREC_OK;
}
