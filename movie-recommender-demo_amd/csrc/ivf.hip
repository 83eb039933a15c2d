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
    const long long r0 = list_off[l];
    long long r1 = list_off[l + 1];
    if (r1 <= r0) return;
    // gridDim.z workgroups share the list's rows (whole 32-row rounds each): one request against the reference's default
    // index (nlist 100, nprobe 10: ten lists of 10 000 rows at 1M ads) was ten workgroups streaming 10 MB each - 1 ms
    long long rb = r0;
    if (gridDim.z > 1) {
        const long long per = ((r1 - r0 + gridDim.z - 1) / gridDim.z + 31) / 32 * 32;
        rb = r0 + (long long)blockIdx.z * per;
        if (rb >= r1) return;
        if (rb + per < r1) r1 = rb + per;
    }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < d; i += 256) qv[i] = Q[(long long)q * ldq + i];
    __syncthreads();
    unsigned long long* dst = keys + (long long)q * pool_ld + base[(long long)q * nprobe + p];
    const int d4 = d >> 2;
    constexpr int NW = 4, U = 8;
    for (long long row0 = rb; row0 < r1; row0 += NW * U) {
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
    // filter mode (tau != nullptr): only rows with score >= tau[q * ld_tau] are kept, appended at fill[q]++ (the caller
    // seeds fill[q] with the keys already in the query's pool row); base / pair_p are not used
    const float* tau;
    long long ld_tau;
    unsigned long long* fill;
    template <class A>
    __device__ void operator()(A& acc, float*) const {
        constexpr int TP = A::TP, TQ = A::TQ;
        const int lane = threadIdx.x & 63;
        // Everything the 16 accumulator rows of a tile need from memory - the member's query, its threshold or pool offset -
        // is fetched for all 16 FIRST (independent loads, in flight together), then the rows' positions; round 3 resolved
        // pair_q -> tau / pair_p -> base row by row inside the store loop: 32-48 dependent round trips per workgroup,
        // longer than the tile's MFMAs on short lists.
        long long rowq[TQ], posq[TQ];
#pragma unroll
        for (int j = 0; j < TQ; ++j) {
            rowq[j] = acc.q(j, lane);                             // row inside the list
            posq[j] = rowq[j] < list_rows ? spos[rowq[j]] + pos_offset : -1;
        }
#pragma unroll
        for (int i = 0; i < TP; ++i) {
            long long qv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long jj = acc.p(i, r, lane);          // member of the group (depends on the lane half)
                qv[r] = jj < g ? pair_q[jj] : -1;
            }
            if (tau != nullptr) {
                float tv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) tv[r] = qv[r] >= 0 ? tau[qv[r] * ld_tau] : __builtin_nanf("");   // NaN: no score passes
#pragma unroll
                for (int r = 0; r < 16; ++r) {
#pragma unroll
                    for (int j = 0; j < TQ; ++j) {
                        float sc = acc.v[i][j][r];
                        if (!(sc == sc)) sc = -INFINITY;
                        if (sc >= tv[r] && posq[j] >= 0)
                            (keys + qv[r] * pool_ld)[atomicAdd(&fill[qv[r]], 1ull)] = make_key(sc, (uint32_t)posq[j]);
                    }
                }
                continue;
            }
            long long off[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long jj = acc.p(i, r, lane);
                off[r] = qv[r] >= 0 ? pair_p[jj] : 0;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) off[r] = qv[r] >= 0 ? qv[r] * pool_ld + base[qv[r] * nprobe + off[r]] : -1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (off[r] < 0) continue;
                unsigned long long* dst = keys + off[r];
#pragma unroll
                for (int j = 0; j < TQ; ++j) {
                    if (posq[j] < 0) continue;
                    float sc = acc.v[i][j][r];
                    if (!(sc == sc)) sc = -INFINITY;
                    dst[rowq[j]] = make_key(sc, (uint32_t)posq[j]);
                }
            }
        }
    }
};

// ---- filter mode with a bf16 PREFILTER (round 4) ------------------------------------------------------------------------
// The second phase of the two-phase scan keeps only rows with score >= tau_q and most of a batch's (query, probe) pairs are
// second-phase pairs: 0.64 of the 0.80 ms scan at the per-rank shape of an 8-way sharded 10M index, fp32-MFMA-bound (64
// queries per list).  Their GEMM runs here on a bf16 shadow of the lists and bf16 queries (v_mfma_f32_32x32x16_bf16: 16x the
// fp32 MFMA rate, half the list bytes) and decides only which rows MAY pass: a row with fp32 score >= tau has a bf16 score
// >= tau - eps_q (eps_bound: the flat search's measured-rounding-error bound, with M / D of the list shadow), so every row
// with approx >= tau_lo[q] = tau[q] - eps_q is RE-SCORED in fp32 from the fp32 list row - one wave per row, one explicit
// fma chain per lane and a fixed reduction order, so a row's score does not depend on which workgroup handled it - and
// appended if that exact score passes tau.  The pool holds exact fp32 keys as before; nothing downstream changes.
// Hits are collected per chunk of four accumulator rows in LDS (the staging area, free after the K loop; capacity = the
// chunk's element count: no overflow path), then the workgroup's waves share them.
struct EpiIvfPrefilter {
    static constexpr const char* name = "ivf_scan_bf16";
    static constexpr double out_bytes_per_elem = 0.0;
    static constexpr size_t lds_bytes(int) { return 0; }        // reuses the staging tiles (checked against their size below)
    const long long* spos;      // positions of this list's rows
    long long list_rows;
    const long long* pair_q;    // this list's group: query of member j
    long long g;
    unsigned long long* keys;
    long long pool_ld, pos_offset;
    const float* tau;           // exact filter: fp32 score >= tau[q * ld_tau] (minus the slack below)
    long long ld_tau;
    const float* tau_lo;        // prefilter: bf16 score >= tau_lo[q]
    unsigned long long* fill;
    const float* xs;            // this list's fp32 rows
    long long ld;
    const float* Q;             // fp32 queries
    long long ldq;
    int d;
    // fp32 score of (list row, query) by the 16 lanes of a quarter wave (sub = lane & 15); every lane returns the sum
    __device__ __forceinline__ float rescore16(long long row, long long q, int sub) const {
        const f32x4* xr = reinterpret_cast<const f32x4*>(xs + row * ld);
        const f32x4* qr = reinterpret_cast<const f32x4*>(Q + q * ldq);
        float part = 0.f;
        for (int c = sub; c < (d >> 2); c += 16) {
            const f32x4 x = xr[c], y = qr[c];
            part = __builtin_fmaf(x[3], y[3], __builtin_fmaf(x[2], y[2], __builtin_fmaf(x[1], y[1], __builtin_fmaf(x[0], y[0], part))));
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        return part;
    }
    // the threshold is the k-th score of the first phase, computed by the fp32 MFMA; the re-score sums in another order
    // (1e-7-level differences): a slack towards KEEPING never costs exactness, the select decides
    __device__ __forceinline__ void keep(float sc, long long row, long long q) const {
        if (!(sc == sc)) sc = -INFINITY;
        const float t = tau[q * ld_tau];
        if (sc >= t - 1e-6f * fmaxf(1.f, fabsf(t)) || t == -INFINITY)
            (keys + q * pool_ld)[atomicAdd(&fill[q], 1ull)] = make_key(sc, (uint32_t)(spos[row] + pos_offset));
    }
    template <class A>
    __device__ void operator()(A& acc, float* smem) const {
        constexpr int TP = A::TP, TQ = A::TQ;
        constexpr int NWAVES = A::WP * A::WQ;
        // the nominations of the whole tile in ONE list over the staging tiles; a nomination that does not fit is re-scored
        // by its own lane on the spot (a tile with thousands of nominations: tau is not selective there)
        constexpr int CAP = (int)(((size_t)(A::WP * TP + A::WQ * TQ) * 32 * BK * sizeof(float) - 16) / 8);
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
        int* cnt = reinterpret_cast<int*>(smem);
        unsigned long long* hits = reinterpret_cast<unsigned long long*>(smem + 4);     // (query << 32) | row in the list
        long long rowq[TQ];
#pragma unroll
        for (int j = 0; j < TQ; ++j) rowq[j] = acc.q(j, lane);
        // member -> query -> prefilter threshold for all of this lane's accumulator rows at once: two dependent round
        // trips per tile (fetched round by round they were eight)
        int qv[TP][16];
        float tv[TP][16];
#pragma unroll
        for (int i = 0; i < TP; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long jj = acc.p(i, r, lane);
                qv[i][r] = jj < g ? (int)pair_q[jj] : -1;
            }
#pragma unroll
        for (int i = 0; i < TP; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) tv[i][r] = qv[i][r] >= 0 ? tau_lo[qv[i][r]] : __builtin_nanf("");
        if (tid == 0) *cnt = 0;
        __syncthreads();                                                  // (also: every wave has left the staging tiles)
#pragma unroll
        for (int i = 0; i < TP; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int j = 0; j < TQ; ++j) {
                    const float a = acc.v[i][j][r];
                    // NaN approx (NaN rows / queries): re-score, the exact path decides; NaN threshold: member absent
                    if ((a >= tv[i][r] || (!(a == a) && tv[i][r] == tv[i][r])) && rowq[j] < list_rows) {
                        const int slot = atomicAdd(cnt, 1);
                        if (slot < CAP) {
                            hits[slot] = ((unsigned long long)(uint32_t)qv[i][r] << 32) | (unsigned long long)rowq[j];
                        } else {
                            const f32x4* xr = reinterpret_cast<const f32x4*>(xs + rowq[j] * ld);
                            const f32x4* qr = reinterpret_cast<const f32x4*>(Q + (long long)qv[i][r] * ldq);
                            float sc = 0.f;
                            for (int c = 0; c < (d >> 2); ++c) {
                                const f32x4 x = xr[c], y = qr[c];
                                sc = __builtin_fmaf(x[3], y[3], __builtin_fmaf(x[2], y[2], __builtin_fmaf(x[1], y[1], __builtin_fmaf(x[0], y[0], sc))));
                            }
                            keep(sc, rowq[j], qv[i][r]);
                        }
                    }
                }
        __syncthreads();
        const int n = *cnt < CAP ? *cnt : CAP;
        // four nominations per wave at a time (a quarter wave each), two rounds in flight
        const int grp = lane >> 4, sub = lane & 15;
        int h = wave * 4 + grp;
        for (; h + NWAVES * 4 < n; h += 2 * NWAVES * 4) {
            const unsigned long long e0 = hits[h], e1 = hits[h + NWAVES * 4];
            const long long q0 = (long long)(e0 >> 32), r0 = (long long)(e0 & 0xffffffffull);
            const long long q1 = (long long)(e1 >> 32), r1 = (long long)(e1 & 0xffffffffull);
            const float s0 = rescore16(r0, q0, sub), s1 = rescore16(r1, q1, sub);
            if (sub == 0) { keep(s0, r0, q0); keep(s1, r1, q1); }
        }
        if (h < n) {
            const unsigned long long e0 = hits[h];
            const long long q0 = (long long)(e0 >> 32), r0 = (long long)(e0 & 0xffffffffull);
            const float s0 = rescore16(r0, q0, sub);
            if (sub == 0) keep(s0, r0, q0);
        }
        __syncthreads();                                                  // the staging tiles are the next row tile's again
    }
};

// tau_lo[q] = tau[q] - eps_q for the prefilter above: one wave per query (||q||, ||bf16(q) - q|| from the two copies of the
// query, M / D from the list shadow's max_norm).  tau = -inf (fewer than k rows in the first phase) stays -inf.
__global__ __launch_bounds__(256) void ivf_filter_bounds_kernel(const float* Q, long long ldq, const uint16_t* Q16, long long ldq16,
                                                                int d, long long nq, const float* max_norm, const float* tau,
                                                                long long ld_tau, float* tau_lo) {
    const int lane = threadIdx.x & 63;
    const long long q = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= nq) return;
    float qn = 0.f, dqn = 0.f;
    for (int c = lane; c < d; c += 64) {
        const float v = Q[q * ldq + c];
        const float b = __uint_as_float((uint32_t)Q16[q * ldq16 + c] << 16);
        qn += v * v;
        dqn += (b - v) * (b - v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { qn += __shfl_xor(qn, o, 64); dqn += __shfl_xor(dqn, o, 64); }
    if (lane == 0) {
        const float eps = eps_bound(sqrtf(qn), sqrtf(dqn) * 1.0001f, max_norm[0], max_norm[1], d);
        const float t = tau[q * ld_tau];
        tau_lo[q] = (eps < INFINITY) ? t - eps : -INFINITY;               // NaN / inf norms: everything is re-scored
    }
}

template <class S>
__global__ __launch_bounds__(S::NT, 2) void ivf_group_scan_mixed_kernel(
    const float* xs, long long ld, const uint16_t* xs16, long long ld16, int d, int ksteps, const long long* spos,
    const long long* list_off, const float* Q, long long ldq, const uint16_t* Q16, long long ldq16, const long long* goff,
    const long long* qt_prefix, int nlist, const long long* pair_q, unsigned long long* keys, long long pool_ld,
    long long pos_offset, const float* tau, long long ld_tau, const float* tau_lo, unsigned long long* fill) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const long long y = blockIdx.y;
    if (y >= qt_prefix[nlist]) return;
    int lo = 0, hi = nlist;                                                // the tile's list: 64-ary search (ivf_group_scan_kernel)
    {
        const int lane = threadIdx.x & 63;
        while (hi - lo > 1) {
            const int step = (hi - lo + 63) >> 6;
            const int idx = lo + lane * step;
            const bool le = idx < hi && qt_prefix[idx] <= y;
            const int c = __builtin_popcountll(__ballot(le));
            const int nlo = lo + (c - 1) * step;
            hi = nlo + step < hi ? nlo + step : hi;
            lo = nlo;
        }
    }
    const int l = lo;
    const long long r0 = list_off[l], len = list_off[l + 1] - r0;
    if ((long long)blockIdx.x * S::BQ >= len) return;
    const long long g0 = goff[l], g = goff[l + 1] - g0;
    // the bf16 matrices as float matrices of d / 2 columns (gemm_core.hpp, Shape::BF16)
    GatherRows lp{reinterpret_cast<const float*>(Q16), pair_q + g0, g, (int)(ldq16 / 2), d / 2};
    DenseRows lq{reinterpret_cast<const float*>(xs16 + r0 * ld16), len, (int)(ld16 / 2), d / 2, 30, 1ll << 30};
    EpiIvfPrefilter epi{spos + r0, len, pair_q + g0, g, keys, pool_ld, pos_offset, tau, ld_tau, tau_lo, fill,
                        xs + r0 * ld, ld, Q, ldq, d};
    const long long p0 = (y - qt_prefix[l]) * S::BP;
    for (long long row0 = (long long)blockIdx.x * S::BQ; row0 < len; row0 += (long long)gridDim.x * S::BQ)
        gemm_block<S>(lp, lq, epi, ksteps, p0, row0, smem);
}

// Coarse quantizer as a dense key table (round 3): scores of every query against every centroid -> keys[nq][ld] (64-bit
// (score, ~centroid), the pool format of ivf_select).  The probes used to come from the general exact search
// (amdrec_flat_search over the centroid table): with nlist <= 8192 rows that path files EVERY row as a candidate through
// atomics and sorts 4096 keys per query - 0.27 ms of a 4.6 ms step at nlist 4096 / 512 queries.  Same mainloop, same
// operand roles (lane <-> query, registers <-> centroids) as that search's filter pass, so the scores - and with
// ivf_select's order rule (score desc, position asc) the probes - are bit-identical to it.
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
struct EpiCoarseKeys {
    static constexpr const char* name = "ivf_coarse";
    static constexpr double out_bytes_per_elem = 2.0;
    static constexpr size_t lds_bytes(int) { return 0; }
    unsigned long long* keys;   // [nq][ld]
    long long ld;
    int nq;
    long long nlist;
    template <class A>
    __device__ void operator()(A& acc, float*) const {
        constexpr int TP = A::TP, TQ = A::TQ;
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int j = 0; j < TQ; ++j) {
            const int q = acc.q(j, lane);
            if (q >= nq) continue;
            unsigned long long* dst = keys + (long long)q * ld;
#pragma unroll
            for (int i = 0; i < TP; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const long long p = acc.p(i, 4 * g, lane);     // 4 consecutive centroids p .. p+3
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        if (p + e >= nlist) continue;
                        float s0 = acc.v[i][j][4 * g + e], s1 = acc.v[i][j][4 * g + e + 1];
                        if (!(s0 == s0)) s0 = -INFINITY;            // NaN scores rank last (as in the list scan)
                        if (!(s1 == s1)) s1 = -INFINITY;
                        u64x2 kk;
                        kk[0] = make_key(s0, (uint32_t)(p + e));
                        kk[1] = make_key(s1, (uint32_t)(p + e + 1)); // p+e+1 == nlist (odd nlist): lands in the ld padding
                        *reinterpret_cast<u64x2*>(dst + p + e) = kk;
                    }
                }
        }
    }
};

using ShapeIvf = Shape<2, 2, 1, 4>;     // 64 queries x 256 list rows per workgroup
using ShapeIvf32 = Shape<1, 4, 1, 2>;   // 32 queries x 256 list rows: for sparse groups (few probing queries per list: 512 x 64
                                        // probes over 4096 lists is 8 per list, a 64-query tile would be 12 % full)
// SHORT lists (round 4): a rank of an 8-way sharded 10M index holds ~305 rows of each of the 4096 lists - two 256-row tiles
// of which the second is 19 % full (68 % of the MFMAs multiply padding); 128-row tiles pad the same list to 384 rows.
using ShapeIvfS = Shape<2, 2, 1, 2>;    // 64 queries x 128 list rows
using ShapeIvf32S = Shape<1, 4, 1, 1>;  // 32 queries x 128 list rows
// (double-buffered staging - one barrier per K-step - measured at the per-rank shape: 0.817 against 0.799 ms, not kept)
using ShapeIvfB = Shape<2, 2, 1, 4, false, true>;      // the bf16-prefilter forms of the four shapes (K-step = 64 bf16)
using ShapeIvf32B = Shape<1, 4, 1, 2, false, true>;
using ShapeIvfSB = Shape<2, 2, 1, 2, false, true>;
using ShapeIvf32SB = Shape<1, 4, 1, 1, false, true>;

template <class ShapeIvf>
__global__ __launch_bounds__(ShapeIvf::NT, 2) void ivf_group_scan_kernel(
    const float* xs, long long ld, int d, int ksteps, const long long* spos, const long long* list_off,
    const float* Q, long long ldq, const long long* goff, const long long* qt_prefix, int nlist,
    const long long* pair_q, const long long* pair_p, const long long* base, int nprobe, unsigned long long* keys,
    long long pool_ld, long long pos_offset, const float* tau, long long ld_tau, unsigned long long* fill) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const long long y = blockIdx.y;
    if (y >= qt_prefix[nlist]) return;
    // list l with qt_prefix[l] <= y < qt_prefix[l+1].  A binary search is log2(nlist) DEPENDENT global loads - 12 round trips
    // at nlist 4096, ~10 us in front of a workgroup whose MFMAs take 14 (the per-rank shape of an 8-way sharded 10M index:
    // 305-row lists, tools/shard_step_probe.py --index ivf) - so every wave searches 64-ary: the lanes probe 64 evenly
    // spaced entries of the bracket at once, a ballot counts those at or below y (the prefix is non-decreasing): two or
    // three round trips for any nlist up to 2^18.
    int lo = 0, hi = nlist;
    {
        const int lane = threadIdx.x & 63;
        while (hi - lo > 1) {                                                  // wave-uniform
            const int step = (hi - lo + 63) >> 6;
            const int idx = lo + lane * step;
            const bool le = idx < hi && qt_prefix[idx] <= y;                   // lane 0 probes lo itself: always true
            const int c = __builtin_popcountll(__ballot(le));                  // >= 1
            const int nlo = lo + (c - 1) * step;
            hi = nlo + step < hi ? nlo + step : hi;
            lo = nlo;
        }
    }
    const int l = lo;
    const long long r0 = list_off[l], len = list_off[l + 1] - r0;
    if ((long long)blockIdx.x * ShapeIvf::BQ >= len) return;
    const long long g0 = goff[l], g = goff[l + 1] - g0;
    GatherRows lp{Q, pair_q + g0, g, (int)ldq, d};
    DenseRows lq{xs + r0 * ld, len, (int)ld, d, 30, 1ll << 30};
    EpiIvfKeys epi{spos + r0, len, pair_q + g0, pair_p + g0, g, base, nprobe, keys, pool_ld, pos_offset, tau, ld_tau, fill};
    const long long p0 = (y - qt_prefix[l]) * ShapeIvf::BP;
    // gridDim.x workgroups share the list's row tiles (round 4: the host no longer launches one workgroup per row tile of
    // the LONGEST list - on short lists three of four workgroups found no rows after paying for the tile search)
    for (long long row0 = (long long)blockIdx.x * ShapeIvf::BQ; row0 < len; row0 += (long long)gridDim.x * ShapeIvf::BQ)
        gemm_block<ShapeIvf>(lp, lq, epi, ksteps, p0, row0, smem);     // (ends behind a barrier: the staging area is free)
}

// k largest of keys[q][0..n_q) -> sorted (score desc, position asc); fewer than k -> padded (-inf, -1).
// The k-th largest SCORE is found by a 3-pass radix select over the keys' high words (11 + 11 + 10 bits); every key at or
// above it is then gathered (the keys above it plus its ties) and the bitonic sort of the full 64-bit keys settles the
// ties by position.  Four reads of the pool instead of the seven of a select over the whole 64-bit key; only when the
// boundary score has so many ties that the gather would not fit (k + ties > 2048) does the kernel fall back to that
// 6-pass select of the exact k-th key.
// the k best (by 64-bit key, descending) of row[0 .. n): sorted in buf[0 .. have), have = min(n, k) valid entries first
// (keys of value 0 = empty slots rank last and are reported invalid by write_result).  All 512 threads call.
__device__ __forceinline__ int select_sorted_keys(const unsigned long long* row, long long n, int k, int* hist, int* scratch,
                                                  unsigned long long* buf, unsigned long long* sorted, int* count,
                                                  int sort_mode) {
    const int tid = threadIdx.x;
    // A key of value 0 is an EMPTY slot, never a row (make_key of a NaN-free score is non-zero): the split select's partial
    // lists are zero-padded, so a pool may hold S * k > k entries of which fewer than k are rows.  Empty slots are skipped
    // by the histograms and by the gather (kstar >= 1); when fewer than k rows exist everything non-empty is gathered.
    // (Round 3 counted the padding: the k-th largest of 300 rows + 7700 zeros was 0, the gather admitted all 8000 entries
    // and kept an arbitrary 2048 of them - ADVICE r3.)
    unsigned long long kstar = 1ull;          // gather every row by default (n <= k, or fewer than k non-empty keys)
    if (n > k) {
        uint32_t prefix = 0u, pmask = 0u;
        int rr = k;
        bool all_rows = false;
        const int shifts[3] = {21, 10, 0};
        const int bits[3] = {11, 11, 10};
        for (int pass = 0; pass < 3; ++pass) {
            for (int i = tid; i < 2048; i += 512) hist[i] = 0;
            __syncthreads();
            const uint32_t bm = (1u << bits[pass]) - 1u;
            for (long long i = tid; i < n; i += 512) {
                const unsigned long long key = row[i];
                const uint32_t hi = (uint32_t)(key >> 32);
                if (key != 0ull && (hi & pmask) == prefix) atomicAdd(&hist[(hi >> shifts[pass]) & bm], 1);
            }
            __syncthreads();
            const int bin = find_bin_desc<2048, 512>(hist, rr, scratch);   // -1: fewer than rr non-empty keys (pass 0 only)
            if (bin < 0) {                                        // block-uniform: the pool holds fewer than k rows
                all_rows = true;
                break;
            }
            prefix |= (uint32_t)bin << shifts[pass];
            pmask |= bm << shifts[pass];
            // keys at or above this bin's lower edge: the k - rr above the bin (rr = the k-th key's rank inside it) + the bin's
            // own.  If they fit the sort buffer
            // the remaining passes are not needed - the gather below takes exactly these and the sort finds the k best.
            // Typical pools stop after the FIRST pass: two walks over the pool instead of four (the select of a large
            // pool is bound by those walks: 512 queries x 100 000 keys, 0.46 ms).
            const int at_or_above = (k - rr) + hist[bin];
            __syncthreads();                                     // hist is cleared by the next pass
            if (at_or_above <= 2048) break;                      // block-uniform
        }
        // every key at or above the k-th largest score's (partial) prefix
        kstar = all_rows || prefix == 0u ? 1ull : (unsigned long long)prefix << 32;
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (tid == 0) *count = 0;
        for (int i = tid; i < 2048; i += 512) buf[i] = 0ull;
        __syncthreads();
        for (long long i = tid; i < n; i += 512) {
            const unsigned long long key = row[i];
            if (key >= kstar) {
                const int pos = atomicAdd(count, 1);
                if (pos < 2048) buf[pos] = key;
            }
        }
        __syncthreads();
        if (*count <= 2048 || attempt == 1) break;        // block-uniform
        // too many ties at the boundary score: the exact k-th KEY by a 6-pass select over all 64 bits, then exactly k pass
        unsigned long long prefix = 0ull, pmask = 0ull;
        int rr = k;
        const int shifts[6] = {53, 42, 31, 20, 9, 0};
        const int bits[6] = {11, 11, 11, 11, 11, 9};
        for (int pass = 0; pass < 6; ++pass) {
            __syncthreads();
            for (int i = tid; i < 2048; i += 512) hist[i] = 0;
            __syncthreads();
            const unsigned long long bm = (1ull << bits[pass]) - 1;
            for (long long i = tid; i < n; i += 512) {
                const unsigned long long key = row[i];
                if (key != 0ull && (key & pmask) == prefix) atomicAdd(&hist[(int)((key >> shifts[pass]) & bm)], 1);
            }
            __syncthreads();
            const int bin = find_bin_desc<2048, 512>(hist, rr, scratch);
            prefix |= (unsigned long long)bin << shifts[pass];
            pmask |= bm << shifts[pass];
        }
        kstar = prefix ? prefix : 1ull;           // (more than 2048 non-empty keys were gathered: the k-th exists)
        __syncthreads();
    }
    const int have_all = *count < 2048 ? *count : 2048;
    __syncthreads();
    // gathered keys -> `sorted`, descending.  sort_mode 0: the all-LDS bitonic network on a power of two >= the gathered count
    // (cheap for a few hundred keys, 66 stages - most behind a block barrier - for 2048); 1: the per-wave register sort +
    // cross-run ranking of the flat search's finalize (topk_utils.hpp sort_desc_runs; keys are unique, zeros rank last: a fixed
    // ~4 us whatever the count); 2 (default): the network up to 512 keys, the runs beyond.
    if (sort_mode == 0 || (sort_mode == 2 && have_all <= 512)) {
        int P = 2;
        while (P < have_all) P <<= 1;             // >= k whenever n >= k (k <= 2048)
        bitonic_desc(buf, P);
        for (int i = threadIdx.x; i < have_all; i += 512) sorted[i] = buf[i];
        __syncthreads();
    } else if (have_all <= 1024) {
        sort_desc_runs<2>(buf, sorted, have_all);
    } else {
        sort_desc_runs<4>(buf, sorted, have_all);
    }
    return have_all < k ? have_all : k;
}

__global__ __launch_bounds__(512) void ivf_select_kernel(const unsigned long long* keys, long long pool_ld,
                                                         const long long* n_pool, int k, float* outD,
                                                         long long* outI, int sort_mode) {
    __shared__ int hist[2048];
    __shared__ int scratch[514];
    __shared__ __attribute__((aligned(16))) unsigned long long buf[2048];
    __shared__ __attribute__((aligned(16))) unsigned long long sorted[2048];
    __shared__ int count;
    const long long q = blockIdx.x;
    const int have = select_sorted_keys(keys + q * pool_ld, n_pool[q], k, hist, scratch, buf, sorted, &count, sort_mode);
    write_result(sorted, have, k, q, outD, outI, 0);
}

// Few queries with LARGE pools (one request against nlist 100 / nprobe 10 at 1M ads: 100 000 keys, one workgroup walking
// them four times: 0.14 ms): gridDim.x workgroups per query each select the k best of a slice of the pool into
// part[q][slice][k]; the workgroup that draws the query's last ticket (agent-scope release / acquire, as in the flat
// search's fix-up) selects the k best of those and writes the result.  The k best of the pool are among the k best of
// the slices, and the order is the key order either way: the same result as ivf_select_kernel.  The ticket returns to 0.
__global__ __launch_bounds__(512) void ivf_select_split_kernel(const unsigned long long* keys, long long pool_ld,
                                                               const long long* n_pool, int k, unsigned long long* part,
                                                               int* tickets, float* outD, long long* outI, int sort_mode) {
    __shared__ int hist[2048];
    __shared__ int scratch[514];
    __shared__ __attribute__((aligned(16))) unsigned long long buf[2048];
    __shared__ __attribute__((aligned(16))) unsigned long long sorted[2048];
    __shared__ int count, last_sh;
    const long long q = blockIdx.y;
    const int s = blockIdx.x, S = gridDim.x, tid = threadIdx.x;
    const long long n = n_pool[q];
    const long long per = (n + S - 1) / S;
    const long long lo = s * per < n ? s * per : n, hi = lo + per < n ? lo + per : n;
    const int have = select_sorted_keys(keys + q * pool_ld + lo, hi - lo, k, hist, scratch, buf, sorted, &count, sort_mode);
    unsigned long long* mine = part + (q * S + s) * k;
    for (int i = tid; i < k; i += 512) mine[i] = i < have ? sorted[i] : 0ull;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int t = __hip_atomic_fetch_add(&tickets[q], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_sh = t == S - 1;
        if (t == S - 1) {
            __hip_atomic_store(&tickets[q], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // ready for the next call
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (!last_sh) return;                                      // block-uniform
    __syncthreads();                                           // (`sorted` was read above by every thread)
    const int have2 = select_sorted_keys(part + q * S * k, (long long)S * k, k, hist, scratch, buf, sorted, &count, sort_mode);
    write_result(sorted, have2, k, q, outD, outI, 0);
}

// ---- index BUILD: max-inner-product assignment and Lloyd iterations (faiss_retrieval.py:83-95, :118: IndexIVFFlat with an
// IndexFlatIP quantizer trains k-means on first add and files every vector under the centroid of largest inner product) ----
// Assignment = the GEMM mainloop (P = centroids: registers, Q = rows: lanes, fp32 MFMA: exact fma chains, so a row's
// scores do not depend on tiling) with an arg-max epilogue: per row the best (score, centroid) key of the block's 256
// centroids, folded across the centroid tiles with one 64-bit atomicMax per (wave, row).  key = (orderable score << 32) |
// ~centroid: larger = better, equal scores -> LOWER centroid index (deterministic; NaN scores never win).
struct EpiArgmax {
    static constexpr const char* name = "ivf_assign";
    static constexpr double out_bytes_per_elem = 0.0;
    static constexpr size_t lds_bytes(int) { return 0; }
    unsigned long long* keys;   // [rows], zero-initialised
    long long rows;
    int nlist;
    template <class A>
    __device__ void operator()(A& acc, float*) const {
        constexpr int TP = A::TP, TQ = A::TQ;
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int j = 0; j < TQ; ++j) {
            unsigned long long best = 0ull;
#pragma unroll
            for (int i = 0; i < TP; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int p = acc.p(i, r, lane);
                    const float sc = acc.v[i][j][r];
                    if (p < nlist && sc == sc) {
                        const unsigned long long key = make_key(sc, (uint32_t)p);
                        best = key > best ? key : best;
                    }
                }
            const unsigned long long other = ((unsigned long long)__shfl_xor((unsigned)(best >> 32), 32, 64) << 32) |
                                             (unsigned long long)__shfl_xor((unsigned)best, 32, 64);
            best = other > best ? other : best;
            const long long q = acc.q(j, lane);
            if (lane < 32 && q < rows && best != 0ull) atomicMax(&keys[q], best);
        }
    }
};

__global__ void ivf_decode_assign_kernel(const unsigned long long* keys, long long rows, long long* assign, float* score) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    const unsigned long long k = keys[i];
    assign[i] = k ? (long long)key_pos(k) : 0ll;          // a row whose scores are all NaN goes to list 0
    if (score) score[i] = k ? key_score(k) : -INFINITY;
}

// Centroid update, DETERMINISTIC: every coordinate is accumulated as a 64-bit fixed-point integer (x * 2^40, exact for
// fp32 unit-vector coordinates down to 2^-16; integer addition is associative, so the sum does not depend on the order
// the atomics land in - float atomics would make the trained centroids, and with them every IVF result, vary run to run).
constexpr float IVF_FIX = 1099511627776.0f;      // 2^40; up to 2^22 unit rows per centroid before 2^62
__global__ __launch_bounds__(256) void ivf_accumulate_kernel(const float* x, long long rows, long long ld, int d,
                                                             const long long* assign, long long* sums, int* counts) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= rows) return;
    const long long a = assign[r];
    for (int c = lane; c < d; c += 64) {
        const float v = x[r * ld + c];
        if (v == v) atomicAdd(reinterpret_cast<unsigned long long*>(&sums[a * d + c]), (unsigned long long)__float2ll_rn(v * IVF_FIX));
    }
    if (lane == 0) atomicAdd(&counts[a], 1);
}
// new centroid = sum / |sum| (spherical k-means: the quantizer compares inner products); empty clusters keep the old one
__global__ __launch_bounds__(256) void ivf_finish_centroids_kernel(const long long* sums, const int* counts, int nlist, int d,
                                                                   float* cent, long long ldc) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= nlist || counts[c] == 0) return;
    double ss = 0.0;
    for (int k = lane; k < d; k += 64) {
        const double v = (double)sums[(long long)c * d + k];
        ss += v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if (!(ss > 0.0)) return;
    const double inv = 1.0 / sqrt(ss);
    for (int k = lane; k < d; k += 64) cent[c * ldc + k] = (float)((double)sums[(long long)c * d + k] * inv);
}

// ---- query-time plumbing, sync-free (replaces torch argsort / bincount / cumsum per search call) ---------------------
// pool layout of every query + rank of every (query, probe) pair inside its list's group (counting sort, pass 1)
__global__ void ivf_group_count_kernel(const long long* probes, long long ldp, int nprobe, long long npairs, int nlist, int* cnt,
                                       int* rank) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npairs) return;
    const long long l = probes[(i / nprobe) * ldp + i % nprobe];
    rank[i] = atomicAdd(&cnt[(l < 0 || l >= nlist) ? nlist : (int)l], 1);
}
__global__ void ivf_pool_layout_kernel(const long long* probes, long long ldp, long long m, int nprobe, int nlist,
                                       const long long* list_len, long long* base, long long* n_pool) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= m) return;
    long long run = 0;
    for (int p = 0; p < nprobe; ++p) {
        const long long l = probes[q * ldp + p];
        base[q * nprobe + p] = run;
        run += (l >= 0 && l < nlist) ? list_len[l] : 0;
    }
    n_pool[q] = run;
}
// exclusive scans over the lists: group offsets and query-tile offsets (tiles of qtile queries; one block)
__global__ __launch_bounds__(1024) void ivf_group_prefix_kernel(const int* cnt, int nlist, int qtile, long long* goff,
                                                                 long long* qtp) {
    __shared__ long long sa[1024], sb[1024];
    const int tid = threadIdx.x;
    const int per = (nlist + 1023) / 1024;
    const int lo = tid * per, hi = (lo + per < nlist) ? lo + per : nlist;
    long long a = 0, b = 0;
    for (int l = lo; l < hi; ++l) { a += cnt[l]; b += (cnt[l] + qtile - 1) / qtile; }
    sa[tid] = a; sb[tid] = b;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {                  // Hillis-Steele inclusive scan of the 1024 partials
        const long long va = tid >= o ? sa[tid - o] : 0, vb = tid >= o ? sb[tid - o] : 0;
        __syncthreads();
        sa[tid] += va; sb[tid] += vb;
        __syncthreads();
    }
    long long ra = sa[tid] - a, rb = sb[tid] - b;         // exclusive prefix of this thread's range
    for (int l = lo; l < hi; ++l) {
        goff[l] = ra; qtp[l] = rb;
        ra += cnt[l]; rb += (cnt[l] + qtile - 1) / qtile;
    }
    if (tid == 1023) { goff[nlist] = sa[1023]; qtp[nlist] = sb[1023]; }
}
__global__ void ivf_group_scatter_kernel(const long long* probes, long long ldp, long long npairs, int nprobe, int nlist,
                                         const int* rank, const long long* goff, long long* pair_q, long long* pair_p) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npairs) return;
    const long long l = probes[(i / nprobe) * ldp + i % nprobe];
    if (l < 0 || l >= nlist) return;
    const long long pos = goff[l] + rank[i];
    pair_q[pos] = i / nprobe;
    pair_p[pos] = i % nprobe;
}

}  // namespace amdrec

using namespace amdrec;

static int assign_impl(const float* x, long long rows, long long ld, int dim, const float* cent, int nlist, long long ldc,
                       long long* assign, float* score, unsigned long long* keys, hipStream_t st) {
    HIP_TRY(hipMemsetAsync(keys, 0, (size_t)rows * 8, st));
    DenseRows lp{cent, nlist, (int)ldc, dim, 30, 1ll << 30};
    DenseRows lq{x, rows, (int)ld, dim, 30, 1ll << 30};
    EpiArgmax epi{keys, rows, nlist};
    HIP_TRY((launch_gemm<Shape<2, 2, 4, 2>, true>(lp, lq, epi, dim, nlist, rows, st)));
    hipLaunchKernelGGL(ivf_decode_assign_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, keys, rows, assign, score);
    HIP_TRY(hipGetLastError());
    return AMDREC_OK;
}

static int assign_check(const float* x, int64_t rows, int64_t ld, int dim, const float* cent, int nlist, int64_t ldc) {
    REQUIRE(dim >= 4 && dim % 4 == 0 && dim <= 2048, "dim=%d must be a multiple of 4 in [4,2048]", dim);
    REQUIRE(nlist >= 1 && nlist <= (1 << 20), "nlist out of range");
    REQUIRE(rows >= 0 && rows < (1ll << 31) - 1024, "rows out of range");
    REQUIRE(ld >= dim && ld % 4 == 0 && ldc >= dim && ldc % 4 == 0 && ld < (1ll << 31) && ldc < (1ll << 31), "bad leading dimension");
    REQUIRE(rows == 0 || (x && cent), "null pointer");
    REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)cent % 16) == 0, "x / centroids must be 16-byte aligned");
    return AMDREC_OK;
}

extern "C" int amdrec_ivf_assign(const float* x, int64_t rows, int64_t ld, int dim, const float* centroids, int nlist,
                                 int64_t ld_centroids, int64_t* assign, float* best_score, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    int rc = assign_check(x, rows, ld, dim, centroids, nlist, ld_centroids);
    if (rc) return rc;
    if (rows == 0) return AMDREC_OK;
    REQUIRE(assign != nullptr, "assign is null");
    const size_t need = align_up((size_t)rows * 8, 256);
    if (!workspace || workspace_bytes < need)
        return set_error(AMDREC_EWORKSPACE, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    return assign_impl(x, rows, ld, dim, centroids, nlist, ld_centroids, (long long*)assign, best_score,
                       reinterpret_cast<unsigned long long*>(workspace), reinterpret_cast<hipStream_t>(stream));
}

extern "C" int amdrec_ivf_kmeans_workspace(int64_t rows, int dim, int nlist, size_t* bytes) {
    REQUIRE(bytes != nullptr && rows >= 0 && dim >= 4 && nlist >= 1, "bad arguments");
    *bytes = align_up((size_t)rows * 8, 256) + align_up((size_t)rows * 8, 256) + align_up((size_t)nlist * dim * 8, 256) +
             align_up((size_t)nlist * 4, 256);
    return AMDREC_OK;
}

extern "C" int amdrec_ivf_kmeans_step(const float* x, int64_t rows, int64_t ld, int dim, float* centroids, int nlist,
                                      int64_t ld_centroids, void* workspace, size_t workspace_bytes, void* stream) {
    int rc = assign_check(x, rows, ld, dim, centroids, nlist, ld_centroids);
    if (rc) return rc;
    if (rows == 0) return AMDREC_OK;
    size_t need = 0;
    amdrec_ivf_kmeans_workspace(rows, dim, nlist, &need);
    if (!workspace || workspace_bytes < need)
        return set_error(AMDREC_EWORKSPACE, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char* ws = reinterpret_cast<char*>(workspace);
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(ws);
    long long* assign = reinterpret_cast<long long*>(ws + align_up((size_t)rows * 8, 256));
    long long* sums = reinterpret_cast<long long*>(ws + 2 * align_up((size_t)rows * 8, 256));
    int* counts = reinterpret_cast<int*>(reinterpret_cast<char*>(sums) + align_up((size_t)nlist * dim * 8, 256));
    rc = assign_impl(x, rows, ld, dim, centroids, nlist, ld_centroids, assign, nullptr, keys, st);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(sums, 0, align_up((size_t)nlist * dim * 8, 256) + (size_t)nlist * 4, st));
    hipLaunchKernelGGL(ivf_accumulate_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, (long long)rows,
                       (long long)ld, dim, assign, sums, counts);
    hipLaunchKernelGGL(ivf_finish_centroids_kernel, dim3((unsigned)((nlist + 3) / 4)), dim3(256), 0, st, sums, counts, nlist,
                       dim, centroids, (long long)ld_centroids);
    HIP_TRY(hipGetLastError());
    return AMDREC_OK;
}

extern "C" int amdrec_ivf_group(const int64_t* probes, int64_t ld_probes, int64_t nq, int nprobe, int nlist, const int64_t* list_len,
                                int64_t* pool_base, int64_t* pool_count, int64_t* pair_query, int64_t* pair_probe,
                                int64_t* group_off, int64_t* qtile_prefix, int qtile, void* workspace,
                                size_t workspace_bytes, void* stream) {
    REQUIRE(nprobe >= 1 && nlist >= 1 && nlist <= (1 << 20), "bad nlist/nprobe");
    REQUIRE(qtile == 32 || qtile == 64, "qtile must be 32 or 64");
    REQUIRE(ld_probes >= nprobe, "ld_probes < nprobe");
    if (nq <= 0) return AMDREC_OK;
    REQUIRE(nq * (int64_t)nprobe < (1ll << 31), "too many (query, probe) pairs for one call");
    REQUIRE(probes && list_len && pool_base && pool_count && pair_query && pair_probe && group_off && qtile_prefix, "null pointer");
    const long long npairs = nq * nprobe;
    const size_t need = align_up((size_t)(nlist + 1) * 4, 256) + align_up((size_t)npairs * 4, 256);
    if (!workspace || workspace_bytes < need)
        return set_error(AMDREC_EWORKSPACE, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    int* cnt = reinterpret_cast<int*>(workspace);
    int* rank = reinterpret_cast<int*>(reinterpret_cast<char*>(workspace) + align_up((size_t)(nlist + 1) * 4, 256));
    HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)(nlist + 1) * 4, st));
    hipLaunchKernelGGL(ivf_group_count_kernel, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, st,
                       (const long long*)probes, (long long)ld_probes, nprobe, npairs, nlist, cnt, rank);
    hipLaunchKernelGGL(ivf_pool_layout_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, (const long long*)probes,
                       (long long)ld_probes, (long long)nq, nprobe, nlist, (const long long*)list_len, (long long*)pool_base,
                       (long long*)pool_count);
    hipLaunchKernelGGL(ivf_group_prefix_kernel, dim3(1), dim3(1024), 0, st, cnt, nlist, qtile, (long long*)group_off,
                       (long long*)qtile_prefix);
    hipLaunchKernelGGL(ivf_group_scatter_kernel, dim3((unsigned)((npairs + 255) / 256)), dim3(256), 0, st,
                       (const long long*)probes, (long long)ld_probes, npairs, nprobe, nlist, rank, (const long long*)group_off,
                       (long long*)pair_query, (long long*)pair_probe);
    HIP_TRY(hipGetLastError());
    return AMDREC_OK;
}

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
    long long splits = 2048 / ((long long)nprobe * nq);            // ~2048 workgroups in all
    splits = splits < 1 ? 1 : (splits > 64 ? 64 : splits);
    ProfScope prof("ivf_scan_pairs", 0.0, 0.0, reinterpret_cast<hipStream_t>(stream));
    hipLaunchKernelGGL(ivf_scan_kernel, dim3(nprobe, (unsigned)nq, (unsigned)splits), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       lists, (long long)ld, dim, (const long long*)row_pos, (const long long*)list_off, queries,
                       (long long)ld_queries, (const long long*)probes, (const long long*)pool_base, nprobe,
                       (unsigned long long*)pool_keys, (long long)pool_ld, (long long)pos_offset);
    HIP_TRY(hipGetLastError());
    return AMDREC_OK;
}

template <class S>
static hipError_t launch_group_scan(const char* tag, const float* lists, long long ld, int dim, const long long* row_pos,
                                    const long long* list_off, int nlist, long long max_list_rows, const float* queries,
                                    long long ld_queries, const long long* group_off, const long long* qtile_prefix,
                                    long long qtile_bound, const long long* pair_query, const long long* pair_probe,
                                    const long long* pool_base, int nprobe, unsigned long long* pool_keys, long long pool_ld,
                                    long long pos_offset, const float* tau, long long ld_tau, unsigned long long* fill,
                                    hipStream_t st) {
    static PerDeviceOnce attr_done;
    if (attr_done.pending()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ivf_group_scan_kernel<S>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done.mark();
    }
    // workgroups per (list, query tile): an eighth of the LONGEST list's row tiles, each looping over its share of the list's
    // tiles.  Round 3 launched one workgroup per row tile of the longest list for every (list, query tile): list lengths
    // spread 0 .. 4x the mean, so three of four workgroups found no rows - after paying for the tile search - and at the
    // per-rank shape of an 8-way sharded 10M index (305-row lists) 48 000 of 61 000 workgroups were empty: scan 1.82 ->
    // 0.80 ms (profiles/r04_shard_ivf_g8_after.log).  AMDREC_IVF_GX overrides for A/B runs.
    static const long long gx_env = [] { const char* v = getenv("AMDREC_IVF_GX"); return v ? atoll(v) : 0ll; }();
    long long gxl = (max_list_rows + S::BQ - 1) / S::BQ;
    const long long cap = gx_env > 0 ? gx_env : (gxl + 7) / 8;
    if (gxl > cap) gxl = cap;
    const unsigned gx = (unsigned)gxl;
    ProfScope prof(tag, 0.0, 0.0, st);
    hipLaunchKernelGGL(ivf_group_scan_kernel<S>, dim3(gx, (unsigned)qtile_bound), dim3(S::NT), S::LDS_BYTES, st, lists, ld, dim,
                       (dim + BK - 1) / BK, row_pos, list_off, queries, ld_queries, group_off, qtile_prefix, nlist, pair_query,
                       pair_probe, pool_base, nprobe, pool_keys, pool_ld, pos_offset, tau, ld_tau, fill);
    return hipGetLastError();
}

extern "C" int amdrec_ivf_scan_grouped(const float* lists, int64_t ld, int dim, const int64_t* row_pos,
                                       const int64_t* list_off, int nlist, int64_t max_list_rows,
                                       const float* queries, int64_t ld_queries, const int64_t* group_off,
                                       const int64_t* qtile_prefix, int64_t qtile_bound, int qtile,
                                       const int64_t* pair_query, const int64_t* pair_probe, const int64_t* pool_base,
                                       int nprobe, uint64_t* pool_keys, int64_t pool_ld, int64_t pos_offset,
                                       const float* tau, int64_t ld_tau, int64_t* pool_fill, void* stream) {
    REQUIRE(dim >= 4 && dim % 4 == 0 && dim <= 2048, "dim=%d must be a multiple of 4 in [4,2048]", dim);
    REQUIRE(nlist >= 1 && nprobe >= 1, "bad nlist/nprobe");
    REQUIRE(qtile == 32 || qtile == 64, "qtile must be 32 or 64 (the value given to amdrec_ivf_group)");
    REQUIRE((tau == nullptr) == (pool_fill == nullptr) && (tau == nullptr || ld_tau >= 1), "tau and pool_fill go together");
    if (qtile_bound <= 0 || max_list_rows <= 0) return AMDREC_OK;
    REQUIRE(qtile_bound <= 65535, "too many (list, query-tile) groups for one launch: chunk the queries");
    REQUIRE(lists && row_pos && list_off && queries && group_off && qtile_prefix && pair_query && pair_probe &&
                (pool_base || tau) && pool_keys, "null pointer");
    REQUIRE(ld % 4 == 0 && ld >= dim && ld_queries >= dim && ld_queries % 4 == 0, "bad leading dimension");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    hipError_t e;
    // row tile: 128 rows when even the longest list is short (the tile count of a launch is sized by it), else 256.
    // AMDREC_IVF_SHORT_ROWS overrides the limit (0: never) for A/B runs (tools/shard_step_probe.py --index ivf).
    static const long long short_rows = [] {
        const char* v = getenv("AMDREC_IVF_SHORT_ROWS");
        return v ? atoll(v) : 1536ll;
    }();
    const bool short_lists = max_list_rows <= short_rows;
#define AMDREC_IVF_GROUP_SCAN(SHAPE, TAG)                                                                                       \
    launch_group_scan<SHAPE>(TAG, lists, ld, dim, (const long long*)row_pos, (const long long*)list_off, nlist, max_list_rows, \
                             queries, ld_queries, (const long long*)group_off, (const long long*)qtile_prefix, qtile_bound,    \
                             (const long long*)pair_query, (const long long*)pair_probe, (const long long*)pool_base, nprobe,  \
                             (unsigned long long*)pool_keys, pool_ld, pos_offset, tau, (long long)ld_tau,                      \
                             (unsigned long long*)pool_fill, st)
    if (qtile == 64) e = short_lists ? AMDREC_IVF_GROUP_SCAN(ShapeIvfS, "ivf_scan_grouped_64x128") : AMDREC_IVF_GROUP_SCAN(ShapeIvf, "ivf_scan_grouped_64x256");
    else             e = short_lists ? AMDREC_IVF_GROUP_SCAN(ShapeIvf32S, "ivf_scan_grouped_32x128") : AMDREC_IVF_GROUP_SCAN(ShapeIvf32, "ivf_scan_grouped_32x256");
#undef AMDREC_IVF_GROUP_SCAN
    HIP_TRY(e);
    return AMDREC_OK;
}

template <class S>
static hipError_t launch_group_scan_mixed(const char* tag, const float* lists, long long ld, const uint16_t* lists16, long long ld16,
                                          int dim, const long long* row_pos, const long long* list_off, int nlist,
                                          long long max_list_rows, const float* queries, long long ldq, const uint16_t* q16,
                                          long long ldq16, const long long* group_off, const long long* qtile_prefix,
                                          long long qtile_bound, const long long* pair_query, unsigned long long* pool_keys,
                                          long long pool_ld, long long pos_offset, const float* tau, long long ld_tau,
                                          const float* tau_lo, unsigned long long* fill, hipStream_t st) {
    static PerDeviceOnce attr_done;
    if (attr_done.pending()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ivf_group_scan_mixed_kernel<S>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_done.mark();
    }
    long long gxl = (max_list_rows + S::BQ - 1) / S::BQ;                  // as launch_group_scan
    const long long cap = (gxl + 7) / 8;
    if (gxl > cap) gxl = cap;
    ProfScope prof(tag, 0.0, 0.0, st);
    hipLaunchKernelGGL(ivf_group_scan_mixed_kernel<S>, dim3((unsigned)gxl, (unsigned)qtile_bound), dim3(S::NT), S::LDS_BYTES, st,
                       lists, ld, lists16, ld16, dim, (dim / 2 + BK - 1) / BK, row_pos, list_off, queries, ldq, q16, ldq16,
                       group_off, qtile_prefix, nlist, pair_query, pool_keys, pool_ld, pos_offset, tau, ld_tau, tau_lo, fill);
    return hipGetLastError();
}

extern "C" int amdrec_ivf_filter_bounds(const float* queries, int64_t nq, int64_t ld_queries, int dim,
                                        const uint16_t* queries_bf16, int64_t ld_queries_bf16, const float* max_norm,
                                        const float* tau, int64_t ld_tau, float* tau_lo, void* stream) {
    REQUIRE(dim >= 8 && dim % 8 == 0 && dim <= 2048, "dim=%d must be a multiple of 8 in [8,2048]", dim);
    if (nq <= 0) return AMDREC_OK;
    REQUIRE(queries && queries_bf16 && max_norm && tau && tau_lo, "null pointer");
    REQUIRE(ld_queries >= dim && ld_queries_bf16 >= dim && ld_tau >= 1, "bad leading dimension");
    ProfScope prof("ivf_filter_bounds", 0.0, 0.0, reinterpret_cast<hipStream_t>(stream));
    hipLaunchKernelGGL(ivf_filter_bounds_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       queries, (long long)ld_queries, queries_bf16, (long long)ld_queries_bf16, dim, (long long)nq, max_norm, tau,
                       (long long)ld_tau, tau_lo);
    HIP_TRY(hipGetLastError());
    return AMDREC_OK;
}

extern "C" int amdrec_ivf_scan_grouped_mixed(const float* lists, int64_t ld, const uint16_t* lists_bf16, int64_t ld_bf16, int dim,
                                             const int64_t* row_pos, const int64_t* list_off, int nlist, int64_t max_list_rows,
                                             const float* queries, int64_t ld_queries, const uint16_t* queries_bf16,
                                             int64_t ld_queries_bf16, const int64_t* group_off, const int64_t* qtile_prefix,
                                             int64_t qtile_bound, int qtile, const int64_t* pair_query, uint64_t* pool_keys,
                                             int64_t pool_ld, int64_t pos_offset, const float* tau, int64_t ld_tau,
                                             const float* tau_lo, int64_t* pool_fill, void* stream) {
    REQUIRE(dim >= 8 && dim % 8 == 0 && dim <= 2048, "dim=%d must be a multiple of 8 in [8,2048]", dim);
    REQUIRE(nlist >= 1, "bad nlist");
    REQUIRE(qtile == 32 || qtile == 64, "qtile must be 32 or 64 (the value given to amdrec_ivf_group)");
    if (qtile_bound <= 0 || max_list_rows <= 0) return AMDREC_OK;
    REQUIRE(qtile_bound <= 65535, "too many (list, query-tile) groups for one launch: chunk the queries");
    REQUIRE(lists && lists_bf16 && row_pos && list_off && queries && queries_bf16 && group_off && qtile_prefix && pair_query &&
                pool_keys && tau && tau_lo && pool_fill, "null pointer");
    REQUIRE(ld % 4 == 0 && ld >= dim && ld_queries >= dim && ld_queries % 4 == 0 && ld_bf16 >= dim && ld_bf16 % 8 == 0 &&
                ld_queries_bf16 >= dim && ld_queries_bf16 % 8 == 0 && ld_tau >= 1, "bad leading dimension");
    REQUIRE(((uintptr_t)lists % 16) == 0 && ((uintptr_t)lists_bf16 % 16) == 0 && ((uintptr_t)queries % 16) == 0 &&
                ((uintptr_t)queries_bf16 % 16) == 0, "lists / queries must be 16-byte aligned");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const bool short_lists = max_list_rows <= 1536;
    hipError_t e;
#define AMDREC_IVF_GROUP_SCAN_MIXED(SHAPE, TAG)                                                                                     \
    launch_group_scan_mixed<SHAPE>(TAG, lists, ld, lists_bf16, ld_bf16, dim, (const long long*)row_pos, (const long long*)list_off, \
                                   nlist, max_list_rows, queries, ld_queries, queries_bf16, ld_queries_bf16,                         \
                                   (const long long*)group_off, (const long long*)qtile_prefix, qtile_bound,                         \
                                   (const long long*)pair_query, (unsigned long long*)pool_keys, pool_ld, pos_offset, tau,           \
                                   (long long)ld_tau, tau_lo, (unsigned long long*)pool_fill, st)
    if (qtile == 64) e = short_lists ? AMDREC_IVF_GROUP_SCAN_MIXED(ShapeIvfSB, "ivf_scan_grouped_bf16_64x128") : AMDREC_IVF_GROUP_SCAN_MIXED(ShapeIvfB, "ivf_scan_grouped_bf16_64x256");
    else             e = short_lists ? AMDREC_IVF_GROUP_SCAN_MIXED(ShapeIvf32SB, "ivf_scan_grouped_bf16_32x128") : AMDREC_IVF_GROUP_SCAN_MIXED(ShapeIvf32B, "ivf_scan_grouped_bf16_32x256");
#undef AMDREC_IVF_GROUP_SCAN_MIXED
    HIP_TRY(e);
    return AMDREC_OK;
}

extern "C" int amdrec_ivf_coarse_keys(const float* centroids, int nlist, int64_t ld_centroids, int dim, const float* queries,
                                      int64_t nq, int64_t ld_queries, uint64_t* keys, int64_t ld_keys, void* stream) {
    REQUIRE(dim >= 4 && dim % 4 == 0 && dim <= 2048, "dim=%d must be a multiple of 4 in [4,2048]", dim);
    REQUIRE(nlist >= 1 && nlist <= (1 << 20), "nlist out of range");
    if (nq <= 0) return AMDREC_OK;
    REQUIRE(centroids && queries && keys, "null pointer");
    REQUIRE(ld_centroids >= dim && ld_centroids % 4 == 0 && ld_queries >= dim && ld_queries % 4 == 0 && ld_keys >= (nlist + 1) / 2 * 2 &&
                ld_keys % 2 == 0,
            "bad leading dimension");
    REQUIRE(((uintptr_t)centroids % 16) == 0 && ((uintptr_t)queries % 16) == 0, "centroids / queries must be 16-byte aligned");
    REQUIRE(((uintptr_t)keys % 16) == 0, "keys must be 16-byte aligned");
    REQUIRE(nq < (1ll << 24), "nq out of range");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    DenseRows lp{centroids, nlist, (int)ld_centroids, dim, 30, 1ll << 30};
    DenseRows lq{queries, nq, (int)ld_queries, dim, 30, 1ll << 30};
    EpiCoarseKeys epi{reinterpret_cast<unsigned long long*>(keys), (long long)ld_keys, (int)nq, (long long)nlist};
    hipError_t e;                                       // the query-tile shapes of amdrec_flat_search's filter pass
    if (nq > 64)      e = launch_gemm<Shape<2, 2, 4, 2>, false>(lp, lq, epi, dim, nlist, nq, st);
    else if (nq > 32) e = launch_gemm<Shape<4, 1, 2, 2>, false>(lp, lq, epi, dim, nlist, nq, st);
    else              e = launch_gemm<Shape<4, 1, 2, 1>, false>(lp, lq, epi, dim, nlist, nq, st);
    HIP_TRY(e);
    return AMDREC_OK;
}

// sort of the gathered keys in the select kernels: 2 = adaptive (default); AMDREC_IVF_SORT = 0 / 1 force one form (A/B runs)
static int ivf_sort_mode() {
    static const int m = [] { const char* v = getenv("AMDREC_IVF_SORT"); return v ? atoi(v) : 2; }();
    return m;
}

extern "C" int amdrec_ivf_select(const uint64_t* pool_keys, int64_t pool_ld, const int64_t* pool_count, int64_t nq,
                                 int k, float* out_scores, int64_t* out_pos, void* stream) {
    REQUIRE(k >= 1 && k <= AMDREC_MAX_K, "k=%d outside [1,%d]", k, AMDREC_MAX_K);
    if (nq <= 0) return AMDREC_OK;
    REQUIRE(pool_keys && pool_count && out_scores && out_pos, "null pointer");
    ProfScope prof("ivf_select", 0.0, 0.0, reinterpret_cast<hipStream_t>(stream));
    hipLaunchKernelGGL(ivf_select_kernel, dim3((unsigned)nq), dim3(512), 0, reinterpret_cast<hipStream_t>(stream),
                       (const unsigned long long*)pool_keys, (long long)pool_ld, (const long long*)pool_count, k,
                       out_scores, (long long*)out_pos, ivf_sort_mode());
    HIP_TRY(hipGetLastError());
    return AMDREC_OK;
}

extern "C" int amdrec_ivf_select_split(const uint64_t* pool_keys, int64_t pool_ld, const int64_t* pool_count, int64_t nq,
                                       int k, int slices, float* out_scores, int64_t* out_pos, void* workspace,
                                       size_t workspace_bytes, int32_t* tickets, void* stream) {
    REQUIRE(k >= 1 && k <= AMDREC_MAX_K, "k=%d outside [1,%d]", k, AMDREC_MAX_K);
    REQUIRE(slices >= 1 && slices <= 1024 && nq <= 65535, "slices / nq out of range");
    if (nq <= 0) return AMDREC_OK;
    REQUIRE(pool_keys && pool_count && out_scores && out_pos && tickets, "null pointer");
    const size_t need = (size_t)nq * slices * k * 8;
    if (!workspace || workspace_bytes < need)
        return set_error(AMDREC_EWORKSPACE, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    REQUIRE(((uintptr_t)workspace % 16) == 0, "workspace must be 16-byte aligned");
    ProfScope prof("ivf_select", 0.0, 0.0, reinterpret_cast<hipStream_t>(stream));
    hipLaunchKernelGGL(ivf_select_split_kernel, dim3((unsigned)slices, (unsigned)nq), dim3(512), 0,
                       reinterpret_cast<hipStream_t>(stream), (const unsigned long long*)pool_keys, (long long)pool_ld,
                       (const long long*)pool_count, k, reinterpret_cast<unsigned long long*>(workspace), tickets, out_scores,
                       (long long*)out_pos, ivf_sort_mode());
    HIP_TRY(hipGetLastError());
    return AMDREC_OK;
}
