// fp32-MFMA "NT" GEMM mainloop for gfx950:   C[p][q] = sum_k P[p][k] * Q[q][k]
//
// Every GEMM-shaped op of the hot path (inner-product search, tower MLPs, ranker layers)
// is this loop with a different operand loader and epilogue.  Both operands are
// K-contiguous ("row of P dot row of Q").
//
// Hardware mapping (MI355X_MICROARCH.md / cdna_hip_programming.md §3 "FP32-input MFMA"):
//  * v_mfma_f32_32x32x2_f32: exact fp32 (fmaf chain), 64 FLOP/clk/SIMD = the fp32 peak.
//    A operand: lane l holds P[p = l&31][k = l>>5]; B operand: Q[q = l&31][k = l>>5].
//    C/D: lane&31 = q column, row p = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
//  * workgroups of WP x WQ waves (4 waves in every shape used), each wave owns TP x TQ tiles of
//    32x32 (acc in VGPRs: 16*TP*TQ per lane); two workgroups per CU = 2 waves per SIMD.
//  * Operands are staged global -> registers -> LDS in full 128-byte row segments
//    (BK = 32 floats); the next tile's global loads are in flight under the current tile's MFMAs.  The LDS image is
//    [row][32 floats] with the 16-byte chunk index XOR-swizzled by (row>>1)&7 so that a
//    fragment read (ds_read_b128, 16-lane groups of distinct rows, same chunk) is
//    bank-conflict free (guide §2 LDS, T2).
//  * Fragment k-order: a lane's ds_read_b128 returns 4 consecutive k (lane half h reads
//    chunk 2c+h); MFMA step s of chunk pair c therefore multiplies k = 8c+4h+s.  P and Q
//    use the same permutation, so the sum over k is complete; only the fp32 summation
//    order differs from a sequential loop.
#pragma once
#include "common.hpp"

namespace amdrec {

constexpr int BK = 32;          // floats per K-step

__device__ __forceinline__ int lds_slot(int row, int chunk) {
    return row * BK + ((chunk ^ ((row >> 1) & 7)) << 2);   // float index
}

// ---- operand loaders: load(row, k) -> 4 consecutive k (k % 4 == 0), zeros outside ----

// Dense row-major matrix [rows][ld] with K valid columns.  ld % 4 == 0, base 16-B aligned.
// Optional strided-block row map (sampling): row r -> (r / G) * S + r % G, G = 1 << gshift.
struct DenseRows {
    const float* base;
    long long rows;      // number of valid *mapped* rows (actual row < rows)
    int ld;
    int K;
    int gshift;          // log2(G); blocks of G consecutive rows
    long long gstride;   // S: distance (rows) between consecutive blocks; == G for identity
    __device__ __forceinline__ long long map(long long r) const {
        return ((r >> gshift) * gstride) + (r & ((1ll << gshift) - 1));
    }
    // Branch-free staging: rows past the end are CLAMPED to the last row (their accumulator rows are
    // never stored or emitted: every epilogue guards on the row index), the K-range test is the same for
    // all rows of a K-step and is hoisted by the caller (k_valid).  Requires rows >= 1.
    // Per-row state: everything about a staged row that does not depend on k is computed ONCE before the
    // K loop (a thread stages the same rows in every K-step).
    using RowState = int;                        // clamped, mapped row (rows < 2^31: checked by the entry points)
    __device__ __forceinline__ RowState row_state(long long r) const {
        long long a = map(r);
        return (int)(a < rows ? a : rows - 1);
    }
    __device__ __forceinline__ bool k_valid(int k) const { return k < K; }
    __device__ __forceinline__ f32x4 load(RowState a, int k) const {
        return *reinterpret_cast<const f32x4*>(base + (long long)a * ld + k);
    }
};

// Concatenation of categorical-embedding rows and a numerical tail, gathered on the fly
// (EmbeddingLayer.forward two_tower_model.py:33-49; embed_features transformer_ranker.py:310-330):
//   k in [0, F*E)         -> table[off[f] + cat[row][f]][k % E],  f = k / E   (E % 4 == 0)
//   k in [F*E, F*E + n_num) -> num[row][k - F*E]
// cat is int64 [rows][F] (the reference casts .long()); a second categorical source lets the
// ranker read user features (F0 columns) and ad features (F - F0 columns) from two arrays.
struct EmbConcatRows {
    const float* tables;        // all feature tables back to back, [sum(card)][E]
    const int* off;             // [F] first row of feature f in `tables`
    const int* card;            // [F] cardinalities: indices are clamped into [0, card) so that a bad
                                //     index can never fault (it is reported by check_index_kernel)
    const long long* cat0;      // [.][F0]     row = (row_base + r) / cat0_rowdiv
    const long long* cat1;      // [.][F - F0] row = rowmap1 ? rowmap1[row_base + r] : row_base + r
    const long long* rowmap1;   // optional candidate id -> ad-feature-table row
    const float* num;           // [.][n_num] (same row as cat0) or nullptr
    long long row_base;         // global index of this chunk's first row
    long long rows;             // rows in this chunk
    long long rows1;            // rows of cat1 (rowmap1 values are clamped into [0, rows1))
    int F, F0, E, eshift;       // E == 1 << eshift
    int n_num;
    int cat0_rowdiv;            // > 1: one user row broadcast over cat0_rowdiv candidate rows
    // Row state: the user row (one division) and the ad-feature row (one rowmap load, clamped) are resolved
    // once per staged row, not once per K-step: what remains per K-step is index -> table row, two loads.
    struct RowState { int u, a; };
    __device__ __forceinline__ RowState row_state(long long r) const {
        r = r < rows ? r : rows - 1;              // clamped (see DenseRows)
        const long long gr = row_base + r;
        long long r1 = rowmap1 ? rowmap1[gr] : gr;
        r1 = r1 < 0 ? 0 : (r1 >= rows1 ? rows1 - 1 : r1);
        return RowState{(int)(gr / cat0_rowdiv), (int)r1};
    }
    __device__ __forceinline__ bool k_valid(int) const { return true; }   // the K tail is handled per segment
    __device__ __forceinline__ f32x4 load(RowState rs, int k) const {
        f32x4 z{0.f, 0.f, 0.f, 0.f};
        const int fe = F << eshift;
        if (k < fe) {
            const int f = k >> eshift;
            long long idx = (f < F0) ? cat0[(long long)rs.u * F0 + f] : cat1[(long long)rs.a * (F - F0) + (f - F0)];
            const long long hi = card[f] - 1;
            idx = idx < 0 ? 0 : (idx > hi ? hi : idx);
            return *reinterpret_cast<const f32x4*>(tables + (((long long)off[f] + idx) << eshift) +
                                                   (k & ((1 << eshift) - 1)));
        }
        const int j = k - fe;
        if (num == nullptr || j >= n_num) return z;
        const float* p = num + (long long)rs.u * n_num;
        z[0] = p[j];
        if (j + 1 < n_num) z[1] = p[j + 1];
        if (j + 2 < n_num) z[2] = p[j + 2];
        if (j + 3 < n_num) z[3] = p[j + 3];
        return z;
    }
};

// ---- block -> tile mapping -----------------------------------------------------------
// One operand is "small" (few tiles: query blocks, output-feature blocks), the other is
// streamed.  Blocks that share a streamed tile differ only in the small index; give them
// ids b, b+8, b+16, ... so that (observed round-robin placement, speed only) they land on
// one XCD and the streamed tile is fetched from HBM once and re-read from that XCD's L2.
struct TileMap {
    int tiles_small, tiles_big;
    __device__ __forceinline__ bool get(int bid, int& small, int& big) const {
        int per = 8 * tiles_small;
        int group = bid / per, within = bid - group * per;
        small = within >> 3;
        big = group * 8 + (within & 7);
        return big < tiles_big;
    }
    int grid() const { return ((tiles_big + 7) / 8) * 8 * tiles_small; }
};

// WP x WQ waves per workgroup, TP x TQ 32x32 tiles per wave.  DBUF = double-buffered LDS (one barrier per
// K-step, LDS 2x) or single-buffered (two barriers per K-step, half the LDS -> more workgroups per CU).
// The shapes used are 4 waves / single buffer (<= 48 KB LDS, <= 256 VGPRs): TWO independent workgroups
// per CU, so one workgroup's prologue (first HBM fetch) and epilogue (stores, residual loads, reductions)
// run under the other's MFMAs.  Measured on the 8-wave / 128 KB / one-per-CU predecessor: MFMA pipe busy
// 0.60-0.75, falling with epilogue weight (rocprofv3 SQ_VALU_MFMA_BUSY_CYCLES; profiles/README.md).
//
// BF16 = the operands are bf16 matrices handed to the loaders as float matrices of K/2 columns (bytes are
// bytes: the same 128-byte row segments, staging, swizzle and fragment reads); a K-step is then 64 bf16 and the
// 16 bytes a lane reads are the 8 consecutive k of one v_mfma_f32_32x32x16_bf16 operand (lane half h reads
// chunk 2c+h = k 16c+8h..+7, exactly the instruction's A/B lane map).  Same C/D layout, so every epilogue is
// shared.  Used by the search prefilter only (search.hip): 16x the fp32 MFMA rate, results re-scored in fp32.
template <int WP_, int WQ_, int TP_, int TQ_, bool DBUF_ = false, bool BF16_ = false>
struct Shape {
    static constexpr int WP = WP_, WQ = WQ_, TP = TP_, TQ = TQ_;
    static constexpr bool DBUF = DBUF_, BF16 = BF16_;
    static constexpr int NT = 64 * WP * WQ;
    static constexpr int BP = WP * TP * 32, BQ = WQ * TQ * 32;
    static constexpr int STAGE_FLOATS = (BP + BQ) * BK;
    static constexpr size_t LDS_BYTES = (DBUF ? 2ull : 1ull) * STAGE_FLOATS * sizeof(float);
    static constexpr int ROWS_PER_PASS = NT / 8;     // 8 x 16-byte chunks per 128-byte row segment
    static_assert(BP % ROWS_PER_PASS == 0, "P tile rows must fill whole staging passes");
};

// Accumulator tile set of one wave and where it sits in the output.
template <int TP_, int TQ_, int WP_, int WQ_>
struct Acc {
    static constexpr int TP = TP_, TQ = TQ_, WP = WP_, WQ = WQ_;
    static constexpr int BQ = WQ * TQ * 32;
    f32x16 v[TP][TQ];
    int p0, q0;   // global p / q index of the wave's first row / column
    int wp, wq;   // wave coordinates inside the workgroup
    // p index of register r in tile tp, q index of this lane in tile tq
    __device__ __forceinline__ int p(int tp, int r, int lane) const {
        return p0 + tp * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    }
    __device__ __forceinline__ int q(int tq, int lane) const { return q0 + tq * 32 + (lane & 31); }
};

// The mainloop for one workgroup: output tile with first P row `prow0` and first Q row `qrow0`.
template <class S, class LoadP, class LoadQ, class Epi>
__device__ __forceinline__ void gemm_block(const LoadP& lp, const LoadQ& lq, const Epi& epi, int ksteps,
                                           long long prow0, long long qrow0, float* smem) {
    constexpr int TP = S::TP, TQ = S::TQ, BP = S::BP, BQ = S::BQ;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wp = wave / S::WQ, wq = wave % S::WQ;

    // staging assignment: thread handles 16-B chunk (tid&7) of rows (tid>>3) + ROWS_PER_PASS*u
    const int srow = tid >> 3, schunk = tid & 7;
    constexpr int RPP = S::ROWS_PER_PASS;
    constexpr int NP = BP / RPP, NQ = (BQ + RPP - 1) / RPP;
    constexpr bool QPART = BQ % RPP != 0;            // Q tile shorter than one staging pass (small-M shapes)
    f32x4 rp[NP], rq[NQ];
    typename LoadP::RowState ps[NP];
    typename LoadQ::RowState qs[NQ];
#pragma unroll
    for (int u = 0; u < NP; ++u) ps[u] = lp.row_state(prow0 + srow + RPP * u);
#pragma unroll
    for (int u = 0; u < NQ; ++u) qs[u] = lq.row_state(qrow0 + ((!QPART || srow + RPP * u < BQ) ? srow + RPP * u : 0));

    auto stage_load = [&](int kt) {
        const int k = kt * BK + schunk * 4;
        const bool pv = lp.k_valid(k), qv = lq.k_valid(k);
#pragma unroll
        for (int u = 0; u < NP; ++u) rp[u] = pv ? lp.load(ps[u], k) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NQ; ++u)
            if (!QPART || srow + RPP * u < BQ)
                rq[u] = qv ? lq.load(qs[u], k) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto stage_store = [&](int buf) {
        float* sp = smem + buf * S::STAGE_FLOATS;
        float* sq = sp + BP * BK;
#pragma unroll
        for (int u = 0; u < NP; ++u) *reinterpret_cast<f32x4*>(sp + lds_slot(srow + RPP * u, schunk)) = rp[u];
#pragma unroll
        for (int u = 0; u < NQ; ++u)
            if (!QPART || srow + RPP * u < BQ) *reinterpret_cast<f32x4*>(sq + lds_slot(srow + RPP * u, schunk)) = rq[u];
    };

    Acc<TP, TQ, S::WP, S::WQ> acc;
    acc.p0 = (int)prow0 + wp * TP * 32;   // NOTE: int indices: streamed dims stay < 2^31 rows
    acc.q0 = (int)qrow0 + wq * TQ * 32;
    acc.wp = wp;
    acc.wq = wq;
#pragma unroll
    for (int i = 0; i < TP; ++i)
#pragma unroll
        for (int j = 0; j < TQ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc.v[i][j][r] = 0.f;

    stage_load(0);
    stage_store(0);
    __syncthreads();

    const int frow = lane & 31, fh = lane >> 5;
    for (int kt = 0; kt < ksteps; ++kt) {
        const bool more = kt + 1 < ksteps;
        const int buf = S::DBUF ? (kt & 1) : 0;
        const float* sp = smem + buf * S::STAGE_FLOATS + (wp * TP * 32) * BK;
        const float* sq = smem + buf * S::STAGE_FLOATS + BP * BK + (wq * TQ * 32) * BK;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (more && c == 0) stage_load(kt + 1);   // global -> registers, in flight under this K-step's MFMAs
            f32x4 a[TP], b[TQ];
#pragma unroll
            for (int i = 0; i < TP; ++i)
                a[i] = *reinterpret_cast<const f32x4*>(sp + lds_slot(i * 32 + frow, 2 * c + fh));
#pragma unroll
            for (int j = 0; j < TQ; ++j)
                b[j] = *reinterpret_cast<const f32x4*>(sq + lds_slot(j * 32 + frow, 2 * c + fh));
            if constexpr (S::BF16) {
#pragma unroll
                for (int i = 0; i < TP; ++i)
#pragma unroll
                    for (int j = 0; j < TQ; ++j)
                        acc.v[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc.v[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < TP; ++i)
#pragma unroll
                        for (int j = 0; j < TQ; ++j)
                            acc.v[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s],
                                                                               acc.v[i][j], 0, 0, 0);
            }
        }
        if (S::DBUF) {
            if (more) stage_store((kt + 1) & 1);
            __syncthreads();
        } else {
            __syncthreads();               // every wave has read this tile
            if (more) {
                stage_store(0);
                __syncthreads();
            }
        }
    }
    epi(acc, smem);
}

// Regular grid: P_IS_SMALL says which operand the TileMap's "small" index addresses.
template <class S, bool P_IS_SMALL, class LoadP, class LoadQ, class Epi>
__global__ __launch_bounds__(S::NT, 2) void gemm_nt_kernel(LoadP lp, LoadQ lq, Epi epi, int ksteps, TileMap tm) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int small, big;
    if (!tm.get(blockIdx.x, small, big)) return;
    const int tile_p = P_IS_SMALL ? small : big;
    const int tile_q = P_IS_SMALL ? big : small;
    gemm_block<S>(lp, lq, epi, ksteps, (long long)tile_p * S::BP, (long long)tile_q * S::BQ, smem);
}

// k_alg: the un-padded K (for the algorithmic FLOP/byte count of the profiling hook only)
template <class S, bool P_IS_SMALL, class LoadP, class LoadQ, class Epi>
inline hipError_t launch_gemm(const LoadP& lp, const LoadQ& lq, const Epi& epi, int K, long long p_rows,
                              long long q_rows, hipStream_t stream, int k_alg = 0) {
    auto kern = gemm_nt_kernel<S, P_IS_SMALL, LoadP, LoadQ, Epi>;
    // dynamic LDS = the larger of the staging tiles and what the epilogue carves out of the same array
    constexpr size_t lds_bytes = S::LDS_BYTES > Epi::lds_bytes(S::NT / 64) ? S::LDS_BYTES : Epi::lds_bytes(S::NT / 64);
    static_assert(lds_bytes <= 160 * 1024, "LDS budget");
    static PerDeviceOnce attr_done;   // per instantiation
    if (attr_done.pending()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        attr_done.mark();
    }
    TileMap tm;
    int tiles_p = (int)((p_rows + S::BP - 1) / S::BP), tiles_q = (int)((q_rows + S::BQ - 1) / S::BQ);
    tm.tiles_small = P_IS_SMALL ? tiles_p : tiles_q;
    tm.tiles_big = P_IS_SMALL ? tiles_q : tiles_p;
    if (tm.tiles_small <= 0 || tm.tiles_big <= 0) return hipSuccess;
    int ksteps = (K + BK - 1) / BK;
    char tag[64];
    if (g_prof_on) snprintf(tag, sizeof(tag), "%s_%dx%d%s", Epi::name, S::BP, S::BQ, S::BF16 ? "_bf16" : "");
    const double ka = k_alg > 0 ? k_alg : (S::BF16 ? 2.0 * K : (double)K);    // elements, not staged floats
    const double eb = S::BF16 ? 2.0 : 4.0;
    ProfScope prof(tag, 2.0 * (double)p_rows * (double)q_rows * ka,
                   eb * ((double)p_rows * ka + (double)q_rows * ka) +
                       4.0 * Epi::out_bytes_per_elem * (double)p_rows * (double)q_rows,
                   stream);
    hipLaunchKernelGGL(kern, dim3(tm.grid()), dim3(S::NT), lds_bytes, stream, lp, lq, epi, ksteps, tm);
    return hipGetLastError();
}

// ======================================================================================================
// Error-compensated fp32 GEMM on the bf16 MFMA ("x6"): the same C[p][q] = sum_k P[p][k] * Q[q][k] with fp32 inputs and
// fp32-level accuracy at up to 16/6 = 2.67x the fp32 MFMA rate.
//  * every fp32 value is split EXACTLY into three bf16 planes h + m + l (truncation split: 8 + 8 + 8 significand bits;
//    x - h and (x - h) - m are exact in fp32);
//  * the products hh, hm, mh, mm, hl, lh are accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (bf16 x bf16 is exact in
//    fp32); the dropped ml, lm, ll terms are <= 2^-23 of the product.  Measured against float64 (tools/x6_probe.hip,
//    K = 256 / 1024): max error 1.17e-6 / 4.3e-6 vs 1.15e-6 / 3.4e-6 for an fp32 fma chain - the same error level as the
//    fp32 MFMA path, which is why it may replace it inside the stated tolerances (tests/cases.py);
//  * P (the weights) is pre-split once on the host: [rows][K/16][3 planes][16] bf16, so the record of a row for one
//    K-step is 96 contiguous bytes; Q (the activations) stays fp32 in HBM and is split while it is staged (5.5 VALU per
//    element);
//  * same workgroup geometry as ShapeWide (256 x 128 tile, 4 waves, wave tile 128 x 64, two workgroups per CU), so every
//    epilogue is shared; a K-step is ONE v_mfma k16 step (48 MFMAs per wave) and everything is double-buffered in LDS
//    (72 KB per workgroup): the P planes arrive by LDS-DMA (no registers), the Q rows through 8 registers + split, both
//    for step k+1 while step k computes - ONE barrier per K-step, nothing between barriers;
//  * LDS image per operand [row][plane][2 chunks of 8 bf16] = 96 B per row (what the DMA writes linearly), chunk index
//    XOR (row>>3)&1 (source-side for P): a fragment read is base + compile-time offset, conflict-free ds_read_b128.
struct PlaneRows {
    const uint16_t* base;   // [rows][ksteps][3][16] bf16
    int rows;
    int ksteps;             // K / 16
};
struct ShapeX6 {
    static constexpr int WP = 2, WQ = 2, TP = 4, TQ = 2;
    static constexpr int NT = 256, BP = 256, BQ = 128, BK16 = 16;
    static constexpr int P_BYTES = BP * 96, Q_BYTES = BQ * 96;           // one K-step of one operand
    static constexpr size_t LDS_BYTES = 2 * (P_BYTES + Q_BYTES);
};
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// exact 3-way truncation split of 8 floats -> three packed bf16x8
__device__ __forceinline__ void x6_split8(const f32x4& a, const f32x4& b, u32x4& h, u32x4& m, u32x4& l) {
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    uint32_t hb[8], mb[8], lb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t u = __float_as_uint(x[i]);
        const float r1 = x[i] - __uint_as_float(u & 0xffff0000u);          // exact
        const uint32_t ru = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(ru & 0xffff0000u);           // exact, <= 8 significant bits
        hb[i] = u; mb[i] = ru; lb[i] = __float_as_uint(r2);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {                                          // high halves of two floats -> one dword
        h[i] = __builtin_amdgcn_perm(hb[2 * i + 1], hb[2 * i], 0x07060302u);
        m[i] = __builtin_amdgcn_perm(mb[2 * i + 1], mb[2 * i], 0x07060302u);
        l[i] = __builtin_amdgcn_perm(lb[2 * i + 1], lb[2 * i], 0x07060302u);
    }
}

// DBG (tools/x6v2_probe.hip only; 0 in the product, every test is `if constexpr`): 1 = no P DMA / Q loads after the
// first K-step, 2 = no LDS fragment reads after the first K-step, 4 = no split / Q plane stores after the first
// K-step, 8 = no barrier inside the K loop.
template <class LoadQ, class Epi, int DBG = 0>
__global__ __launch_bounds__(256, 2) void gemm_x6_kernel(PlaneRows lp, LoadQ lq, Epi epi, int ksteps, TileMap tm) {
    using S = ShapeX6;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned char* sP = reinterpret_cast<unsigned char*>(smem);           // [2][P_BYTES]
    unsigned char* sQ = sP + 2 * S::P_BYTES;                               // [2][Q_BYTES]
    int small, big;
    if (!tm.get(blockIdx.x, small, big)) return;
    const long long prow0 = (long long)small * S::BP, qrow0 = (long long)big * S::BQ;   // P (weights) is the small operand
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = w / S::WQ, wq = w % S::WQ;

    // P by LDS-DMA: the buffer is 1536 16-byte chunks in [row][plane][slot] order; pass u, thread t writes chunk
    // L = u * 256 + t (linear in the lane, as the DMA requires) and fetches source chunk slot ^ ((row >> 3) & 1)
    uint32_t pg[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        const int L = u * S::NT + tid, row = L / 6, within = L % 6;
        long long r = prow0 + row;
        r = r < lp.rows ? r : lp.rows - 1;                                  // clamped (epilogues guard on the feature index)
        pg[u] = (uint32_t)((r * lp.ksteps) * 96 + (within >> 1) * 32 + (((within & 1) ^ ((row >> 3) & 1)) * 16));
    }
    auto dma_p = [&](int kt, int buf) {
        const unsigned char* gb = reinterpret_cast<const unsigned char*>(lp.base) + (size_t)kt * 96;
        unsigned char* lb = sP + buf * S::P_BYTES + (w * 64) * 16;          // wave-uniform; the hardware adds lane * 16
#pragma unroll
        for (int u = 0; u < 6; ++u)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + pg[u]),
                                             (__attribute__((address_space(3))) void*)(lb + u * S::NT * 16), 16, 0, 0);
    };
    // Q through registers: thread t stages the 8 floats [c * 8, c * 8 + 8) of row t / 2 (c = t % 2)
    const int qrow = tid >> 1, qc = tid & 1;
    const typename LoadQ::RowState qs = lq.row_state(qrow0 + qrow);
    const int qoff = qrow * 96 + ((qc ^ ((qrow >> 3) & 1)) * 16);
    f32x4 rq[2];
    auto load_q = [&](int kt) {
        const int k = kt * S::BK16 + qc * 8;
        rq[0] = lq.k_valid(k) ? lq.load(qs, k) : f32x4{0.f, 0.f, 0.f, 0.f};
        rq[1] = lq.k_valid(k + 4) ? lq.load(qs, k + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto store_q = [&](int buf) {
        unsigned char* dst = sQ + buf * S::Q_BYTES + qoff;
        u32x4 h, m, l;
        x6_split8(rq[0], rq[1], h, m, l);
        *reinterpret_cast<u32x4*>(dst) = h;
        *reinterpret_cast<u32x4*>(dst + 32) = m;
        *reinterpret_cast<u32x4*>(dst + 64) = l;
    };

    Acc<S::TP, S::TQ, S::WP, S::WQ> acc;
    acc.p0 = (int)prow0 + wp * S::TP * 32;
    acc.q0 = (int)qrow0 + wq * S::TQ * 32;
    acc.wp = wp;
    acc.wq = wq;
#pragma unroll
    for (int i = 0; i < S::TP; ++i)
#pragma unroll
        for (int j = 0; j < S::TQ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc.v[i][j][r] = 0.f;

    dma_p(0, 0);
    load_q(0);
    store_q(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int frow = lane & 31, fh = lane >> 5;
    const int fbase = frow * 96 + ((fh ^ ((frow >> 3) & 1)) * 16);         // lane part of every fragment address
    for (int kt = 0; kt < ksteps; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < ksteps;
        if (more && !(DBG & 1)) {
            dma_p(kt + 1, buf ^ 1);
            load_q(kt + 1);
        }
        const unsigned char* pa_base = sP + ((DBG & 1) ? 0 : buf) * S::P_BYTES + (wp * 128) * 96 + fbase;
        const unsigned char* pb_base = sQ + ((DBG & 1) ? 0 : buf) * S::Q_BYTES + (wq * 64) * 96 + fbase;
        bf16x8 b[S::TQ][3];
        bf16x8 a3[3][S::TP];
        if (!(DBG & 2) || kt == 0) {
#pragma unroll
            for (int j = 0; j < S::TQ; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p) b[j][p] = *reinterpret_cast<const bf16x8*>(pb_base + j * 32 * 96 + p * 32);
        }
        // P plane h with Q planes l, m, h; P plane m with Q planes m, h; P plane l with Q plane h
#pragma unroll
        for (int pa = 0; pa < 3; ++pa) {
            bf16x8 (&a)[S::TP] = a3[(DBG & 2) ? pa : 0];
            if (!(DBG & 2) || kt == 0) {
#pragma unroll
                for (int i = 0; i < S::TP; ++i) a[i] = *reinterpret_cast<const bf16x8*>(pa_base + i * 32 * 96 + pa * 32);
            }
#pragma unroll
            for (int pb = 2 - pa; pb >= 0; --pb)
#pragma unroll
                for (int i = 0; i < S::TP; ++i)
#pragma unroll
                    for (int j = 0; j < S::TQ; ++j)
                        acc.v[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j][pb], acc.v[i][j], 0, 0, 0);
        }
        if (more && !(DBG & 4)) store_q(buf ^ 1);
        if (!(DBG & 8)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // my share of the next P planes has landed
            __syncthreads();               // step kt consumed by every wave; step kt+1's planes visible
        }
    }
    if (DBG & 8) __syncthreads();
    epi(acc, smem);
}

// P = pre-split weights (the small operand), Q = fp32 rows.  K = padded K of the weights (multiple of 32).
template <class LoadQ, class Epi>
inline hipError_t launch_gemm_x6(const uint16_t* planes, int p_rows, const LoadQ& lq, const Epi& epi, int K,
                                 long long q_rows, hipStream_t stream, int k_alg = 0) {
    using S = ShapeX6;
    auto kern = gemm_x6_kernel<LoadQ, Epi>;
    constexpr size_t lds_bytes = S::LDS_BYTES > Epi::lds_bytes(S::NT / 64) ? S::LDS_BYTES : Epi::lds_bytes(S::NT / 64);
    static_assert(lds_bytes <= 160 * 1024, "LDS budget");
    static PerDeviceOnce attr_done;   // per instantiation
    if (attr_done.pending()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
        attr_done.mark();
    }
    TileMap tm;
    tm.tiles_small = (p_rows + S::BP - 1) / S::BP;
    tm.tiles_big = (int)((q_rows + S::BQ - 1) / S::BQ);
    if (tm.tiles_small <= 0 || tm.tiles_big <= 0) return hipSuccess;
    const int ksteps = K / S::BK16;
    PlaneRows lp{planes, p_rows, ksteps};
    char tag[64];
    if (g_prof_on) snprintf(tag, sizeof(tag), "%s_%dx%d_x6", Epi::name, S::BP, S::BQ);
    const double ka = k_alg > 0 ? k_alg : K;
    ProfScope prof(tag, 2.0 * (double)p_rows * (double)q_rows * ka,
                   4.0 * ((double)p_rows * ka + (double)q_rows * ka) +
                       4.0 * Epi::out_bytes_per_elem * (double)p_rows * (double)q_rows,
                   stream);
    hipLaunchKernelGGL(kern, dim3(tm.grid()), dim3(S::NT), lds_bytes, stream, lp, lq, epi, ksteps, tm);
    return hipGetLastError();
}

}  // namespace amdrec
