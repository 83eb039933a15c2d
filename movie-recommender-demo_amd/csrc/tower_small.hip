// Small batches of the two-tower forward (two_tower_model.py:98-121 / :167-184, eval mode, BatchNorm folded) in ONE launch.
//
// amdrec_tower_forward's general path is one tiled-GEMM launch per layer; for the serving path's batches (one request,
// or 512 users per step) each of those launches is latency-bound - a UserTower took 12 + 23 + 14 us in three kernels of
// 16-32 workgroups each (kernel trace, profiles/r03_trace_b1.log).  Here a workgroup of 8 waves owns 16 rows through the
// WHOLE tower:
//   * the gathered input row (embedding lookups of two_tower_model.py:33-49 + the numerical features, zero-padded to a
//     multiple of 128) and every layer's activations live in LDS (two ping-pong buffers of 16 x width floats);
//   * a layer is width / 16 output tiles of 16 features, dealt in pairs to the waves; a tile is K / 4
//     v_mfma_f32_16x16x4_f32 (exact fp32 products, fp32 accumulate - the arithmetic of the general path's fp32 MFMA);
//     A = weights straight from global memory (each element is used once per workgroup: a lane loads 16 B of its feature's
//     row = 4 k values = 4 MFMAs; the k index a lane group supplies is permuted the same way for both operands), B = the
//     activations from LDS (one ds_read_b128 per 4 MFMAs, shared by the pair of tiles);
//   * what bounds a tile is the latency of its weight loads, so they are issued in straight-line chunks of up to 16
//     k-steps (32 loads of 16 B per lane in flight) in front of the chunk's MFMAs, which then start as the data arrives
//     (counted vmcnt waits).  Chunks are compile-time sized: a conditional inside the pipeline makes the compiler wait
//     for every outstanding load at the join (two earlier versions with run-time step predicates: 74 and 54 us);
//   * bias + ReLU go back to LDS; the last layer's rows are L2-normalised (x / max(||x||, 1e-12), F.normalize) and written
//     with 16-byte stores.
// Same inputs and layout contract as the general path (weights [out][ldw], K zero-padded to a multiple of 32); indices are
// clamped into their tables like EmbConcatRows does (a bad index is reported by check_index_kernel, never dereferenced).
#include "gemm_core.hpp"
#include "../../include/amdrec.h"

namespace amdrec {

constexpr int TS_ROWS = 16, TS_WAVES = 8;
constexpr long long TS_MAX_ROWS = 4096;      // one 16-row workgroup per CU; beyond it the tiled GEMMs have enough work per launch
constexpr int TS_MAX_WIDTH = 1024;
constexpr long long TS_GEMV_MAX_ROWS = 1024; // rows that take the vector-ALU GEMV kernel (reference tower shapes): 1 / 2 / 4 rows per workgroup, one workgroup per CU

struct TowerSmallArgs {
    EmbConcatRows in;
    int n_layers;
    int dims[AMDREC_MAX_LAYERS + 1];
    int kp[AMDREC_MAX_LAYERS];                 // K of layer l as the kernel walks it: dims[l] rounded up (layer 0: to 128)
    int ldw[AMDREC_MAX_LAYERS];
    const float* w[AMDREC_MAX_LAYERS];
    const float* b[AMDREC_MAX_LAYERS];
    float* out;
    long long ld_out;
    long long rows;
    int ld_act;                                // floats between the rows of an activation buffer
    int renorm;                                // amdrec_tower_params.renormalize
};

// LDS pointers keep their address space: through a generic pointer (e.g. an array of two buffer pointers indexed at run time)
// every activation read becomes a FLAT load, which the compiler must wait for with vmcnt(0) - draining the weight loads in
// flight at every k-step.
typedef __attribute__((address_space(3))) float lds_f32;

// STEPS k-steps (16 k each) of a pair of output tiles: every weight load first, then the MFMAs, each step behind a COUNTED
// wait for its own two loads.  Loads and waits are inline asm: left to the compiler, the first MFMA waited for ALL loads of the
// chunk (s_waitcnt vmcnt(0)) and the chunk's memory latency was never overlapped with its own arithmetic.  The chunk is the
// only vector-memory traffic of a wave at this point (the input gather ended at a barrier), loads return in issue order, and
// each wait names the registers it releases, so no MFMA can be scheduled ahead of it.
template <int STEPS>
__device__ __forceinline__ void tower_chunk(const float* w0, const float* w1, const lds_f32* src, f32x4& acc0, f32x4& acc1) {
    f32x4 a0[STEPS], a1[STEPS];
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(a0[i]) : "v"(w0), "n"(64 * i));
        asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(a1[i]) : "v"(w1), "n"(64 * i));
    }
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        const f32x4 bv = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(src + 16 * i);
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a0[i]), "+v"(a1[i]) : "n"(2 * (STEPS - 1 - i)));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[i][e], bv[e], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i][e], bv[e], acc1, 0, 0, 0);
        }
    }
}

// L2 normalisation of the last layer's rows (x / max(||x||, 1e-12), F.normalize) and the store: wave w takes rows 2 w, 2 w + 1,
// a lane 16 B at a time
__device__ __forceinline__ void normalize_rows_out(const lds_f32* fin, int LD, int nout, long long row0, const TowerSmallArgs& a,
                                                   int wave, int lane) {
#pragma unroll 1
    for (int rr = 0; rr < TS_ROWS / TS_WAVES; ++rr) {
        const int r = (TS_ROWS / TS_WAVES) * wave + rr;
        const lds_f32* xr = fin + r * LD;
        float ss = 0.f;
        for (int c = 4 * lane; c < nout; c += 256) {
            const f32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(xr + c);
            ss += sumsq4(v);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        const float nrm = fmaxf(sqrtf(ss), 1e-12f);
        float inv = 1.0f;
        if (a.renorm) {                                      // amdrec_l2_normalize of the normalised row: l2_normalize_kernel's
            float s2 = 0.f;                                  // lane map and expressions (csrc/rows.hip), so the same bits
            for (int c = 4 * lane; c < nout; c += 256) {
                f32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(xr + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] / nrm;
                s2 += sumsq4(v);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
            inv = s2 > 0.f ? 1.0f / sqrtf(s2) : 1.0f;
        }
        if (row0 + r < a.rows) {
            float* o = a.out + (row0 + r) * a.ld_out;
            for (int c = 4 * lane; c < nout; c += 256) {
                f32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(xr + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] / nrm;
                if (a.renorm) v = f32x4{v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv};
                *reinterpret_cast<f32x4*>(o + c) = v;
            }
        }
    }
}

// one row, one wave (the GEMV kernel): normalize_rows_out's arithmetic
__device__ __forceinline__ void normalize_rows_out_single(const lds_f32* xr, int nout, const TowerSmallArgs& a, int lane) {
    float ss = 0.f;
    for (int c = 4 * lane; c < nout; c += 256) {
        const f32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(xr + c);
        ss += sumsq4(v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float nrm = fmaxf(sqrtf(ss), 1e-12f);
    float inv = 1.0f;
    if (a.renorm) {
        float s2 = 0.f;
        for (int c = 4 * lane; c < nout; c += 256) {
            f32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(xr + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] / nrm;
            s2 += sumsq4(v);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
        inv = s2 > 0.f ? 1.0f / sqrtf(s2) : 1.0f;
    }
    for (int c = 4 * lane; c < nout; c += 256) {
        f32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(xr + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] / nrm;
        if (a.renorm) v = f32x4{v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv};
        *reinterpret_cast<f32x4*>(a.out + c) = v;
    }
}

__global__ __launch_bounds__(64 * TS_WAVES) void tower_small_kernel(TowerSmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) float act[];         // [2][TS_ROWS][ld_act]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const long long row0 = (long long)blockIdx.x * TS_ROWS;
    const int LD = a.ld_act;
    lds_f32* const act3 = (lds_f32*)act;
    auto buf = [&](int i) -> lds_f32* { return act3 + (i & 1) * (TS_ROWS * LD); };
    // input rows -> buf[0][row][0 .. kp[0]), zero beyond dims[0] (EmbConcatRows::load returns zeros there)
    {
        const int chunks = a.kp[0] >> 2;
        for (int i = tid; i < TS_ROWS * chunks; i += 64 * TS_WAVES) {
            const int r = i / chunks, k = (i - r * chunks) << 2;
            const f32x4 v = a.in.load(a.in.row_state(row0 + r), k);           // rows past the end are clamped, never written out
            *reinterpret_cast<__attribute__((address_space(3))) f32x4*>(buf(0) + r * LD + k) = v;
        }
    }
    __syncthreads();
    for (int l = 0; l < a.n_layers; ++l) {
        const float* __restrict__ W = a.w[l];
        const float* __restrict__ bias = a.b[l];
        const int ldw = a.ldw[l], nout = a.dims[l + 1];
        const int steps = a.kp[l] >> 4;                                      // kp <= ldw, weights and activations zero beyond K
        const bool last = l == a.n_layers - 1;
        const lds_f32* src0 = buf(l) + n * LD + 4 * g;
        lds_f32* dst = buf(l + 1);
        const int tiles = nout >> 4;
        for (int t = 2 * wave; t < tiles; t += 2 * TS_WAVES) {               // a pair of output tiles per turn (wave-uniform)
            const bool two = t + 1 < tiles;
            const float* w0 = W + (long long)(16 * t + n) * ldw + 4 * g;    // A: lane (i = n) is output feature 16 t + n
            const float* w1 = two ? w0 + 16ll * ldw : w0;
            const lds_f32* src = src0;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            int rem = steps;                                                  // wave-uniform walk in compile-time chunks
            while (rem >= 16) { tower_chunk<16>(w0, w1, src, acc0, acc1); w0 += 256; w1 += 256; src += 256; rem -= 16; }
            if (rem >= 8) { tower_chunk<8>(w0, w1, src, acc0, acc1); w0 += 128; w1 += 128; src += 128; rem -= 8; }
            if (rem >= 4) { tower_chunk<4>(w0, w1, src, acc0, acc1); w0 += 64; w1 += 64; src += 64; rem -= 4; }
            if (rem >= 2) { tower_chunk<2>(w0, w1, src, acc0, acc1); w0 += 32; w1 += 32; src += 32; rem -= 2; }
            if (rem >= 1) tower_chunk<1>(w0, w1, src, acc0, acc1);
            // accumulator: lane (n, g) holds features 16 t + 4 g + {0..3} of row n
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + 16 * t + 4 * g);
            f32x4 y0;
#pragma unroll
            for (int e = 0; e < 4; ++e) y0[e] = last ? acc0[e] + b0[e] : fmaxf(acc0[e] + b0[e], 0.f);
            *reinterpret_cast<__attribute__((address_space(3))) f32x4*>(dst + n * LD + 16 * t + 4 * g) = y0;
            if (two) {
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(bias + 16 * (t + 1) + 4 * g);
                f32x4 y1;
#pragma unroll
                for (int e = 0; e < 4; ++e) y1[e] = last ? acc1[e] + b1[e] : fmaxf(acc1[e] + b1[e], 0.f);
                *reinterpret_cast<__attribute__((address_space(3))) f32x4*>(dst + n * LD + 16 * (t + 1) + 4 * g) = y1;
            }
        }
        __syncthreads();
    }
    normalize_rows_out(buf(a.n_layers), LD, a.dims[a.n_layers], row0, a, wave, lane);
}

// ---- the same tower as ONE software pipeline, for three-layer towers whose widths are multiples of 256 / 128 ------------------
// In the kernel above every (layer, tile pair, chunk) starts with its own round of weight loads: five exposed load latencies
// for the reference's user tower (109 -> 512 -> 256 -> 256) - 41 us for 1 MB of weights, 25 GB/s.  The weights do not depend
// on the activations, so here the whole tower is a compile-time list of UNITS (one tile pair x 128 k = 16 loads of 16 B per
// lane) and unit u + 1's loads go out before unit u's MFMAs, across turn and layer boundaries (two register buffers of 64);
// the biases are loaded once, ahead of everything.  Loads, counted waits and the registers they release are inline asm as in
// tower_chunk; between the input gather and the final store the compiler issues no vector-memory instruction of its own, so
// the counts hold.  K0 = layer 0's K padded to 128, D1..D3 = the layer widths (pairs of tiles per layer: a multiple of 8).
template <int K0, int D1, int D2, int D3>
struct TowerPipe {
    static constexpr int T0 = D1 / 256, T1 = D2 / 256, T2 = D3 / 256;          // turns per wave (pairs / 8)
    static constexpr int C0 = K0 / 128, C1 = D1 / 128, C2 = D2 / 128;          // 128-k chunks per turn
    static constexpr int U0 = T0 * C0, U1 = T1 * C1, U2 = T2 * C2, NU = U0 + U1 + U2;
    static constexpr int NB = T0 + T1 + T2;                                    // bias pairs per wave
    static constexpr int layer(int u) { return u < U0 ? 0 : (u < U0 + U1 ? 1 : 2); }
    static constexpr int local(int u) { return u < U0 ? u : (u < U0 + U1 ? u - U0 : u - U0 - U1); }
    static constexpr int chunks(int l) { return l == 0 ? C0 : (l == 1 ? C1 : C2); }
    static constexpr int turn(int u) { return local(u) / chunks(layer(u)); }
    static constexpr int chunk(int u) { return local(u) % chunks(layer(u)); }
    static constexpr int bias_slot(int u) { return (layer(u) == 0 ? 0 : (layer(u) == 1 ? T0 : T0 + T1)) + turn(u); }

    const TowerSmallArgs& a;
    lds_f32* act3;
    int LD, wave, n, g;
    f32x4 A0[2][8], A1[2][8];
    f32x4 B0[NB], B1[NB];
    f32x4 acc0, acc1;

    __device__ __forceinline__ lds_f32* buf(int i) const { return act3 + (i & 1) * (TS_ROWS * LD); }
    template <int U>
    __device__ __forceinline__ int tile() const { return 2 * (wave + TS_WAVES * turn(U)); }

    template <int U>
    __device__ __forceinline__ void issue() {
        constexpr int L = layer(U), P = U & 1;
        const float* w0 = a.w[L] + (long long)(16 * tile<U>() + n) * a.ldw[L] + 4 * g + 128 * chunk(U);
        const float* w1 = w0 + 16ll * a.ldw[L];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(A0[P][i]) : "v"(w0), "n"(64 * i));
            asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(A1[P][i]) : "v"(w1), "n"(64 * i));
        }
    }
    template <int S>
    __device__ __forceinline__ void issue_bias() {
        if constexpr (S < NB) {
            constexpr int L = S < T0 ? 0 : (S < T0 + T1 ? 1 : 2);
            constexpr int T = S - (L == 0 ? 0 : (L == 1 ? T0 : T0 + T1));
            const float* b = a.b[L] + 16 * (2 * (wave + TS_WAVES * T)) + 4 * g;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(B0[S]) : "v"(b));
            asm volatile("global_load_dwordx4 %0, %1, off offset:64" : "=v"(B1[S]) : "v"(b));
            issue_bias<S + 1>();
        }
    }
    template <int S>
    __device__ __forceinline__ void bias_landed() {                            // behind the first counted wait (loads return in order)
        if constexpr (S < NB) {
            asm volatile("" : "+v"(B0[S]), "+v"(B1[S]));
            bias_landed<S + 1>();
        }
    }
    template <int U>
    __device__ __forceinline__ void unit() {
        constexpr int L = layer(U), P = U & 1, C = chunk(U);
        constexpr bool more = U + 1 < NU;
        if constexpr (more) issue<U + 1>();
        if constexpr (C == 0 && turn(U) == 0) {                                // the layer's input (gather / previous layer) is complete;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // not __syncthreads(): its vmcnt(0) would drain the
            __builtin_amdgcn_s_barrier();                                      // weight loads in flight
        }
        if constexpr (C == 0) { acc0 = f32x4{0.f, 0.f, 0.f, 0.f}; acc1 = acc0; }
        const lds_f32* src = buf(L) + n * LD + 4 * g + 128 * C;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 bv = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(src + 16 * i);
            asm volatile("s_waitcnt vmcnt(%2)" : "+v"(A0[P][i]), "+v"(A1[P][i]) : "n"(2 * (7 - i) + (more ? 16 : 0)));
            if constexpr (U == 0) { if (i == 0) bias_landed<0>(); }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[P][i][e], bv[e], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[P][i][e], bv[e], acc1, 0, 0, 0);
            }
        }
        if constexpr (C == chunks(L) - 1) {                                    // the pair is complete: bias (+ ReLU) -> LDS
            constexpr int S = bias_slot(U);
            constexpr bool last = L == 2;
            lds_f32* dst = buf(L + 1) + n * LD + 16 * tile<U>() + 4 * g;
            f32x4 y0, y1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                y0[e] = last ? acc0[e] + B0[S][e] : fmaxf(acc0[e] + B0[S][e], 0.f);
                y1[e] = last ? acc1[e] + B1[S][e] : fmaxf(acc1[e] + B1[S][e], 0.f);
            }
            *reinterpret_cast<__attribute__((address_space(3))) f32x4*>(dst) = y0;
            *reinterpret_cast<__attribute__((address_space(3))) f32x4*>(dst + 16) = y1;
        }
        if constexpr (more) unit<U + 1>();
    }
};

template <int K0, int D1, int D2, int D3>
__global__ __launch_bounds__(64 * TS_WAVES) void tower_pipe_kernel(TowerSmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) float act[];         // [2][TS_ROWS][ld_act]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long row0 = (long long)blockIdx.x * TS_ROWS;
    TowerPipe<K0, D1, D2, D3> P{a, (lds_f32*)act, a.ld_act, wave, lane & 15, lane >> 4};
    P.template issue_bias<0>();
    P.template issue<0>();
    {   // input rows -> buf[0][row][0 .. K0), zero beyond dims[0]; its loads are the compiler's own, behind the ones above
        constexpr int chunks = K0 >> 2;
        for (int i = tid; i < TS_ROWS * chunks; i += 64 * TS_WAVES) {
            const int r = i / chunks, k = (i - r * chunks) << 2;
            const f32x4 v = a.in.load(a.in.row_state(row0 + r), k);
            *reinterpret_cast<__attribute__((address_space(3))) f32x4*>(P.buf(0) + r * P.LD + k) = v;
        }
    }
    P.template unit<0>();
    __syncthreads();
    normalize_rows_out(P.buf(3), P.LD, D3, row0, a, wave, lane);
}

// ---- one or two rows: a vector-ALU GEMV ---------------------------------------------------------------------------------
// A single request's user row on the kernels above is a 16-row MFMA tile with 15 rows of padding: 8.4 MFLOP of fp32 MFMA on ONE
// CU = 16 us of the 36.  Here a 512-thread workgroup per row computes 64 output features per pass, eight lanes per feature
// (each reads 16 contiguous bytes of the feature's weight row per 32-k step, fp32 FMA chains, a three-step lane reduction);
// the passes of all three layers are one compile-time list and pass p + 1's weight loads (they do not depend on the
// activations) are issued before pass p's arithmetic, across the layer barriers (LDS-only fences: a __syncthreads() would
// order - and wait for - the global loads too).  Bound: the row's 1 MB of weights through one CU's vector-memory pipeline.
template <int K0, int D1, int D2, int D3, int R>
struct TowerGemv {
    static constexpr int P0 = D1 / 64, P1 = D2 / 64, P2 = D3 / 64, NP = P0 + P1 + P2;
    static constexpr int layer(int p) { return p < P0 ? 0 : (p < P0 + P1 ? 1 : 2); }
    static constexpr int local(int p) { return p < P0 ? p : (p < P0 + P1 ? p - P0 : p - P0 - P1); }
    static constexpr int kdim(int l) { return l == 0 ? K0 : (l == 1 ? D1 : D2); }
    static constexpr int MAXS = (K0 > D1 ? (K0 > D2 ? K0 : D2) : (D1 > D2 ? D1 : D2)) / 32;
    const TowerSmallArgs& a;
    lds_f32* act;                       // [R][2][1024]: the workgroup's R rows, two ping-pong buffers each
    int n8, j;                          // feature within a pass (tid >> 3), lane of its group of eight
    f32x4 W[2][MAXS];
    float Bv[2];

    template <int P>
    __device__ __forceinline__ void load() {
        constexpr int L = layer(P), S = kdim(L) / 32;
        const int n = 64 * local(P) + n8;
        const float* w = a.w[L] + (long long)n * a.ldw[L] + 4 * j;
#pragma unroll
        for (int i = 0; i < S; ++i) W[P & 1][i] = *reinterpret_cast<const f32x4*>(w + 32 * i);
        Bv[P & 1] = a.b[L][n];
    }
    template <int P>
    __device__ __forceinline__ void run() {
        constexpr int L = layer(P), S = kdim(L) / 32;
        if constexpr (P + 1 < NP) load<P + 1>();
        if constexpr (local(P) == 0) {                          // the layer's input is complete (LDS traffic only)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {                           // the pass's weights, once loaded, serve every row of the workgroup
            const lds_f32* x = act + r * 2048 + (L & 1) * 1024 + 4 * j;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int i = 0; i < S; ++i) {
                const f32x4 xv = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(x + 32 * i);
                a0 = __builtin_fmaf(W[P & 1][i][0], xv[0], a0);
                a1 = __builtin_fmaf(W[P & 1][i][1], xv[1], a1);
                a2 = __builtin_fmaf(W[P & 1][i][2], xv[2], a2);
                a3 = __builtin_fmaf(W[P & 1][i][3], xv[3], a3);
            }
            float v = (a0 + a1) + (a2 + a3);
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 4, 64);
            v += Bv[P & 1];
            if (j == 0) act[r * 2048 + ((L + 1) & 1) * 1024 + 64 * local(P) + n8] = L == 2 ? v : fmaxf(v, 0.f);
        }
        if constexpr (P + 1 < NP) run<P + 1>();
    }
};

// R rows per workgroup: one row per workgroup up to a chip-full of rows (256), then two and four - the weights are read once
// per workgroup whatever R is, and the arithmetic is nothing
template <int K0, int D1, int D2, int D3, int R>
__global__ __launch_bounds__(512) void tower_gemv_kernel(TowerSmallArgs a) {
    __shared__ __attribute__((aligned(16))) float act[R * 2 * 1024];
    const int tid = threadIdx.x;
    const long long row0 = (long long)blockIdx.x * R;
    TowerGemv<K0, D1, D2, D3, R> T{a, (lds_f32*)act, tid >> 3, tid & 7};
    T.template load<0>();
    for (int i = tid; i < R * (K0 / 4); i += 512) {              // the input rows (zero beyond dims[0]; past the end: clamped)
        const int r = i / (K0 / 4), c = i - r * (K0 / 4);
        const f32x4 v = a.in.load(a.in.row_state(row0 + r), 4 * c);
        *reinterpret_cast<__attribute__((address_space(3))) f32x4*>(T.act + r * 2048 + 4 * c) = v;
    }
    T.template run<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // F.normalize: row r's D3 outputs are in its buffer 1 (three layers); wave r, the lane map of the kernels above
    const int w = tid >> 6;
    if (w < R && row0 + w < a.rows) {
        TowerSmallArgs one = a;
        one.rows = 1;
        one.out = a.out + (row0 + w) * a.ld_out;
        normalize_rows_out_single(T.act + w * 2048 + 1024, D3, one, tid & 63);
    }
}

// eligibility: every hidden / output width a multiple of 16 (whole output tiles, and the next layer's K needs no padding)
bool tower_small_ok(const amdrec_tower_params* p, long long rows) {
    if (rows > TS_MAX_ROWS) return false;
    if (p->dims[0] > TS_MAX_WIDTH || ((p->dims[0] + 127) & ~127) > p->ldw[0]) return false;   // layer 0 is walked in 128-k units
    for (int l = 1; l <= p->n_layers; ++l)
        if (p->dims[l] % 16 != 0 || p->dims[l] > TS_MAX_WIDTH) return false;
    return true;
}

hipError_t tower_small_run(const amdrec_tower_params* p, const long long* cat, const float* num, long long rows, float* out,
                           long long ld_out, hipStream_t st) {
    TowerSmallArgs a{};
    a.in.tables = p->tables; a.in.off = p->table_off; a.in.card = p->cards;
    a.in.cat0 = cat; a.in.cat1 = nullptr; a.in.rowmap1 = nullptr; a.in.num = num;
    a.in.row_base = 0; a.in.rows = rows; a.in.rows1 = rows;
    a.in.F = p->n_feat; a.in.F0 = p->n_feat; a.in.E = p->emb_dim;
    a.in.eshift = 31 - __builtin_clz((unsigned)p->emb_dim);                  // emb_dim is a power of two (tower_check)
    a.in.n_num = p->n_num; a.in.cat0_rowdiv = 1;
    a.n_layers = p->n_layers;
    int wmax = (p->dims[0] + 127) & ~127;
    double flops = 0;
    for (int l = 0; l <= p->n_layers; ++l) {
        a.dims[l] = p->dims[l];
        if (l >= 1 && p->dims[l] > wmax) wmax = p->dims[l];
        if (l < p->n_layers) {
            a.kp[l] = l == 0 ? (p->dims[0] + 127) & ~127 : p->dims[l];
            a.ldw[l] = p->ldw[l]; a.w[l] = p->w[l]; a.b[l] = p->b[l];
            flops += 2.0 * p->dims[l] * p->dims[l + 1];
        }
    }
    a.out = out; a.ld_out = ld_out; a.rows = rows; a.renorm = p->renormalize;
    a.ld_act = wmax + 4;                                  // + 16 bytes: rows of a buffer start on different banks
    const size_t lds = 2ull * TS_ROWS * a.ld_act * sizeof(float);
    static PerDeviceOnce attr_done;
    if (attr_done.pending()) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tower_small_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TS_ROWS * (TS_MAX_WIDTH + 4) * 4);
        if (e != hipSuccess) return e;
        attr_done.mark();
    }
    const dim3 grid((unsigned)((rows + TS_ROWS - 1) / TS_ROWS)), block(64 * TS_WAVES);
    // the reference's towers (user: 109 -> 512 -> 256 -> 256, ad: 320 -> 512 -> 256 -> 256) as one software pipeline
    if (p->n_layers == 3 && p->dims[1] == 512 && p->dims[2] == 256 && p->dims[3] == 256 && (a.kp[0] == 128 || a.kp[0] == 384)) {
        static PerDeviceOnce attr_pipe;
        if (attr_pipe.pending()) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tower_pipe_kernel<128, 512, 256, 256>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TS_ROWS * (TS_MAX_WIDTH + 4) * 4);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(tower_pipe_kernel<384, 512, 256, 256>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TS_ROWS * (TS_MAX_WIDTH + 4) * 4);
            if (e != hipSuccess) return e;
            attr_pipe.mark();
        }
        if (rows <= TS_GEMV_MAX_ROWS) {                  // up to four rows per workgroup and one workgroup per CU: no MFMA tile to fill
            ProfScope prof("tower_gemv_1row", flops * (double)rows, (double)rows * 4.0 * (p->dims[0] + p->dims[p->n_layers]), st);
            const int R = rows <= 256 ? 1 : (rows <= 512 ? 2 : 4);
            const dim3 g((unsigned)((rows + R - 1) / R));
#define AMDREC_GEMV(K0_, R_) hipLaunchKernelGGL((tower_gemv_kernel<K0_, 512, 256, 256, R_>), g, dim3(512), 0, st, a)
            if (a.kp[0] == 128) { if (R == 1) AMDREC_GEMV(128, 1); else if (R == 2) AMDREC_GEMV(128, 2); else AMDREC_GEMV(128, 4); }
            else                { if (R == 1) AMDREC_GEMV(384, 1); else if (R == 2) AMDREC_GEMV(384, 2); else AMDREC_GEMV(384, 4); }
#undef AMDREC_GEMV
            return hipGetLastError();
        }
        ProfScope prof("tower_pipe_16rows", flops * (double)rows, (double)rows * 4.0 * (p->dims[0] + p->dims[p->n_layers]), st);
        if (a.kp[0] == 128) hipLaunchKernelGGL((tower_pipe_kernel<128, 512, 256, 256>), grid, block, lds, st, a);
        else hipLaunchKernelGGL((tower_pipe_kernel<384, 512, 256, 256>), grid, block, lds, st, a);
        return hipGetLastError();
    }
    ProfScope prof("tower_fused_16rows", flops * (double)rows, (double)rows * 4.0 * (p->dims[0] + p->dims[p->n_layers]), st);
    hipLaunchKernelGGL(tower_small_kernel, grid, block, lds, st, a);
    return hipGetLastError();
}

}  // namespace amdrec
