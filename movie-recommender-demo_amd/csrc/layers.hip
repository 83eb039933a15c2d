// Tower and ranker forward passes (eval mode) as chains of fp32-MFMA GEMMs with fused
// epilogues.  Orientation: P = weights [out_features][K] (the "small" operand, re-read from
// L2 by every block), Q = data rows [rows][K] (streamed once).  In the accumulator a LANE
// holds one data row and its REGISTERS hold output features, so per-row reductions
// (LayerNorm, L2 norm) are in-lane + one lane-half exchange + one 2-wave LDS exchange.
//
//   tower  (two_tower_model.py:98-121 / :167-184): gather+concat -> [Linear+BN(folded)+ReLU]*n
//                                                  -> Linear -> L2 normalise
//   ranker (transformer_ranker.py:332-380): gather+concat -> Linear(+pos[0]) ->
//          3x { x = LN(x + W_o(W_v x)) ; x = LN(x + W_2 relu(W_1 x)) } -> 3x cross -> 3 heads
//          (seq_len == 1, :358, makes the attention exactly W_o(W_v x + b_v) + b_o: SURVEY fact 1)
#include <type_traits>
#include "gemm_core.hpp"
#include "../../include/amdrec.h"

namespace amdrec {

using ShapeWide = Shape<2, 2, 4, 2>;    // 256 features x 128 rows per workgroup (4 waves)
using ShapeNarrow = Shape<2, 2, 1, 4>;  //  64 features x 256 rows per workgroup
// Small batches (a single request = 500 candidate rows): with the shapes above a layer occupies 4 CUs and
// every wave runs a 128-MFMA chain per K-step (~40 us per 256x256 layer).  These shapes give each of 8
// waves ONE 32x32 tile: 16 workgroups for 500 rows and a 16-MFMA chain per K-step.
using ShapeSmall = Shape<8, 1, 1, 1>;        // 256 features x 32 rows, 8 waves
using ShapeSmallNarrow = Shape<2, 4, 1, 1>;  //  64 features x 128 rows, 8 waves
constexpr long long SMALL_ROWS = 8192;       // rows <= this use the small shapes

// Epilogue data movement.  In the accumulator a lane owns row q(j) and, per tile i and register group g,
// the 4 consecutive features 32*i + 8*g + 4*(lane>>5) + {0..3}: a direct store instruction would cover
// 32 rows x 32 bytes (quarter cache lines).  Instead every 32x32 tile goes through a wave-private 4 KB
// LDS tile (16-byte chunk index XOR (row & 7): conflict-free both ways) so that each global store / load
// instruction moves 8 rows x 128 bytes = whole cache lines (measured on the layer shapes: +4.5 % at
// K = 1024, +5 % at K = 256, +7.5 % for the 1024-wide FFN output; tools/gemm_probe.hip).
// LDS instructions of one wave execute in order, so the tile needs no barrier.
#define AMDREC_EPI_FENCE() asm volatile("" ::: "memory")   // bounds the loads in flight (register pressure)
#define FOFF(i, g) ((i) * 32 + (g) * 8)
constexpr int EPI_TILE_FLOATS = 1024;        // 32 rows x 32 floats per wave
constexpr int EPI_TILE_BASE = 1024;          // floats reserved below the tiles for row reductions
constexpr size_t epi_lds_bytes(int nwaves) { return (size_t)(EPI_TILE_BASE + nwaves * EPI_TILE_FLOATS) * sizeof(float); }

struct WaveTile {
    float* t;
    int r, h, lr, lc;                         // acc view: row r, half h;  line view: row lr (+8*pass), chunk lc
    __device__ __forceinline__ WaveTile(float* smem, int lane)
        : t(smem + EPI_TILE_BASE + (threadIdx.x >> 6) * EPI_TILE_FLOATS), r(lane & 31), h(lane >> 5), lr(lane >> 3),
          lc(lane & 7) {}
    __device__ __forceinline__ void put_acc(const f32x4 (&v)[4]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(t + r * 32 + (((2 * g + h) ^ (r & 7)) << 2)) = v[g];
    }
    __device__ __forceinline__ void get_acc(f32x4 (&v)[4]) const {
#pragma unroll
        for (int g = 0; g < 4; ++g) v[g] = *reinterpret_cast<const f32x4*>(t + r * 32 + (((2 * g + h) ^ (r & 7)) << 2));
    }
    __device__ __forceinline__ f32x4 get_line(int pass) const {
        const int row = lr + 8 * pass;
        return *reinterpret_cast<const f32x4*>(t + row * 32 + ((lc ^ (row & 7)) << 2));
    }
    __device__ __forceinline__ void put_line(int pass, f32x4 v) {
        const int row = lr + 8 * pass;
        *reinterpret_cast<f32x4*>(t + row * 32 + ((lc ^ (row & 7)) << 2)) = v;
    }
};

// full-line store of one 32x32 tile held in acc layout `v`: rows row0.., features feat0..
template <bool FULL>
__device__ __forceinline__ void tile_store(WaveTile& wt, const f32x4 (&v)[4], float* out, long long ld, long long row0,
                                           long long rows, int feat0, int nout) {
    wt.put_acc(v);
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const f32x4 line = wt.get_line(pass);
        const long long row = row0 + wt.lr + 8 * pass;
        const int f = feat0 + wt.lc * 4;
        if (row < rows && (FULL || f < nout)) *reinterpret_cast<f32x4*>(out + row * ld + f) = line;
    }
}
// full-line load of one 32x32 tile into acc layout (rows past the end are clamped, features masked)
template <bool FULL>
__device__ __forceinline__ void tile_load(WaveTile& wt, f32x4 (&v)[4], const float* src, long long ld, long long row0,
                                          long long rows, int feat0, int nout) {
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        long long row = row0 + wt.lr + 8 * pass;
        row = row < rows ? row : rows - 1;
        const int f = feat0 + wt.lc * 4;
        const f32x4 line = (FULL || f < nout) ? *reinterpret_cast<const f32x4*>(src + row * ld + f)
                                              : f32x4{0.f, 0.f, 0.f, 0.f};
        wt.put_line(pass, line);
    }
    wt.get_acc(v);
}

// out[row][f] = act(acc + bias[f])
template <bool FULL>
struct EpiLinearT {
    static constexpr const char* name = "linear";
    static constexpr double out_bytes_per_elem = 1.0;
    static constexpr size_t lds_bytes(int nwaves) { return epi_lds_bytes(nwaves); }
    const float* bias;
    float* out;
    long long ldo;
    long long rows;
    int nout;
    int relu;
    template <class A>
    __device__ void operator()(A& acc, float* smem) const {
        constexpr int TP = A::TP, TQ = A::TQ;
        const int lane = threadIdx.x & 63;
        WaveTile wt(smem, lane);
        const int f0 = acc.p(0, 0, lane);
        const float* bp = bias + f0;
#pragma unroll
        for (int i = 0; i < TP; ++i) {
            f32x4 b[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                b[g] = (FULL || f0 + FOFF(i, g) < nout) ? *reinterpret_cast<const f32x4*>(bp + FOFF(i, g))
                                                        : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < TQ; ++j) {
                f32x4 v[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x = acc.v[i][j][4 * g + e] + b[g][e];
                        v[g][e] = relu ? fmaxf(x, 0.f) : x;
                    }
                tile_store<FULL>(wt, v, out, ldo, (long long)acc.q0 + j * 32, rows, acc.p0 + i * 32, nout);
            }
        }
    }
};

// out[row][f] = acc + ubias[row / rowdiv][f]: the candidate (ad) half of the feature projection plus its
// user's precomputed half (which already carries the bias and positional row)
template <bool FULL>
struct EpiRowBiasT {
    static constexpr const char* name = "linear";
    static constexpr double out_bytes_per_elem = 1.0;
    static constexpr size_t lds_bytes(int nwaves) { return epi_lds_bytes(nwaves); }
    const float* ubias;     // [n_users][ld]
    float* out;
    long long ld;
    long long rows;
    long long row_base;     // global index of the chunk's first row
    int rowdiv;
    int nout;
    template <class A>
    __device__ void operator()(A& acc, float* smem) const {
        constexpr int TP = A::TP, TQ = A::TQ;
        const int lane = threadIdx.x & 63;
        WaveTile wt(smem, lane);
        const int f0 = acc.p(0, 0, lane);
#pragma unroll
        for (int j = 0; j < TQ; ++j) {
            long long row = acc.q(j, lane);
            row = row < rows ? row : rows - 1;
            const float* up = ubias + ((row_base + row) / rowdiv) * ld + f0;
#pragma unroll
            for (int i = 0; i < TP; ++i) {
                f32x4 v[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 u = (FULL || f0 + FOFF(i, g) < nout) ? *reinterpret_cast<const f32x4*>(up + FOFF(i, g))
                                                                     : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[g][e] = acc.v[i][j][4 * g + e] + u[e];
                }
                tile_store<FULL>(wt, v, out, ld, (long long)acc.q0 + j * 32, rows, acc.p0 + i * 32, nout);
            }
        }
    }
};

// out[row][f] = x0[row][f] * (acc + bias[f]) + xl[row][f]     (FeatureInteractionLayer :201)
template <bool FULL>
struct EpiCrossT {
    static constexpr const char* name = "cross";
    static constexpr double out_bytes_per_elem = 1.0;
    static constexpr size_t lds_bytes(int nwaves) { return epi_lds_bytes(nwaves); }
    const float* bias;
    const float* x0;
    const float* xl;
    float* out;
    long long ld;
    long long rows;
    int nout;
    template <class A>
    __device__ void operator()(A& acc, float* smem) const {
        constexpr int TP = A::TP, TQ = A::TQ;
        const int lane = threadIdx.x & 63;
        WaveTile wt(smem, lane);
        const int f0 = acc.p(0, 0, lane);
        const float* bp = bias + f0;
#pragma unroll
        for (int i = 0; i < TP; ++i) {
            f32x4 b[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                b[g] = (FULL || f0 + FOFF(i, g) < nout) ? *reinterpret_cast<const f32x4*>(bp + FOFF(i, g))
                                                        : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < TQ; ++j) {
                const long long row0 = (long long)acc.q0 + j * 32;
                f32x4 a0[4], al[4], v[4];
                tile_load<FULL>(wt, a0, x0, ld, row0, rows, acc.p0 + i * 32, nout);
                tile_load<FULL>(wt, al, xl, ld, row0, rows, acc.p0 + i * 32, nout);
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[g][e] = a0[g][e] * (acc.v[i][j][4 * g + e] + b[g][e]) + al[g][e];
                tile_store<FULL>(wt, v, out, ld, row0, rows, acc.p0 + i * 32, nout);
                AMDREC_EPI_FENCE();
            }
        }
    }
};

// Row statistics over the nout (<= 256) features of a row.  The workgroup is one P tile wide
// (ShapeWide: 2 waves x 128 features), so: in-lane sum over the lane's 64 values, exchange
// with lane^32 (other row-group half of the same tiles), exchange between the 2 feature
// waves through LDS.  `red` is [2 phases][WP][BQ rows] (<= EPI_TILE_BASE floats).
template <int TQ, int BQ, int WP>
__device__ __forceinline__ void row_allreduce(float (&part)[TQ], float* red, int phase, int wp, int wq, int lane) {
    static_assert(2 * WP * BQ <= EPI_TILE_BASE, "reduction scratch overlaps the epilogue tiles");
#pragma unroll
    for (int j = 0; j < TQ; ++j) part[j] += __shfl_xor(part[j], 32, 64);
    float* r = red + phase * WP * BQ;
    if (lane < 32) {
#pragma unroll
        for (int j = 0; j < TQ; ++j) r[wp * BQ + wq * TQ * 32 + j * 32 + lane] = part[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TQ; ++j) {
        const int idx = wq * TQ * 32 + j * 32 + (lane & 31);
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < WP; ++w) a += r[w * BQ + idx];     // fixed order: deterministic
        part[j] = a;
    }
}

// out = LayerNorm(resid + acc + bias) * gamma + beta   (transformer_ranker.py:149, :153; eps 1e-5)
template <bool FULL>
struct EpiResidualLNT {
    static constexpr const char* name = "residual_ln";
    static constexpr double out_bytes_per_elem = 1.0;
    static constexpr size_t lds_bytes(int nwaves) { return epi_lds_bytes(nwaves); }
    const float* bias;
    const float* resid;
    const float* gamma;
    const float* beta;
    float* out;
    long long ld;
    long long rows;
    int nout;
    float eps;
    template <class A>
    __device__ void operator()(A& acc, float* smem) const {
        constexpr int TP = A::TP, TQ = A::TQ;
        static_assert(A::WP * TP * 32 == 256, "LayerNorm epilogue: the workgroup spans one 256-feature P tile");
        const int lane = threadIdx.x & 63;
        const int wp = acc.wp, wq = acc.wq;
        WaveTile wt(smem, lane);
        const int f0 = acc.p(0, 0, lane);
        const float* bp = bias + f0;
        float s[TQ];
#pragma unroll
        for (int j = 0; j < TQ; ++j) s[j] = 0.f;
#pragma unroll
        for (int i = 0; i < TP; ++i) {
            f32x4 b[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                b[g] = (FULL || f0 + FOFF(i, g) < nout) ? *reinterpret_cast<const f32x4*>(bp + FOFF(i, g))
                                                        : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < TQ; ++j) {
                f32x4 rs[4];
                tile_load<FULL>(wt, rs, resid, ld, (long long)acc.q0 + j * 32, rows, acc.p0 + i * 32, nout);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bool fv = FULL || f0 + FOFF(i, g) < nout;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = fv ? acc.v[i][j][4 * g + e] + b[g][e] + rs[g][e] : 0.f;
                        acc.v[i][j][4 * g + e] = x;
                        s[j] += x;
                    }
                }
            }
            AMDREC_EPI_FENCE();
        }
        row_allreduce<TQ, A::BQ, A::WP>(s, smem, 0, wp, wq, lane);
        const float inv_n = 1.0f / (float)nout;
        float mean[TQ], q2[TQ];
#pragma unroll
        for (int j = 0; j < TQ; ++j) {
            mean[j] = s[j] * inv_n;
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < TP; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bool fv = FULL || f0 + FOFF(i, g) < nout;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float dlt = fv ? acc.v[i][j][4 * g + e] - mean[j] : 0.f;
                        a += dlt * dlt;
                    }
                }
            q2[j] = a;
        }
        row_allreduce<TQ, A::BQ, A::WP>(q2, smem, 1, wp, wq, lane);
        float rstd[TQ];
#pragma unroll
        for (int j = 0; j < TQ; ++j) rstd[j] = 1.0f / sqrtf(q2[j] * inv_n + eps);
        const float* gp = gamma + f0;
        const float* tp = beta + f0;
#pragma unroll
        for (int i = 0; i < TP; ++i) {
            f32x4 ga[4], be[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const bool fv = FULL || f0 + FOFF(i, g) < nout;
                ga[g] = fv ? *reinterpret_cast<const f32x4*>(gp + FOFF(i, g)) : f32x4{0.f, 0.f, 0.f, 0.f};
                be[g] = fv ? *reinterpret_cast<const f32x4*>(tp + FOFF(i, g)) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int j = 0; j < TQ; ++j) {
                f32x4 v[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[g][e] = (acc.v[i][j][4 * g + e] - mean[j]) * rstd[j] * ga[g][e] + be[g][e];
                tile_store<FULL>(wt, v, out, ld, (long long)acc.q0 + j * 32, rows, acc.p0 + i * 32, nout);
            }
            AMDREC_EPI_FENCE();
        }
    }
};

// out = (acc + bias) / max(||acc + bias||_2, eps)      (F.normalize, two_tower_model.py:119)
template <bool FULL>
struct EpiL2NormT {
    static constexpr const char* name = "l2norm";
    static constexpr double out_bytes_per_elem = 1.0;
    static constexpr size_t lds_bytes(int nwaves) { return epi_lds_bytes(nwaves); }
    const float* bias;
    float* out;
    long long ldo;
    long long rows;
    int nout;
    float eps;
    template <class A>
    __device__ void operator()(A& acc, float* smem) const {
        constexpr int TP = A::TP, TQ = A::TQ;
        static_assert(A::WP * TP * 32 == 256, "L2-norm epilogue: the workgroup spans one 256-feature P tile");
        const int lane = threadIdx.x & 63;
        const int wp = acc.wp, wq = acc.wq;
        WaveTile wt(smem, lane);
        const int f0 = acc.p(0, 0, lane);
        const float* bp = bias + f0;
        float s[TQ];
#pragma unroll
        for (int j = 0; j < TQ; ++j) s[j] = 0.f;
#pragma unroll
        for (int i = 0; i < TP; ++i) {
            f32x4 b[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
                b[g] = (FULL || f0 + FOFF(i, g) < nout) ? *reinterpret_cast<const f32x4*>(bp + FOFF(i, g))
                                                        : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < TQ; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const bool fv = FULL || f0 + FOFF(i, g) < nout;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = fv ? acc.v[i][j][4 * g + e] + b[g][e] : 0.f;
                        acc.v[i][j][4 * g + e] = x;
                        s[j] += x * x;
                    }
                }
            AMDREC_EPI_FENCE();
        }
        row_allreduce<TQ, A::BQ, A::WP>(s, smem, 0, wp, wq, lane);
#pragma unroll
        for (int j = 0; j < TQ; ++j) {
            const float inv = 1.0f / fmaxf(sqrtf(s[j]), eps);
#pragma unroll
            for (int i = 0; i < TP; ++i) {
                f32x4 v[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[g][e] = acc.v[i][j][4 * g + e] * inv;
                tile_store<FULL>(wt, v, out, ldo, (long long)acc.q0 + j * 32, rows, acc.p0 + i * 32, nout);
            }
        }
    }
};

// out[row] = dot(h[row][0..n), w) + b   (last Linear(64,1) of a prediction head + squeeze)
__global__ __launch_bounds__(256) void rowdot_kernel(const float* h, long long ldh, int n, const float* w,
                                                     const float* b, float* out, long long rows) {
    // 16 lanes per row, float4 each
    const long long row = ((long long)blockIdx.x * 256 + threadIdx.x) >> 4;
    const int sub = threadIdx.x & 15;
    float a = 0.f;
    if (row < rows) {
        for (int c = sub * 4; c < n; c += 64) {
            f32x4 x = *reinterpret_cast<const f32x4*>(h + row * ldh + c);
            f32x4 y = *reinterpret_cast<const f32x4*>(w + c);
            a += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
        }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (row < rows && sub == 0) out[row] = a + b[0];
}

// Validate categorical indices (torch raises IndexError; a kernel must not fault): flag = 1
// if any index is outside [0, card).  The loaders clamp, so nothing reads out of bounds.
__global__ void check_index_kernel(const long long* cat, long long rows, int F, const int* card, int* flag) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * F) return;
    long long v = cat[i];
    if (v < 0 || v >= card[i % F]) *flag = 1;
}

static inline DenseRows dense(const float* p, long long rows, long long ld, int K) {
    return DenseRows{p, rows, (int)ld, K, 30, 1ll << 30};
}

// y = epilogue(x W^T): dispatch on the output width; FULL = the width fills whole P tiles, so the
// epilogue needs no feature-bound checks (true for every layer of the default architecture).
template <class SW, class SN, bool ALLOW_NARROW, template <bool> class EpiT, class LoadQ, class... EpiArgs>
static hipError_t linear_shapes(const float* W, int ldw, int nout, const LoadQ& lq, long long rows, hipStream_t st,
                                int k_alg, EpiArgs... ea) {
    DenseRows lp = dense(W, nout, ldw, ldw);
    if constexpr (ALLOW_NARROW) {
        if (nout <= 64) {
            if (nout == SN::BP) return launch_gemm<SN, true>(lp, lq, EpiT<true>{ea...}, ldw, nout, rows, st, k_alg);
            return launch_gemm<SN, true>(lp, lq, EpiT<false>{ea...}, ldw, nout, rows, st, k_alg);
        }
    }
    if (nout % SW::BP == 0) return launch_gemm<SW, true>(lp, lq, EpiT<true>{ea...}, ldw, nout, rows, st, k_alg);
    return launch_gemm<SW, true>(lp, lq, EpiT<false>{ea...}, ldw, nout, rows, st, k_alg);
}
// The error-compensated bf16 path (gemm_core.hpp "x6") for the big passes: needs the host-split weight planes, a dense
// fp32 row operand and at least one full 256-feature tile.
template <template <bool> class EpiT, class LoadQ, class... EpiArgs>
static bool try_x6(hipError_t& e, const uint16_t* Wx6, int ldw, int nout, const LoadQ& lq, long long rows, hipStream_t st,
                   int k_alg, EpiArgs... ea) {
    if constexpr (std::is_same<LoadQ, DenseRows>::value) {
        if (Wx6 == nullptr || rows <= SMALL_ROWS || nout < 256) return false;
        if (nout % 256 == 0) e = launch_gemm_x6(Wx6, nout, lq, EpiT<true>{ea...}, ldw, rows, st, k_alg);
        else e = launch_gemm_x6(Wx6, nout, lq, EpiT<false>{ea...}, ldw, rows, st, k_alg);
        return true;
    }
    return false;
}
// narrow outputs (<= 64 features) may use the 64-feature shapes
template <template <bool> class EpiT, class LoadQ, class... EpiArgs>
static hipError_t linear(const float* W, const uint16_t* Wx6, int ldw, int nout, const LoadQ& lq, long long rows,
                         hipStream_t st, int k_alg, EpiArgs... ea) {
    hipError_t e;
    if (try_x6<EpiT>(e, Wx6, ldw, nout, lq, rows, st, k_alg, ea...)) return e;
    if (rows <= SMALL_ROWS)
        return linear_shapes<ShapeSmall, ShapeSmallNarrow, true, EpiT>(W, ldw, nout, lq, rows, st, k_alg, ea...);
    return linear_shapes<ShapeWide, ShapeNarrow, true, EpiT>(W, ldw, nout, lq, rows, st, k_alg, ea...);
}
// epilogues that need a whole 256-feature row in one workgroup (LayerNorm, L2 norm)
template <template <bool> class EpiT, class LoadQ, class... EpiArgs>
static hipError_t linear_wide(const float* W, const uint16_t* Wx6, int ldw, int nout, const LoadQ& lq, long long rows,
                              hipStream_t st, int k_alg, EpiArgs... ea) {
    hipError_t e;
    if (try_x6<EpiT>(e, Wx6, ldw, nout, lq, rows, st, k_alg, ea...)) return e;
    if (rows <= SMALL_ROWS)
        return linear_shapes<ShapeSmall, ShapeSmallNarrow, false, EpiT>(W, ldw, nout, lq, rows, st, k_alg, ea...);
    return linear_shapes<ShapeWide, ShapeNarrow, false, EpiT>(W, ldw, nout, lq, rows, st, k_alg, ea...);
}

static inline int ilog2(int v) {
    int s = 0;
    while ((1 << s) < v) ++s;
    return s;
}

// Rows per pass.  A 256-feature layer launches rows/256 workgroups, so a pass must be >= 65536 rows to
// give every one of the 256 CUs a block and several times that to keep the last wave of blocks full
// (measured: 32768-row passes left half the chip idle, 45-74 TF; fp32 GEMMs are MFMA-bound, not
// HBM-bound, so keeping a pass inside the Infinity Cache buys nothing here).
constexpr long long ROW_CHUNK = 262144;

}  // namespace amdrec

using namespace amdrec;

// ============================== towers ==============================================
static int tower_check(const amdrec_tower_params* p) {
    REQUIRE(p != nullptr, "params is null");
    REQUIRE(p->n_feat >= 1 && p->n_feat <= 64, "n_feat out of range");
    REQUIRE(p->emb_dim >= 4 && (p->emb_dim & (p->emb_dim - 1)) == 0, "emb_dim must be a power of two >= 4");
    REQUIRE(p->n_num >= 0, "n_num < 0");
    REQUIRE(p->n_layers >= 1 && p->n_layers <= AMDREC_MAX_LAYERS, "n_layers out of range");
    REQUIRE(p->dims[0] == p->n_feat * p->emb_dim + p->n_num, "dims[0] != n_feat*emb_dim + n_num");
    for (int l = 0; l < p->n_layers; ++l) {
        REQUIRE(p->dims[l + 1] >= 4 && p->dims[l + 1] % 4 == 0, "layer width must be a multiple of 4");
        REQUIRE(p->ldw[l] % 32 == 0 && p->ldw[l] >= p->dims[l], "ldw must be a multiple of 32 and >= K");
        REQUIRE(p->w[l] && p->b[l], "null weight pointer");
    }
    REQUIRE(p->dims[p->n_layers] <= 256, "output_dim > 256 is not supported by the fused L2-norm epilogue");
    REQUIRE(p->tables && p->table_off && p->cards, "null table pointer");
    return AMDREC_OK;
}

static size_t tower_ws_bytes(const amdrec_tower_params* p, long long rows) {
    long long chunk = rows < ROW_CHUNK ? rows : ROW_CHUNK;
    int wmax = 4;
    for (int l = 1; l < p->n_layers; ++l) wmax = p->dims[l] > wmax ? p->dims[l] : wmax;
    return align_up((size_t)chunk * wmax * 4, 256) * 2;
}

extern "C" int amdrec_tower_workspace(const amdrec_tower_params* p, int64_t rows, size_t* bytes) {
    int rc = tower_check(p);
    if (rc) return rc;
    REQUIRE(bytes && rows >= 0, "bad arguments");
    *bytes = tower_ws_bytes(p, rows);
    return AMDREC_OK;
}

namespace amdrec {   // tower_small.hip: the whole tower in one launch for batches of <= 4096 rows
bool tower_small_ok(const amdrec_tower_params* p, long long rows);
hipError_t tower_small_run(const amdrec_tower_params* p, const long long* cat, const float* num, long long rows, float* out,
                           long long ld_out, hipStream_t st);
}

extern "C" int amdrec_tower_forward(const amdrec_tower_params* p, const int64_t* cat, const float* num,
                                    int64_t rows, float* out, int64_t ld_out, int* bad_index_flag,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    int rc = tower_check(p);
    if (rc) return rc;
    if (rows <= 0) return AMDREC_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    REQUIRE(cat && out && (num || p->n_num == 0), "null pointer");
    REQUIRE(ld_out % 4 == 0 && ld_out >= p->dims[p->n_layers], "bad ld_out");
    size_t need = tower_ws_bytes(p, rows);
    if (!workspace || workspace_bytes < need)
        return set_error(AMDREC_EWORKSPACE, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    const long long chunk = rows < ROW_CHUNK ? rows : ROW_CHUNK;
    int wmax = 4;
    for (int l = 1; l < p->n_layers; ++l) wmax = p->dims[l] > wmax ? p->dims[l] : wmax;
    float* bufs[2] = {reinterpret_cast<float*>(workspace),
                      reinterpret_cast<float*>((char*)workspace + align_up((size_t)chunk * wmax * 4, 256))};
    if (bad_index_flag) {
        long long n = rows * p->n_feat;
        hipLaunchKernelGGL(check_index_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                           (const long long*)cat, (long long)rows, p->n_feat, p->cards, bad_index_flag);
    }
    if (tower_small_ok(p, rows)) {                      // serving batches: one launch for the whole tower
        HIP_TRY(tower_small_run(p, (const long long*)cat, num, rows, out, (long long)ld_out, st));
        return AMDREC_OK;
    }
    for (long long r0 = 0; r0 < rows; r0 += chunk) {
        const long long m = rows - r0 < chunk ? rows - r0 : chunk;
        EmbConcatRows g{};
        g.tables = p->tables; g.off = p->table_off; g.card = p->cards;
        g.cat0 = (const long long*)cat; g.cat1 = nullptr; g.rowmap1 = nullptr;
        g.num = num;
        g.row_base = r0;
        g.rows = m; g.F = p->n_feat; g.F0 = p->n_feat; g.E = p->emb_dim; g.eshift = ilog2(p->emb_dim);
        g.n_num = p->n_num; g.cat0_rowdiv = 1;
        const float* cur = nullptr;
        int curw = 0;
        for (int l = 0; l < p->n_layers; ++l) {
            const bool last = l == p->n_layers - 1;
            const int nout = p->dims[l + 1];
            hipError_t e;
            if (last) {
                float* dst = out + r0 * ld_out;
                e = (l == 0) ? linear_wide<EpiL2NormT>(p->w[l], nullptr, p->ldw[l], nout, g, m, st, p->dims[l], p->b[l], dst,
                                                       (long long)ld_out, m, nout, 1e-12f)
                             : linear_wide<EpiL2NormT>(p->w[l], nullptr, p->ldw[l], nout, dense(cur, m, curw, curw), m, st,
                                                       p->dims[l], p->b[l], dst, (long long)ld_out, m, nout, 1e-12f);
            } else {
                float* dst = bufs[l & 1];
                e = (l == 0) ? linear<EpiLinearT>(p->w[l], nullptr, p->ldw[l], nout, g, m, st, p->dims[l], p->b[l], dst,
                                                  (long long)nout, m, nout, 1)
                             : linear<EpiLinearT>(p->w[l], nullptr, p->ldw[l], nout, dense(cur, m, curw, curw), m, st,
                                                  p->dims[l], p->b[l], dst, (long long)nout, m, nout, 1);
                cur = dst;
                curw = nout;
            }
            HIP_TRY(e);
        }
    }
    if (p->renormalize)                                   // the general path: a second launch (amdrec_l2_normalize in place)
        return amdrec_l2_normalize(out, ld_out, out, ld_out, rows, p->dims[p->n_layers], stream);
    return AMDREC_OK;
}

// ============================== ranker ==============================================
namespace amdrec {   // ranker_x3.hip: the fp16x3 row-owner engine (everything after the feature projection in one kernel)
bool ranker_x3_wanted(const amdrec_ranker_params* p, long long rows);
int ranker_x3_run(const amdrec_ranker_params* p, const float* X, long long ldx, const float* U, const long long* rowmap,
                  long long row_base, int rowdiv, long long n_cache, long long rows, float* scratch, float* logits,
                  long long ld_logits, hipStream_t st);
}
static int ranker_check(const amdrec_ranker_params* p) {
    REQUIRE(p != nullptr, "params is null");
    REQUIRE(p->n_user_feat >= 0 && p->n_ad_feat >= 0 && p->n_user_feat + p->n_ad_feat >= 1 &&
                p->n_user_feat + p->n_ad_feat <= 128, "feature counts out of range");
    REQUIRE(p->emb_dim >= 4 && (p->emb_dim & (p->emb_dim - 1)) == 0, "emb_dim must be a power of two >= 4");
    REQUIRE(p->d_model >= 4 && p->d_model % 4 == 0 && p->d_model <= 256, "d_model must be a multiple of 4, <= 256");
    REQUIRE(p->d_ff >= 4 && p->d_ff % 4 == 0, "d_ff must be a multiple of 4");
    REQUIRE(p->n_layers >= 0 && p->n_layers <= AMDREC_MAX_LAYERS, "n_layers out of range");
    REQUIRE(p->n_cross >= 0 && p->n_cross <= AMDREC_MAX_LAYERS, "n_cross out of range");
    REQUIRE(p->n_tasks >= 1 && p->n_tasks <= AMDREC_MAX_TASKS, "n_tasks out of range");
    REQUIRE(p->head_h1 % 4 == 0 && p->head_h2 % 4 == 0 && p->head_h1 >= 4 && p->head_h2 >= 4, "bad head widths");
    REQUIRE(p->tables && p->table_off && p->cards && p->w_proj && p->b_proj, "null pointer in params");
    return AMDREC_OK;
}

struct RankerWs {
    size_t off_x, off_t, off_x0, off_h, off_u, bytes;
    long long chunk;
};
static RankerWs ranker_ws(const amdrec_ranker_params* p, long long rows, long long n_users) {
    RankerWs w;
    w.chunk = rows < ROW_CHUNK ? rows : ROW_CHUNK;
    if (w.chunk < 1) w.chunk = 1;
    size_t dm = (size_t)w.chunk * p->d_model * 4;
    long long hw = p->d_ff;
    long long headw = (long long)p->n_tasks * (p->head_h1 + p->head_h2);
    if (headw > hw) hw = headw;
    size_t o = 0;
    w.off_x = o;  o = align_up(o + dm, 256);
    w.off_t = o;  o = align_up(o + dm, 256);
    // X0 doubles as the row-owner engine's x0 scratch, which is addressed in whole 128-row workgroups
    w.off_x0 = o; o = align_up(o + (size_t)((w.chunk + 127) / 128 * 128) * p->d_model * 4, 256);
    w.off_h = o;  o = align_up(o + (size_t)w.chunk * hw * 4, 256);
    w.off_u = o;  o = align_up(o + (size_t)(n_users > 0 ? n_users : 0) * p->d_model * 4, 256);
    w.bytes = o;
    return w;
}

extern "C" int amdrec_ranker_workspace(const amdrec_ranker_params* p, int64_t rows, size_t* bytes) {
    int rc = ranker_check(p);
    if (rc) return rc;
    REQUIRE(bytes && rows >= 0, "bad arguments");
    *bytes = ranker_ws(p, rows, rows).bytes;      // upper bound: one user row per batch row
    return AMDREC_OK;
}

// x0[r] = ad_proj_cache[ad row of r] + U[user of r]  (the cached form of the EpiRowBiasT projection: same addends,
// same order).  One wave per row.
__global__ __launch_bounds__(256) void proj_gather_kernel(const float* cache, long long ldc, long long n_cache,
                                                          const long long* rowmap, long long row_base, const float* U,
                                                          int dm, int rowdiv, float* X, long long m) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= m) return;
    const long long gr = row_base + r;
    long long a = rowmap ? rowmap[gr] : gr;
    a = a < 0 ? 0 : (a >= n_cache ? n_cache - 1 : a);        // clamped like the gather loader (reported separately)
    const f32x4* cp = reinterpret_cast<const f32x4*>(cache + a * ldc);
    const f32x4* up = reinterpret_cast<const f32x4*>(U + (gr / rowdiv) * dm);
    f32x4* xp = reinterpret_cast<f32x4*>(X + r * dm);
    for (int c = lane; c < (dm >> 2); c += 64) {
        const f32x4 a4 = cp[c], u4 = up[c];
        xp[c] = f32x4{a4[0] + u4[0], a4[1] + u4[1], a4[2] + u4[2], a4[3] + u4[3]};
    }
}

extern "C" int amdrec_ranker_project_ads(const amdrec_ranker_params* p, const int64_t* ad_cat, int64_t n_ads,
                                         float* out, int64_t ld_out, void* workspace, size_t workspace_bytes,
                                         void* stream) {
    int rc = ranker_check(p);
    if (rc) return rc;
    if (n_ads <= 0) return AMDREC_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int dm = p->d_model, F0 = p->n_user_feat, F = p->n_user_feat + p->n_ad_feat;
    REQUIRE(p->w_proj_ad && p->n_ad_feat > 0, "params carry no split projection (w_proj_ad)");
    REQUIRE(ad_cat && out, "null pointer");
    REQUIRE(ld_out >= dm && ld_out % 4 == 0 && ((uintptr_t)out % 16) == 0, "bad output layout");
    REQUIRE(n_ads < (1ll << 31) - 1024, "n_ads out of range");
    const size_t need = align_up((size_t)dm * 4, 256);
    if (!workspace || workspace_bytes < need)
        return set_error(AMDREC_EWORKSPACE, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    float* zero = reinterpret_cast<float*>(workspace);
    HIP_TRY(hipMemsetAsync(zero, 0, (size_t)dm * 4, st));
    EmbConcatRows ga{};
    ga.tables = p->tables; ga.off = p->table_off + F0; ga.card = p->cards + F0;
    ga.cat0 = nullptr; ga.cat1 = (const long long*)ad_cat; ga.rowmap1 = nullptr; ga.num = nullptr;
    ga.row_base = 0; ga.rows = n_ads; ga.rows1 = n_ads; ga.F = F - F0; ga.F0 = 0; ga.E = p->emb_dim;
    ga.eshift = ilog2(p->emb_dim); ga.n_num = 0; ga.cat0_rowdiv = 1;
    // the same GEMM (shape, K order) as the uncached candidate half, with an all-zero "user row"
    HIP_TRY(linear_wide<EpiRowBiasT>(p->w_proj_ad, nullptr, p->ldw_proj_ad, dm, ga, n_ads, st, (F - F0) * p->emb_dim,
                                     (const float*)zero, out, (long long)ld_out, n_ads, 0ll, 0x7fffffff, dm));
    return AMDREC_OK;
}

// The user half of the projection for a handful of requests (hoisted form: U[u] = W_user [user emb | numerical] + b).  The tile
// GEMM spends 18 us on ONE user row (7 dependent k-steps of global load -> LDS -> MFMA on four workgroups).  Here a workgroup
// stages the user's 205 features in LDS and computes 32 output features, eight lanes per feature: the eight read 128
// contiguous bytes of the weight row per step (7 steps, all in flight), fp32 FMA chains, then a three-step lane reduction.
// (One thread per output feature reads its row alone: 64 cache lines per load instruction, 12 us on one CU.)
constexpr int USER_PROJ_SMALL_MAX = 64, USER_PROJ_SMALL_K = 256;     // users per call; features (8 steps of 32)
__global__ __launch_bounds__(256) void user_proj_small_kernel(EmbConcatRows g, const float* W, int ldw, int K, const float* bias,
                                                              float* U, int dm) {
    __shared__ __attribute__((aligned(16))) float feat[USER_PROJ_SMALL_K];
    const long long u = blockIdx.x;
    const int K4 = (K + 3) / 4 * 4;
    const int n = blockIdx.y * 32 + (threadIdx.x >> 3), j = threadIdx.x & 7;
    const float* w = W + (long long)(n < dm ? n : dm - 1) * ldw;
    f32x4 wv[USER_PROJ_SMALL_K / 32];                              // the weight loads go out first: they do not depend on the
#pragma unroll                                                     // features, whose gather is two dependent loads deep
    for (int i = 0; i < USER_PROJ_SMALL_K / 32; ++i) {
        const int k = 4 * j + 32 * i;
        wv[i] = k < K4 ? *reinterpret_cast<const f32x4*>(w + k) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const EmbConcatRows::RowState rs = g.row_state(u);
    for (int k = threadIdx.x * 4; k < K4; k += 1024) *reinterpret_cast<f32x4*>(feat + k) = g.load(rs, k);
    __syncthreads();
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int i = 0; i < USER_PROJ_SMALL_K / 32; ++i) {
        const int k = 4 * j + 32 * i;
        const f32x4 x = k < K4 ? *reinterpret_cast<const f32x4*>(feat + k) : f32x4{0.f, 0.f, 0.f, 0.f};
        a0 = __builtin_fmaf(wv[i][0], x[0], a0);
        a1 = __builtin_fmaf(wv[i][1], x[1], a1);
        a2 = __builtin_fmaf(wv[i][2], x[2], a2);
        a3 = __builtin_fmaf(wv[i][3], x[3], a3);
    }
    float a = (a0 + a1) + (a2 + a3);
    a += __shfl_xor(a, 1, 64);
    a += __shfl_xor(a, 2, 64);
    a += __shfl_xor(a, 4, 64);
    if (j == 0 && n < dm) U[u * dm + n] = a + bias[n];
}

extern "C" int amdrec_ranker_forward(const amdrec_ranker_params* p, const int64_t* user_cat, const float* numerical,
                                     int64_t user_rowdiv, const int64_t* ad_cat, const int64_t* ad_rowmap,
                                     int64_t rows, float* out_logits, int64_t ld_logits, int* bad_index_flag,
                                     int64_t n_user_rows, int64_t n_ad_rows, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    int rc = ranker_check(p);
    if (rc) return rc;
    if (rows <= 0) return AMDREC_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    REQUIRE(user_rowdiv >= 1, "user_rowdiv must be >= 1");
    REQUIRE((user_cat || p->n_user_feat == 0) && (ad_cat || p->n_ad_feat == 0) && out_logits, "null pointer");
    REQUIRE(numerical || p->n_num == 0, "numerical is null");
    REQUIRE(ld_logits >= rows, "ld_logits < rows");
    REQUIRE(p->n_ad_feat == 0 || n_ad_rows >= 1, "n_ad_rows must be >= 1");
    REQUIRE(ad_rowmap != nullptr || p->n_ad_feat == 0 || n_ad_rows >= rows, "ad_cat has fewer rows than the batch");
    REQUIRE(p->n_user_feat == 0 || n_user_rows * user_rowdiv >= rows, "user_cat has too few rows");
    const bool hoist = user_rowdiv > 1 && p->w_proj_user && p->w_proj_ad && p->n_ad_feat > 0 &&
                       (p->n_user_feat > 0 || p->n_num > 0);
    const long long n_users = hoist ? (rows + user_rowdiv - 1) / user_rowdiv : 0;
    RankerWs w = ranker_ws(p, rows, rows);        // same layout as the workspace query
    if (!workspace || workspace_bytes < w.bytes)
        return set_error(AMDREC_EWORKSPACE, "workspace too small: need %zu bytes, got %zu", w.bytes, workspace_bytes);
    char* ws = reinterpret_cast<char*>(workspace);
    float* X = reinterpret_cast<float*>(ws + w.off_x);
    float* T = reinterpret_cast<float*>(ws + w.off_t);
    float* X0 = reinterpret_cast<float*>(ws + w.off_x0);
    float* H = reinterpret_cast<float*>(ws + w.off_h);
    float* U = reinterpret_cast<float*>(ws + w.off_u);
    const int dm = p->d_model, F0 = p->n_user_feat, F = p->n_user_feat + p->n_ad_feat;

    if (bad_index_flag) {
        if (F0 && n_user_rows > 0) {
            long long n = n_user_rows * F0;
            hipLaunchKernelGGL(check_index_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                               (const long long*)user_cat, (long long)n_user_rows, F0, p->cards, bad_index_flag);
        }
        if (F - F0 && n_ad_rows > 0) {
            long long n = n_ad_rows * (F - F0);
            hipLaunchKernelGGL(check_index_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                               (const long long*)ad_cat, (long long)n_ad_rows, F - F0, p->cards + F0, bad_index_flag);
        }
    }

    if (hoist) {
        // U[u] = W_user [user emb(u) | numerical(u)] + b_proj (+ pos[0]): once per user row
        EmbConcatRows gu{};
        gu.tables = p->tables; gu.off = p->table_off; gu.card = p->cards;
        gu.cat0 = (const long long*)user_cat; gu.cat1 = nullptr; gu.rowmap1 = nullptr; gu.num = numerical;
        gu.row_base = 0; gu.rows1 = 1; gu.rows = n_users; gu.F = F0; gu.F0 = F0; gu.E = p->emb_dim;
        gu.eshift = ilog2(p->emb_dim); gu.n_num = p->n_num; gu.cat0_rowdiv = 1;
        const int Ku = F0 * p->emb_dim + p->n_num;
        if (n_users <= USER_PROJ_SMALL_MAX && (Ku + 3) / 4 * 4 <= USER_PROJ_SMALL_K && p->ldw_proj_user >= (Ku + 3) / 4 * 4) {
            ProfScope prof("user_proj_small", 2.0 * n_users * dm * Ku, (double)dm * Ku * 4, st);
            hipLaunchKernelGGL(user_proj_small_kernel, dim3((unsigned)n_users, (unsigned)((dm + 31) / 32)), dim3(256), 0, st, gu, p->w_proj_user,
                               (int)p->ldw_proj_user, Ku, p->b_proj, U, dm);
            HIP_TRY(hipGetLastError());
        } else {
            HIP_TRY(linear_wide<EpiLinearT>(p->w_proj_user, nullptr, p->ldw_proj_user, dm, gu, n_users, st, Ku, p->b_proj, U,
                                            (long long)dm, n_users, dm, 0));
        }
    }
    for (long long r0 = 0; r0 < rows; r0 += w.chunk) {
        const long long m = rows - r0 < w.chunk ? rows - r0 : w.chunk;
        // ---- embed + project (+ pos[0], folded into b_proj on the host) ----
        EmbConcatRows g{};
        g.tables = p->tables; g.off = p->table_off; g.card = p->cards;
        g.cat0 = (const long long*)user_cat; g.cat1 = (const long long*)ad_cat;
        g.rowmap1 = (const long long*)ad_rowmap;
        g.num = numerical;
        g.row_base = r0;
        g.rows1 = n_ad_rows > 0 ? n_ad_rows : 1;
        g.rows = m; g.F = F; g.F0 = F0; g.E = p->emb_dim; g.eshift = ilog2(p->emb_dim);
        g.n_num = p->n_num; g.cat0_rowdiv = (int)user_rowdiv;
        // fp16x3 row-owner engine: the rest of the chain is one kernel; with the candidate-side cache it also does the
        // gather (x0 = cache row + user half).  Its x0 scratch is the X0 region (sized in whole 128-row workgroups).
        const bool use_x3 = ranker_x3_wanted(p, m);
        if (use_x3 && hoist && p->ad_proj_cache) {
            int rc3 = ranker_x3_run(p, nullptr, 0, (const float*)U, (const long long*)ad_rowmap, r0, (int)user_rowdiv,
                                    (long long)(n_ad_rows > 0 ? n_ad_rows : 1), m, X0, out_logits + r0, (long long)ld_logits, st);
            if (rc3) return rc3;
            continue;
        }
        if (hoist && p->ad_proj_cache) {
            hipLaunchKernelGGL(proj_gather_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, st, p->ad_proj_cache,
                               (long long)p->ld_ad_proj_cache, (long long)(n_ad_rows > 0 ? n_ad_rows : 1),
                               (const long long*)ad_rowmap, r0, (const float*)U, dm, (int)user_rowdiv, X, m);
        } else if (hoist) {
            // candidate half: ad embeddings only (K = n_ad_feat * emb_dim), plus the user's row of U
            EmbConcatRows ga = g;
            ga.off = p->table_off + F0; ga.card = p->cards + F0;
            ga.cat0 = nullptr; ga.num = nullptr; ga.n_num = 0; ga.F = F - F0; ga.F0 = 0; ga.cat0_rowdiv = 1;
            HIP_TRY(linear_wide<EpiRowBiasT>(p->w_proj_ad, nullptr, p->ldw_proj_ad, dm, ga, m, st, (F - F0) * p->emb_dim,
                                             (const float*)U, X, (long long)dm, m, r0, (int)user_rowdiv, dm));
        } else {
            HIP_TRY(linear_wide<EpiLinearT>(p->w_proj, nullptr, p->ldw_proj, dm, g, m, st, F * p->emb_dim + p->n_num,
                                            p->b_proj, X, (long long)dm, m, dm, 0));
        }
        if (use_x3) {
            int rc3 = ranker_x3_run(p, (const float*)X, (long long)dm, nullptr, nullptr, 0, 1, 0, m, X0, out_logits + r0,
                                    (long long)ld_logits, st);
            if (rc3) return rc3;
            continue;
        }
        // ---- encoder layers ----
        for (int l = 0; l < p->n_layers; ++l) {
            const amdrec_encoder_layer& L = p->layers[l];
            if (L.w_v != nullptr) {
                // T = W_v x + b_v ; X = LN1(X + W_o T + b_o)
                HIP_TRY(linear_wide<EpiLinearT>(L.w_v, nullptr, L.ldw_dm, dm, dense(X, m, dm, dm), m, st, dm, L.b_v, T,
                                                (long long)dm, m, dm, 0));
                HIP_TRY(linear_wide<EpiResidualLNT>(L.w_o, L.w_o_x6, L.ldw_dm, dm, dense(T, m, dm, dm), m, st, dm, L.b_o,
                                                    (const float*)X, L.ln1_g, L.ln1_b, X, (long long)dm, m, dm,
                                                    p->ln_eps));
            } else {
                // host pre-multiplied W_ov = W_o W_v: LN1(X + W_ov X + b_ov), written to the spare buffer
                HIP_TRY(linear_wide<EpiResidualLNT>(L.w_o, L.w_o_x6, L.ldw_dm, dm, dense(X, m, dm, dm), m, st, dm, L.b_o,
                                                    (const float*)X, L.ln1_g, L.ln1_b, T, (long long)dm, m, dm,
                                                    p->ln_eps));
                float* tmp = X; X = T; T = tmp;
            }
            // H = relu(W_1 X + b_1)
            HIP_TRY(linear<EpiLinearT>(L.w_1, L.w_1_x6, L.ldw_dm, p->d_ff, dense(X, m, dm, dm), m, st, dm, L.b_1, H,
                                       (long long)p->d_ff, m, p->d_ff, 1));
            // X = LN2(X + W_2 H + b_2)
            HIP_TRY(linear_wide<EpiResidualLNT>(L.w_2, L.w_2_x6, L.ldw_ff, dm, dense(H, m, p->d_ff, p->d_ff), m, st, p->d_ff,
                                                L.b_2, (const float*)X, L.ln2_g, L.ln2_b, X, (long long)dm, m, dm,
                                                p->ln_eps));
        }
        // ---- cross layers: xl <- x0 * (xl W_i + b_i) + xl ; x0 = X ----
        const float* xl = X;
        for (int c = 0; c < p->n_cross; ++c) {
            float* dst = (c & 1) ? X0 : T;
            HIP_TRY(linear_wide<EpiCrossT>(p->cross_wt[c], p->cross_wt_x6[c], p->ldw_cross, dm, dense(xl, m, dm, dm), m, st, dm,
                                           p->cross_b[c], (const float*)X, xl, dst, (long long)dm, m, dm));
            xl = dst;
        }
        // ---- heads ----
        const int h1 = p->head_h1, h2 = p->head_h2, nt = p->n_tasks;
        float* H1 = H;                                   // [m][nt*h1]
        float* H2 = H + (size_t)m * nt * h1;             // [m][nt*h2]
        HIP_TRY(linear<EpiLinearT>(p->head_w1, p->head_w1_x6, p->ldw_head1, nt * h1, dense(xl, m, dm, dm), m, st, dm, p->head_b1, H1,
                                   (long long)nt * h1, m, nt * h1, 1));
        for (int t = 0; t < nt; ++t) {
            HIP_TRY(linear<EpiLinearT>(p->head_w2[t], nullptr, p->ldw_head2, h2, dense(H1 + t * h1, m, (long long)nt * h1, h1), m,
                                       st, h1, p->head_b2[t], H2 + t * h2, (long long)nt * h2, m, h2, 1));
            hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)((m * 16 + 255) / 256)), dim3(256), 0, st,
                               H2 + t * h2, (long long)nt * h2, h2, p->head_w3[t], p->head_b3[t],
                               out_logits + (long long)t * ld_logits + r0, m);
        }
        HIP_TRY(hipGetLastError());
    }
    return AMDREC_OK;
}
