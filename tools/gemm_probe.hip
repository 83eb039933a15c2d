// Developer probe (not part of the product): times the fp32-MFMA mainloop of csrc/gemm_core.hpp on a
// layer-shaped problem with operand loaders / epilogues selectively stubbed out, to see which part of the
// pipeline bounds it.   hipcc -O3 --offload-arch=gfx950 tools/gemm_probe.hip -o gpurun_out/gemm_probe
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../movie-recommender-demo_amd/csrc/gemm_core.hpp"

namespace amdrec {
thread_local char g_err[512];
int set_error(int c, const char*, ...) { return c; }
bool g_prof_on = false;

ProfScope::ProfScope(const char*, double, double, hipStream_t s) : slot(-1), st(s) {}
ProfScope::~ProfScope() {}

struct NullRows {   // no memory traffic: operand values come from registers
    float v;
    static constexpr bool can_dma = false;
    using RowState = long long;
    __device__ __forceinline__ RowState row_state(long long r) const { return r; }
    __device__ __forceinline__ bool k_valid(int) const { return true; }
    __device__ __forceinline__ f32x4 load(long long r, int k) const {
        float x = v + (float)(r & 7) * 0.125f + (float)(k & 31) * 0.01f;
        return f32x4{x, x, x, x};
    }
};
struct EpiStore {   // plain store of the accumulators (row-major [q][p])
    static constexpr const char* name = "store";
    static constexpr double out_bytes_per_elem = 1.0;
    static constexpr size_t lds_bytes(int) { return 0; }
    float* out; long long ld; long long rows;
    template <class A> __device__ void operator()(A& acc, float*) const {
        constexpr int TP = A::TP, TQ = A::TQ;
        const int lane = threadIdx.x & 63;
        const int f0 = acc.p(0, 0, lane);
#pragma unroll
        for (int j = 0; j < TQ; ++j) {
            long long row = acc.q(j, lane);
            if (row >= rows) continue;
            float* op = out + row * ld + f0;
#pragma unroll
            for (int i = 0; i < TP; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc.v[i][j][4 * g + e];
                    *reinterpret_cast<f32x4*>(op + i * 32 + g * 8) = v;
                }
        }
    }
};
// store through a wave-private, XOR-swizzled LDS tile so that every global store writes full 128-B lines
struct EpiStoreLines {
    static constexpr const char* name = "store_lines";
    static constexpr double out_bytes_per_elem = 1.0;
    static constexpr size_t lds_bytes(int nwaves) { return (size_t)nwaves * 4096; }
    float* out; long long ld; long long rows;
    template <class A> __device__ void operator()(A& acc, float* smem) const {
        constexpr int TP = A::TP, TQ = A::TQ;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        float* tile = smem + wave * 1024;                       // 32 rows x 32 floats
        const int r = lane & 31, h = lane >> 5;
        const int rr = lane >> 3, c = lane & 7;
#pragma unroll
        for (int i = 0; i < TP; ++i)
#pragma unroll
            for (int j = 0; j < TQ; ++j) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc.v[i][j][4 * g + e];
                    *reinterpret_cast<f32x4*>(tile + r * 32 + (((2 * g + h) ^ (r & 7)) << 2)) = v;
                }
#pragma unroll
                for (int pass = 0; pass < 4; ++pass) {
                    const int row_l = rr + 8 * pass;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(tile + row_l * 32 + ((c ^ (row_l & 7)) << 2));
                    const long long row = acc.q0 + j * 32 + row_l;
                    if (row < rows) *reinterpret_cast<f32x4*>(out + row * ld + acc.p0 + i * 32 + c * 4) = v;
                }
            }
    }
};
struct EpiSink {    // keeps the accumulators alive, writes 1 float per lane
    static constexpr const char* name = "sink";
    static constexpr double out_bytes_per_elem = 0.0;
    static constexpr size_t lds_bytes(int) { return 0; }
    float* out;
    template <class A> __device__ void operator()(A& acc, float*) const {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < A::TP; ++i)
#pragma unroll
            for (int j = 0; j < A::TQ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc.v[i][j][r];
        if (s == 12345.678f) out[threadIdx.x] = s;
    }
};
// ---- experiment: weights (P) staged by LDS-DMA (global_load_lds, double-buffered), rows (Q) through registers ----
template <class S, class LoadQ, class Epi>
__global__ __launch_bounds__(S::NT, 2) void gemm_dma_kernel(DenseRows lp, LoadQ lq, Epi epi, int ksteps, TileMap tm) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TP = S::TP, TQ = S::TQ, BP = S::BP, BQ = S::BQ;
    int small, big;
    if (!tm.get(blockIdx.x, small, big)) return;
    const long long prow0 = (long long)small * BP, qrow0 = (long long)big * BQ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wp = wave / S::WQ, wq = wave % S::WQ;
    const int srow = tid >> 3, schunk = tid & 7;
    constexpr int RPP = S::ROWS_PER_PASS, NP = BP / RPP, NQ = BQ / RPP;
    float* pbuf = smem;                       // [2][BP*32]
    float* qbuf = smem + 2 * BP * BK;         // [BQ*32]
    f32x4 rq[NQ];
    auto dma_p = [&](int kt, int buf) {
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int rl = srow + RPP * u;                                   // LDS row of this lane
            long long r = prow0 + rl;
            r = r < lp.rows ? r : lp.rows - 1;
            const int c = schunk ^ ((rl >> 1) & 7);                          // source-side swizzle (LDS image stays linear)
            const float* g = lp.base + r * lp.ld + kt * BK + c * 4;
            float* l = pbuf + buf * BP * BK + ((wave * 8 + RPP * u) * BK);   // wave-uniform base; HW adds lane*16
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)l, 16, 0, 0);
        }
    };
    auto load_q = [&](int kt) {
        const int k = kt * BK + schunk * 4;
#pragma unroll
        for (int u = 0; u < NQ; ++u) rq[u] = lq.load(lq.row_state(qrow0 + srow + RPP * u), k);
    };
    auto store_q = [&]() {
#pragma unroll
        for (int u = 0; u < NQ; ++u) *reinterpret_cast<f32x4*>(qbuf + lds_slot(srow + RPP * u, schunk)) = rq[u];
    };
    Acc<TP, TQ, S::WP, S::WQ> acc;
    acc.p0 = (int)prow0 + wp * TP * 32; acc.q0 = (int)qrow0 + wq * TQ * 32; acc.wp = wp; acc.wq = wq;
#pragma unroll
    for (int i = 0; i < TP; ++i)
#pragma unroll
        for (int j = 0; j < TQ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc.v[i][j][r] = 0.f;
    dma_p(0, 0);
    load_q(0);
    store_q();
    __syncthreads();
    const int frow = lane & 31, fh = lane >> 5;
    for (int kt = 0; kt < ksteps; ++kt) {
        const bool more = kt + 1 < ksteps;
        if (more) { dma_p(kt + 1, (kt + 1) & 1); load_q(kt + 1); }
        const float* sp = pbuf + (kt & 1) * BP * BK + (wp * TP * 32) * BK;
        const float* sq = qbuf + (wq * TQ * 32) * BK;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x4 a[TP], b[TQ];
#pragma unroll
            for (int i = 0; i < TP; ++i) a[i] = *reinterpret_cast<const f32x4*>(sp + lds_slot(i * 32 + frow, 2 * c + fh));
#pragma unroll
            for (int j = 0; j < TQ; ++j) b[j] = *reinterpret_cast<const f32x4*>(sq + lds_slot(j * 32 + frow, 2 * c + fh));
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                for (int i = 0; i < TP; ++i)
#pragma unroll
                    for (int j = 0; j < TQ; ++j)
                        acc.v[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s2], b[j][s2], acc.v[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) { store_q(); __syncthreads(); }
    }
    epi(acc, smem);
}

template <class S, class LQ, class E>
double run_dma(const char* label, DenseRows lp, LQ lq, E epi, int K, long long prow, long long qrow, int reps = 20) {
    auto kern = gemm_dma_kernel<S, LQ, E>;
    constexpr size_t lds = (2 * S::BP + S::BQ) * BK * 4;   // P double buffer + Q
    static bool done = false;
    if (!done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); done = true; }
    TileMap tm; tm.tiles_small = (int)((prow + S::BP - 1) / S::BP); tm.tiles_big = (int)((qrow + S::BQ - 1) / S::BQ);
    int ksteps = K / BK;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(tm.grid()), dim3(S::NT), lds, 0, lp, lq, epi, ksteps, tm);
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(tm.grid()), dim3(S::NT), lds, 0, lp, lq, epi, ksteps, tm);
    (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b); ms /= reps;
    double tf = 2.0 * prow * qrow * K / (ms * 1e-3) / 1e12;
    printf("%-44s N=%5lld rows=%7lld K=%4d  %8.3f ms  %7.2f TF  (%.1f%% of 157.3) [%s]\n", label, prow, qrow, K, ms, tf, tf / 157.3 * 100, hipGetErrorString(hipGetLastError()));
    return tf;
}

}  // namespace amdrec
using namespace amdrec;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <class S, class LP, class LQ, class E>
double run(const char* label, LP lp, LQ lq, E epi, int K, long long prow, long long qrow, int reps = 20) {
    hipStream_t st = nullptr;
    for (int i = 0; i < 3; ++i) CK((launch_gemm<S, true>(lp, lq, epi, K, prow, qrow, st)));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, st));
    for (int i = 0; i < reps; ++i) CK((launch_gemm<S, true>(lp, lq, epi, K, prow, qrow, st)));
    CK(hipEventRecord(b, st));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    double tf = 2.0 * prow * qrow * K / (ms * 1e-3) / 1e12;
    printf("%-44s N=%5lld rows=%7lld K=%4d  %8.3f ms  %7.2f TF  (%.1f%% of 157.3)\n", label, prow, qrow, K, ms, tf, tf / 157.3 * 100);
    return tf;
}

int main() {
    const long long rows = 256000;
    float *X, *W, *Y, *H;
    CK(hipMalloc(&X, rows * 1024 * 4)); CK(hipMalloc(&W, 1024 * 1024 * 4)); CK(hipMalloc(&Y, rows * 1024 * 4));
    CK(hipMalloc(&H, 4096));
    CK(hipMemset(X, 0, rows * 1024 * 4)); CK(hipMemset(W, 0, 1024 * 1024 * 4));
    std::vector<float> h(1 << 20);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    for (int i = 0; i < 250; ++i) CK(hipMemcpy(X + (size_t)i * (1 << 20), h.data(), 4 << 20, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, h.data(), 4 << 20, hipMemcpyHostToDevice));
    using S = Shape<2, 2, 4, 2>;
    for (int K : {256, 1024}) {   // warm-up pass (the first configuration measured in a process reads ~20 % low)
        DenseRows lw{W, 256, K, K, 30, 1ll << 30};
        DenseRows lx{X, rows, K, K, 30, 1ll << 30};
        run<S>("warm-up", lw, lx, EpiStore{Y, 256, rows}, K, 256, rows);
    }
    for (int rep = 0; rep < 2; ++rep)
        for (int K : {256, 1024}) {
            DenseRows lw{W, 256, K, K, 30, 1ll << 30};
            DenseRows lx{X, rows, K, K, 30, 1ll << 30};
            run<S>("reg-staged P+Q: full-line store", lw, lx, EpiStoreLines{Y, 256, rows}, K, 256, rows);
            run_dma<S>("LDS-DMA P, reg Q: full-line store", lw, lx, EpiStoreLines{Y, 256, rows}, K, 256, rows);
            run<S>("reg-staged P+Q: sink", lw, lx, EpiSink{H}, K, 256, rows);
            run_dma<S>("LDS-DMA P, reg Q: sink", lw, lx, EpiSink{H}, K, 256, rows);
        }
    // correctness of the DMA variant against the register-staged one
    {
        int K = 256;
        DenseRows lw{W, 256, K, K, 30, 1ll << 30};
        DenseRows lx{X, 4096, K, K, 30, 1ll << 30};
        float* Y2; CK(hipMalloc(&Y2, 4096 * 256 * 4));
        CK(hipMemset(Y, 0, 4096 * 256 * 4)); CK(hipMemset(Y2, 0, 4096 * 256 * 4));
        run<S>("check: reg", lw, lx, EpiStoreLines{Y, 256, 4096}, K, 256, 4096, 1);
        run_dma<S>("check: dma", lw, lx, EpiStoreLines{Y2, 256, 4096}, K, 256, 4096, 1);
        std::vector<float> h1(4096 * 256), h2(4096 * 256);
        CK(hipMemcpy(h1.data(), Y, h1.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h2.data(), Y2, h2.size() * 4, hipMemcpyDeviceToHost));
        double md = 0; for (size_t i = 0; i < h1.size(); ++i) md = fmax(md, fabs((double)h1[i] - h2[i]));
        printf("DMA vs register staging: max |diff| = %g (h1[5]=%g)\n", md, h1[5]);
    }
    using S64 = Shape<2, 2, 4, 1>;     // 256 x 64 tile, 64 acc regs
    using S64b = Shape<4, 1, 2, 2>;    // 256 x 64 tile, waves split over features only
    for (int rep = 0; rep < 0; ++rep)
        for (int K : {256, 1024}) {
            DenseRows lw{W, 256, K, K, 30, 1ll << 30};
            DenseRows lx{X, rows, K, K, 30, 1ll << 30};
            run<S>("256x128 (2 wg/CU): real loads, full-line store", lw, lx, EpiStoreLines{Y, 256, rows}, K, 256, rows);
            run<S64>("256x64 <2,2,4,1>: real loads, full-line store", lw, lx, EpiStoreLines{Y, 256, rows}, K, 256, rows);
            run<S64b>("256x64 <4,1,2,2>: real loads, full-line store", lw, lx, EpiStoreLines{Y, 256, rows}, K, 256, rows);
        }
    for (int rep = 0; rep < 0; ++rep)
        for (int K : {256, 1024})
            for (int pad : {0, 8, 32, 96}) {      // leading dimension K + pad floats (power-of-two stride vs padded)
                char lab[64];
                snprintf(lab, sizeof(lab), "ld = K + %d: real loads, sink", pad);
                DenseRows lw{W, 256, K + pad, K, 30, 1ll << 30};
                DenseRows lx{X, rows * 1024 / (K + pad) < rows ? rows * 1024 / (K + pad) : rows, K + pad, K, 30, 1ll << 30};
                run<S>(lab, lw, lx, EpiSink{H}, K, 256, rows * 1024 / (K + pad) < rows ? rows * 1024 / (K + pad) : rows);
            }
    for (int K : {256, 1024}) {
        int N = (K == 256) ? 256 : 256;
        DenseRows lw{W, N, K, K, 30, 1ll << 30};
        DenseRows lx{X, rows, K, K, 30, 1ll << 30};
        NullRows nz{0.5f};
        run<S>("4w 256x128: real loads, store epilogue", lw, lx, EpiStore{Y, N, rows}, K, N, rows);
        run<S>("4w 256x128: real loads, full-line store epi", lw, lx, EpiStoreLines{Y, N, rows}, K, N, rows);
        run<S>("4w 256x128: real loads, sink epilogue", lw, lx, EpiSink{H}, K, N, rows);
        run<S>("4w 256x128: null Q (rows), real P (weights)", lw, nz, EpiSink{H}, K, N, rows);
        run<S>("4w 256x128: real Q, null P", nz, lx, EpiSink{H}, K, N, rows);
        run<S>("4w 256x128: null loads (LDS+MFMA only)", nz, nz, EpiSink{H}, K, N, rows);
    }
    {   // FFN1-like: N = 1024
        int K = 256, N = 1024;
        DenseRows lw{W, N, K, K, 30, 1ll << 30};
        DenseRows lx{X, rows, K, K, 30, 1ll << 30};
        run<S>("4w 256x128: N=1024 real loads, store", lw, lx, EpiStore{Y, N, rows}, K, N, rows);
        run<S>("4w 256x128: N=1024 real loads, full-line store", lw, lx, EpiStoreLines{Y, N, rows}, K, N, rows);
    }
    return 0;
}
