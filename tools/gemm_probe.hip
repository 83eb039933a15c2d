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
    using S8 = Shape<2, 4, 4, 2, true>;
    for (int K : {256, 1024}) {   // warm-up pass (the first configuration measured in a process reads ~20 % low)
        DenseRows lw{W, 256, K, K, 30, 1ll << 30};
        DenseRows lx{X, rows, K, K, 30, 1ll << 30};
        run<S>("warm-up", lw, lx, EpiStore{Y, 256, rows}, K, 256, rows);
    }
    using S64 = Shape<2, 2, 4, 1>;     // 256 x 64 tile, 64 acc regs
    using S64b = Shape<4, 1, 2, 2>;    // 256 x 64 tile, waves split over features only
    for (int rep = 0; rep < 2; ++rep)
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
        run<S8>("8w 256x256 dbuf: real loads, store epilogue", lw, lx, EpiStore{Y, N, rows}, K, N, rows);
        run<S8>("8w 256x256 dbuf: null loads", nz, nz, EpiSink{H}, K, N, rows);
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
