// Dev probe (not product): times csrc/gemm_core.hpp's gemm_x6_kernel with parts of the K loop switched off (DBG bits).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/x6v2_probe.hip -o tools/bin/x6v2_probe
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

struct EpiSink {    // keeps the accumulators alive, writes (almost) nothing
    static constexpr const char* name = "sink";
    static constexpr double out_bytes_per_elem = 0.0;
    static constexpr size_t lds_bytes(int) { return 0; }
    float* out;
    template <class A>
    __device__ void operator()(A& acc, float*) const {
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

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int DBG>
static void run(const char* label, const uint16_t* Wp, const float* X, float* H, int M, int N, int K) {
    using S = ShapeX6;
    auto kern = gemm_x6_kernel<DenseRows, EpiSink, DBG>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS_BYTES));
    TileMap tm; tm.tiles_small = N / S::BP; tm.tiles_big = M / S::BQ;
    PlaneRows lp{Wp, N, K / 16};
    DenseRows lq{X, M, K, K, 30, 1ll << 30};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(tm.grid()), dim3(S::NT), S::LDS_BYTES, 0, lp, lq, EpiSink{H}, K / 16, tm);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(tm.grid()), dim3(S::NT), S::LDS_BYTES, 0, lp, lq, EpiSink{H}, K / 16, tm);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    printf("%-52s %.3f ms  %6.1f TFLOP/s fp32-equivalent\n", label, ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 262144, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 256;
    printf("x6 v2 probe: M=%d N=%d K=%d (sink epilogue)\n", M, N, K);
    std::vector<float> hX((size_t)M * K);
    std::vector<uint16_t> hW((size_t)N * K * 3);
    srand(1);
    for (auto& v : hX) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
    for (auto& v : hW) v = (uint16_t)(0x3c00 + (rand() & 0xff));      // arbitrary finite bf16 patterns
    float *dX, *dH; uint16_t* dW;
    CK(hipMalloc(&dX, hX.size() * 4)); CK(hipMalloc(&dH, 4096)); CK(hipMalloc(&dW, hW.size() * 2));
    CK(hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    run<0>("full", dW, dX, dH, M, N, K);
    run<0>("full (again)", dW, dX, dH, M, N, K);
    run<4>("no split / Q stores", dW, dX, dH, M, N, K);
    run<1 | 4>("no loads, no split (LDS reads + MFMA + barrier)", dW, dX, dH, M, N, K);
    run<1 | 4 | 8>("... and no barrier", dW, dX, dH, M, N, K);
    run<1 | 2 | 4>("MFMA + barrier only", dW, dX, dH, M, N, K);
    run<1 | 2 | 4 | 8>("MFMA only", dW, dX, dH, M, N, K);
    run<2>("no LDS fragment reads, rest as full", dW, dX, dH, M, N, K);
    return 0;
}
