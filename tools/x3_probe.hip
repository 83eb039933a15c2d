// Elimination runs of the row-owner kernel (csrc/rowowner.hpp): the product kernel compiled with AMDREC_X3_DBG switches,
// timed on the full chain (3 encoder layers, 3 cross, 3 heads) over 256000 synthetic rows.  Numerics are meaningless
// here (random fragment bits); only time is read.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DAMDREC_X3_DBG=<bits> tools/x3_probe.hip -o tools/bin/x3_probe_<bits>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../movie-recommender-demo_amd/csrc/rowowner.hpp"
#include "../movie-recommender-demo_amd/csrc/rowowner16.hpp"
#ifndef AMDREC_X3_VARIANT
#define AMDREC_X3_VARIANT 32
#endif
#ifndef PROBE_PARAM_FLOATS
#define PROBE_PARAM_FLOATS x3::PARAM_FLOATS
#endif
#ifndef AMDREC_X3_PROBE_NS           // x3b: 8 waves x 16 rows per workgroup; x3b4: 4 waves (the small-pass shape)
#define AMDREC_X3_PROBE_NS x3b
#endif
#if AMDREC_X3_VARIANT == 16
#define KERNEL AMDREC_X3_PROBE_NS::ranker_x3b_kernel
#define NTHREADS (64 * AMDREC_X3_PROBE_NS::WAVES)
#define ROWS_WG (16 * AMDREC_X3_PROBE_NS::WAVES)
#else
#define KERNEL x3::ranker_x3_kernel
#define NTHREADS 256
#define ROWS_WG 128
#endif
using namespace amdrec;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    const long long rows = argc > 1 ? atoll(argv[1]) : 256000;
    const int L = 3, C = 3, T = 3, dff = 1024, h1 = 256;
    const long long chunks = L * (16 + 4ll * (dff / 32)) + 16ll * C + (long long)T * (h1 / 32) * 40 / 16;
    std::vector<uint16_t> st((size_t)chunks * 8192);
    srand(1);
    for (auto& v : st) {   // fp16 with exponent in [2^4, 2^11], random sign and mantissa
        const uint16_t e = 15 + 4 + rand() % 8;
        v = (uint16_t)(((rand() & 1) << 15) | (e << 10) | (rand() & 0x3ff));
    }
    if (argc > 2 && atoi(argv[2]) == 1) std::fill(st.begin(), st.end(), (uint16_t)0);     // zero weights: data-dependent power (DVFS) check
    uint16_t* dstream; float *dX, *dpar, *dscratch, *dlog;
    CK(hipMalloc(&dstream, st.size() * 2));
    CK(hipMemcpy(dstream, st.data(), st.size() * 2, hipMemcpyHostToDevice));
    std::vector<float> X((size_t)rows * 256);
    for (auto& v : X) v = (argc > 2 && atoi(argv[2]) == 2) ? 0.f : (float)(rand() % 2001 - 1000) / 500.f;   // 2: zero activations
    CK(hipMalloc(&dX, X.size() * 4));
    CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> par(x3::PARAM_FLOATS, 0.01f);     // the blob: every parameter 0.01 except "gamma" = 1 at 2048..4095
    for (int i = 2048; i < 4096; ++i) par[i] = 1.0f;
    CK(hipMalloc(&dpar, par.size() * 4));
    CK(hipMemcpy(dpar, par.data(), par.size() * 4, hipMemcpyHostToDevice));
    const long long prow = (rows + 127) / 128 * 128;
    CK(hipMalloc(&dscratch, (size_t)prow * 1024));
    CK(hipMalloc(&dlog, (size_t)rows * 4 * 4));
    x3::Program G{};
    int n = 0;
    const float sw = 65536.f * 64.f;     // packed weights ~2^10 -> real ~2^-12 .. 2^-5
    for (int l = 0; l < L; ++l) {
        x3::Phase& A = G.ph[n++];
        A.type = x3::PH_ATTN_LN; A.b1 = 0; A.gamma = 2048; A.beta = 0; A.sw1 = sw; A.sw2 = 1; A.ln_eps = 1e-5f;
        x3::Phase& F = G.ph[n++];
        F.type = x3::PH_FFN_LN; F.n_steps = dff / 32; F.b1 = 0; F.b2 = 0; F.gamma = 2048; F.beta = 0;
        F.sw1 = sw; F.sw2 = sw; F.hn = 16.f * 0.5f; F.hb = 0.01f; F.ln_eps = 1e-5f;
    }
    for (int c = 0; c < C; ++c) { x3::Phase& P = G.ph[n++]; P.type = x3::PH_CROSS; P.b1 = 0; P.sw1 = sw; P.sw2 = 1; }
    x3::Phase& H = G.ph[n++];
    H.type = x3::PH_HEADS; H.n_steps = h1 / 32; H.n_tasks = T; H.b1 = 0; H.sw1 = sw; H.sw2 = sw; H.hn = 8.f; H.hb = 0.01f;
    for (int t = 0; t < T; ++t) { G.hb2[t] = 0; G.hw3[t] = 64; G.hb3[t] = 128; }
    G.params = dpar; G.n_params = PROBE_PARAM_FLOATS;
    G.n_phases = n - (argc > 3 ? atoi(argv[3]) : 0);   // argv[3]: drop the last phases (time of a prefix of the chain)
    G.total_chunks = (int)chunks; G.stream = (const unsigned char*)dstream;
    x3::Input in{};
    in.X = dX; in.ldx = 256;
    CK(hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, x3::RING_BYTES + PROBE_PARAM_FLOATS * 4));
    const unsigned grid = (unsigned)((rows + ROWS_WG - 1) / ROWS_WG);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(NTHREADS), x3::RING_BYTES + PROBE_PARAM_FLOATS * 4, 0, G, in, rows, dscratch, (float*)nullptr, 0ll, dlog, rows);
    CK(hipDeviceSynchronize());
    const int reps = 10;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(NTHREADS), x3::RING_BYTES + PROBE_PARAM_FLOATS * 4, 0, G, in, rows, dscratch, (float*)nullptr, 0ll, dlog, rows);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    if (AMDREC_X3_DBG & 16) {              // cycle stamps of every wave (rowowner16.hpp, DBG & 16)
        std::vector<float> lg((size_t)rows * 4);
        CK(hipMemcpy(lg.data(), dlog, lg.size() * 4, hipMemcpyDeviceToHost));
        const long long nw = (long long)grid * (NTHREADS / 64);
        double tot = 0, w = 0, b = 0, d = 0;
        for (long long i = 0; i < nw; ++i) {
            const float* p = &lg[3 * rows + i * 4];
            tot += p[0]; w += p[1]; b += p[2]; d += p[3];
        }
        printf("per wave (cycles, mean of %lld waves): total %.0f  dma-wait %.0f (%.1f%%)  barrier %.0f (%.1f%%)  dma-issue %.0f (%.1f%%)  rest %.0f\n",
               nw, tot / nw, w / nw, 100 * w / tot, b / nw, 100 * b / tot, d / nw, 100 * d / tot, (tot - w - b - d) / nw);
    }
    const double flop = 2.0 * rows * (L * (65536.0 + 2 * 262144.0) + C * 65536.0 + T * (65536.0 + 16384.0 + 64.0));
    printf("variant=%d DBG=%d rows=%lld: %.3f ms  %.1f TF fp32-equivalent  (%.3f of 833 TF)\n", AMDREC_X3_VARIANT, AMDREC_X3_DBG, rows, ms, flop / ms / 1e9,
           flop / ms / 1e9 / 833.3);
    return 0;
}
