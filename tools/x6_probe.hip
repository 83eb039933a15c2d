// Dev probe (not product): error-compensated fp32 GEMM on the bf16 MFMA ("bf16x6").
//   C[m][n] = sum_k X[m][k] * W[n][k], X / W fp32, each split exactly into three bf16 planes h + m + l (truncation
//   split, 8 + 8 + 8 mantissa bits); products hh, hm, mh, mm, hl, lh are accumulated in fp32 by
//   v_mfma_f32_32x32x16_bf16 (the dropped ml, lm, ll terms are <= 2^-23 relative per product).
//   6 bf16 MFMAs per fp32 MFMA-equivalent: 16/6 = 2.67x the fp32 MFMA peak on paper.
// Weights pre-split on the host (static), activations split on the fly while staging.
// Tile 256 (W rows) x 256 (X rows), 8 waves (2 x 4), wave tile 128 x 64, BK = 32, one LDS buffer + register prefetch.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/x6_probe.hip -o tools/bin/x6_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int BP = 256, BQ = 256, BK = 32, NT = 512;
constexpr int PLANE_BYTES = 256 * 4 * 16;          // one plane of one operand per K-step: 256 rows x 4 chunks x 16 B

__device__ __forceinline__ int lds_off(int plane, int row, int chunk) {
    return ((plane * 256 + row) * 4 + (chunk ^ ((row >> 2) & 3))) * 16;
}

// exact 3-way truncation split of 8 floats -> three packed bf16x8 (as 4 dwords each)
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, u32x4& h, u32x4& m, u32x4& l) {
    float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    uint32_t hb[8], mb[8], lb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const uint32_t u = __float_as_uint(x[i]);
        const float hf = __uint_as_float(u & 0xffff0000u);
        const float r1 = x[i] - hf;                         // exact
        const uint32_t ru = __float_as_uint(r1);
        const float mf = __uint_as_float(ru & 0xffff0000u);
        const float r2 = r1 - mf;                           // exact, <= 8 significant bits
        hb[i] = u; mb[i] = ru; lb[i] = __float_as_uint(r2);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {                           // pack the high halves of two floats into one dword
        h[i] = __builtin_amdgcn_perm(hb[2 * i + 1], hb[2 * i], 0x07060302u);
        m[i] = __builtin_amdgcn_perm(mb[2 * i + 1], mb[2 * i], 0x07060302u);
        l[i] = __builtin_amdgcn_perm(lb[2 * i + 1], lb[2 * i], 0x07060302u);
    }
}

// MODE bits: 1 = store C, 2 = skip global loads after the first K-step, 4 = skip the split (store raw planes),
// 8 = skip LDS fragment reads after the first K-step, 16 = skip the staging stores + second barrier
template <int MODE>
__global__ __launch_bounds__(NT, 1) void x6_kernel(const uint16_t* __restrict__ Wp,   // [N][K/32][3][32] bf16 planes
                                                   const float* __restrict__ X,       // [M][K] fp32
                                                   float* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sP = smem;                      // 3 planes
    unsigned char* sQ = smem + 3 * PLANE_BYTES;    // 3 planes
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = w >> 2, wq = w & 3;
    const int tiles_n = N / BP;
    const int tn = blockIdx.x % tiles_n, tm = blockIdx.x / tiles_n;
    const int ksteps = K / BK;

    // staging maps
    // P: 3072 chunks per K-step: L = u * 512 + tid -> row L / 12, within L % 12 (plane = within / 4, chunk = within % 4)
    // Q: 1024 8-float chunks: L = u * 512 + tid -> row L / 4, chunk L % 4
    u32x4 rp[6];
    f32x4 rq[4];
    auto load_stage = [&](int kt) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int L = u * NT + tid, row = L / 12, within = L % 12;
            const uint16_t* g = Wp + ((size_t)(tn * BP + row) * ksteps + kt) * 96 + within * 8;
            rp[u] = *reinterpret_cast<const u32x4*>(g);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int L = u * NT + tid, row = L / 4, c = L % 4;
            const float* g = X + (size_t)(tm * BQ + row) * K + kt * BK + c * 8;
            rq[2 * u] = *reinterpret_cast<const f32x4*>(g);
            rq[2 * u + 1] = *reinterpret_cast<const f32x4*>(g + 4);
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int L = u * NT + tid, row = L / 12, within = L % 12;
            *reinterpret_cast<u32x4*>(sP + lds_off(within / 4, row, within % 4)) = rp[u];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int L = u * NT + tid, row = L / 4, c = L % 4;
            u32x4 h, m, l;
            if (MODE & 4) { h = __builtin_bit_cast(u32x4, rq[2 * u]); m = __builtin_bit_cast(u32x4, rq[2 * u + 1]); l = h; }
            else split8(rq[2 * u], rq[2 * u + 1], h, m, l);
            *reinterpret_cast<u32x4*>(sQ + lds_off(0, row, c)) = h;
            *reinterpret_cast<u32x4*>(sQ + lds_off(1, row, c)) = m;
            *reinterpret_cast<u32x4*>(sQ + lds_off(2, row, c)) = l;
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_stage(0);
    store_stage();
    __syncthreads();
    const int frow = lane & 31, fh = lane >> 5;
    for (int kt = 0; kt < ksteps; ++kt) {
        const bool more = kt + 1 < ksteps;
        if (more && !(MODE & 2)) load_stage(kt + 1);
        bf16x8 b[2][3];
        bf16x8 a3[3][4];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int chunk = 2 * s + fh;
            if (!(MODE & 8) || kt == 0)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    b[j][p] = *reinterpret_cast<const bf16x8*>(sQ + lds_off(p, wq * 64 + j * 32 + frow, chunk));
            // A plane h with B planes l, m, h; A plane m with B planes m, h; A plane l with B plane h (small terms first)
#pragma unroll
            for (int pa = 0; pa < 3; ++pa) {
                bf16x8 (&a)[4] = a3[pa];
                if (!(MODE & 8) || kt == 0)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    a[i] = *reinterpret_cast<const bf16x8*>(sP + lds_off(pa, wp * 128 + i * 32 + frow, chunk));
#pragma unroll
                for (int pb = 2 - pa; pb >= 0; --pb)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j][pb], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
        if (more && !(MODE & 16)) {
            store_stage();
            __syncthreads();
        }
    }
    // epilogue: C[m][n], lane&31 = m (X row), registers = n (W row)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const size_t mrow = (size_t)tm * BQ + wq * 64 + j * 32 + frow;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n0 = tn * BP + wp * 128 + i * 32 + 8 * g + 4 * fh;
                f32x4 v{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                if ((MODE & 1) || v[0] == 12345.678f) *reinterpret_cast<f32x4*>(C + mrow * N + n0) = v;
            }
    }
}

static void split3(float x, uint16_t& h, uint16_t& m, uint16_t& l) {
    uint32_t u; memcpy(&u, &x, 4);
    uint32_t hu = u & 0xffff0000u; float hf; memcpy(&hf, &hu, 4);
    float r1 = x - hf; uint32_t ru; memcpy(&ru, &r1, 4);
    uint32_t mu = ru & 0xffff0000u; float mf; memcpy(&mf, &mu, 4);
    float r2 = r1 - mf; uint32_t lu; memcpy(&lu, &r2, 4);
    h = hu >> 16; m = mu >> 16; l = lu >> 16;
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 65536, N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 256;
    if (M % BQ || N % BP || K % BK) { printf("bad sizes\n"); return 1; }
    printf("x6 probe: M=%d N=%d K=%d\n", M, N, K);
    std::vector<float> hX((size_t)M * K), hW((size_t)N * K);
    srand(1);
    for (auto& v : hX) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
    for (auto& v : hW) v = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    std::vector<uint16_t> hWp((size_t)N * K * 3);
    const int ks = K / 32;
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            uint16_t h, m, l; split3(hW[(size_t)n * K + k], h, m, l);
            size_t base = ((size_t)n * ks + k / 32) * 96 + (k % 32);
            hWp[base] = h; hWp[base + 32] = m; hWp[base + 64] = l;
        }
    float *dX, *dC; uint16_t* dWp;
    CK(hipMalloc(&dX, hX.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4)); CK(hipMalloc(&dWp, hWp.size() * 2));
    CK(hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dWp, hWp.data(), hWp.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(dC, 0, (size_t)M * N * 4));
    const size_t lds = 6 * PLANE_BYTES;
    const int grid = (M / BQ) * (N / BP);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](auto kern, const char* label) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, 0, dWp, dX, dC, M, N, K);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        const int reps = 20;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, 0, dWp, dX, dC, M, N, K);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("%-46s %.3f ms  %6.1f TFLOP/s fp32-equivalent\n", label, ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
    };
    run(x6_kernel<1>, "full, store");
    run(x6_kernel<1>, "full, store (again)");
    run(x6_kernel<0>, "full, sink");
    run(x6_kernel<4>, "no split");
    run(x6_kernel<2>, "no global loads");
    run(x6_kernel<2 | 4>, "no global loads, no split");
    run(x6_kernel<2 | 4 | 16>, "no loads/split/staging stores");
    run(x6_kernel<2 | 4 | 16 | 8>, "MFMA only (1 barrier per K-step)");
    run(x6_kernel<8>, "no LDS fragment reads, everything else");
    // accuracy: x6 vs fp64 and (for scale) an fp32 fma chain vs fp64, on a sample of outputs
    hipLaunchKernelGGL(x6_kernel<1>, dim3(grid), dim3(NT), lds, 0, dWp, dX, dC, M, N, K);
    CK(hipDeviceSynchronize());
    std::vector<float> hC((size_t)1024 * N);
    CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double e6 = 0, e32 = 0, scale = 0;
    for (int m = 0; m < 1024; m += 7)
        for (int n = 0; n < N; n += 13) {
            double ref = 0; float f = 0.f;
            for (int k = 0; k < K; ++k) {
                ref += (double)hX[(size_t)m * K + k] * hW[(size_t)n * K + k];
                f = fmaf(hX[(size_t)m * K + k], hW[(size_t)n * K + k], f);
            }
            e6 = fmax(e6, fabs(hC[(size_t)m * N + n] - ref));
            e32 = fmax(e32, fabs(f - ref));
            scale = fmax(scale, fabs(ref));
        }
    printf("max |err| vs fp64: x6 %.3e   fp32 fma chain %.3e   (max |value| %.3f)\n", e6, e32, scale);
    return 0;
}
