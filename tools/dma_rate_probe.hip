// Per-CU LDS-DMA stream rate: G workgroups of W waves each stream the SAME `bytes` (the ranker's weight stream is 8.6 MB)
// from L2 / MALL into a ring of NBUF 16 KB chunks with one barrier per chunk (the row-owner kernels' ring, nothing else).
// Sizes a column-split small-batch ranker: how fast can ONE CU take the whole weight stream?
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/dma_rate_probe.hip -o tools/bin/dma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int CHUNK = 16384, NBUF = 7;
typedef __attribute__((address_space(3))) unsigned char lds_byte;

template <int W, int DEPTH>
__global__ __launch_bounds__(64 * W) void stream_kernel(const unsigned char* src, int chunks, float* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int PER = CHUNK / 1024 / W;                 // 1 KB pieces per wave and chunk
    lds_byte* lds = (lds_byte*)smem;
    int issued = 0;
    auto issue = [&]() {
        const int c = issued < chunks ? issued : chunks - 1;
        const unsigned char* s = src + (size_t)c * CHUNK + (wave * PER) * 1024 + lane * 16;
        lds_byte* d = lds + (issued % NBUF) * CHUNK + (wave * PER) * 1024;
#pragma unroll
        for (int u = 0; u < PER; ++u)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + u * 1024),
                                             (__attribute__((address_space(3))) void*)(d + u * 1024), 16, 0, 0);
        ++issued;
    };
#pragma unroll
    for (int c = 0; c < DEPTH; ++c) issue();
    float acc = 0.f;
    for (int c = 0; c < chunks; ++c) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER * (DEPTH - 1)) : "memory");
        __builtin_amdgcn_s_barrier();
        issue();
        // one 16-byte read per lane and chunk, so the data is observed (each wave a different KB)
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 v = *reinterpret_cast<const __attribute__((address_space(3))) f4*>(lds + (c % NBUF) * CHUNK + (wave % 16) * 1024 + lane * 16);
        acc += v[0];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 123.456f) out[0] = acc;
}

template <int W, int DEPTH>
int run(const unsigned char* d, int chunks, int G, float* out) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute((const void*)stream_kernel<W, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, NBUF * CHUNK));
    float best = 1e9f;
    for (int it = 0; it < 12; ++it) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((stream_kernel<W, DEPTH>), dim3(G), dim3(64 * W), NBUF * CHUNK, 0, d, chunks, out);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (it >= 2 && ms < best) best = ms;
    }
    printf("waves=%d depth=%d workgroups=%4d: %7.1f us  -> %6.1f GB/s per workgroup, %7.1f GB/s in all\n", W, DEPTH, G, best * 1e3,
           chunks * (double)CHUNK / best / 1e6, G * chunks * (double)CHUNK / best / 1e6);
    return 0;
}

int main(int argc, char** argv) {
    const int chunks = argc > 1 ? atoi(argv[1]) : 528;   // 528 x 16 KB = 8.65 MB
    unsigned char* d; float* out;
    CK(hipMalloc(&d, (size_t)chunks * CHUNK)); CK(hipMemset(d, 1, (size_t)chunks * CHUNK)); CK(hipMalloc(&out, 64));
    for (int G : {1, 32, 256}) {
        if (run<4, 2>(d, chunks, G, out)) return 1;
        if (run<4, 3>(d, chunks, G, out)) return 1;
        if (run<4, 4>(d, chunks, G, out)) return 1;
        if (run<4, 6>(d, chunks, G, out)) return 1;
        if (run<8, 4>(d, chunks, G, out)) return 1;
        if (run<8, 6>(d, chunks, G, out)) return 1;
        if (run<16, 6>(d, chunks, G, out)) return 1;
    }
    return 0;
}
