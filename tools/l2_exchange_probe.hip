// would splitting the single-request ranker's weight stream over C CUs per 16-row tile pay?
//
// The column-split kernel (csrc/rowowner16c.hpp) is bound by ONE CU pulling the 8.8 MB weight stream through LDS-DMA
// (84 us).  With the stream split over C workgroups on C CUs (each owns 1/C of every layer's output features) a workgroup
// streams 1/C of the bytes, but after every GEMM phase the C workgroups must exchange their slices of the 16 x 256 fp32
// row state through L2 (an all-gather of 16 KB per group) - 13 times per pass (10 phases + a partial-sum reduction per FFN).
// This probe measures exactly that trade with nothing else in the kernel: G groups of C workgroups, each workgroup
// streams chunks / C chunks of 16 KB through the column-split kernel's ring (4 waves, 4 chunks in flight) and at NEX evenly
// spaced points publishes its 16 KB / C slice (plain stores, every wave drained, a barrier, one lane's agent-scope release,
// a relaxed agent flag: a valid form of MI355X_MICROARCH.md) and collects its peers' slices (relaxed poll, agent acquire,
// barrier, plain loads).  C = 1 is the kernel as it stands (no exchange).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/l2_exchange_probe.hip -o tools/bin/l2_exchange_probe
// Every spin is bounded (a peer that never arrives sets an error flag instead of hanging the box).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int CHUNK = 16384, NBUF = 4, W = 4, DEPTH = 3;
constexpr int ROWSTATE = 16 * 256 * 4;                    // 16 rows x 256 features x fp32
typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef float f4 __attribute__((ext_vector_type(4)));

template <int C>
__global__ __launch_bounds__(64 * W) void split_stream_kernel(const unsigned char* src, int chunks_total, int nex,
                                                              unsigned char* xbuf, int* flags, int* err, float* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = blockIdx.x / C, me = blockIdx.x % C;
    constexpr int PER = CHUNK / 1024 / W;
    constexpr int SLICE = ROWSTATE / C;                   // bytes this workgroup publishes per exchange
    const int chunks = chunks_total / C;                  // this workgroup's share of the stream
    const unsigned char* mysrc = src + (size_t)me * chunks * CHUNK;
    lds_byte* lds = (lds_byte*)smem;
    int issued = 0;
    auto issue = [&]() {
        const int c = issued < chunks ? issued : chunks - 1;
        const unsigned char* s = mysrc + (size_t)c * CHUNK + (wave * PER) * 1024 + lane * 16;
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
    const int every = nex > 0 ? (chunks + nex - 1) / nex : chunks + 1;
    int ex = 0;
    // exchange buffers: [group][workgroup][parity][SLICE]; flags[group][workgroup] = exchanges published so far
    unsigned char* gbuf = xbuf + (size_t)group * C * 2 * SLICE;
    int* gflag = flags + group * C;
    for (int c = 0; c < chunks; ++c) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER * (DEPTH - 1)) : "memory");
        __builtin_amdgcn_s_barrier();
        issue();
        const f4 v = *reinterpret_cast<const __attribute__((address_space(3))) f4*>(lds + (c % NBUF) * CHUNK + (wave % 16) * 1024 + lane * 16);
        acc += v[0];
        if (C > 1 && (c + 1) % every == 0 && ex < nex) {
            // publish my slice: 256 threads x 16 B per pass, write-through
            unsigned char* mine = gbuf + ((size_t)me * 2 + (ex & 1)) * SLICE;
            for (int o = tid * 16; o < SLICE; o += 64 * W * 16) {
                const f4 p{acc, (float)c, (float)ex, (float)me};
                *reinterpret_cast<f4*>(mine + o) = p;       // plain stores; the release fence below writes them back
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&gflag[me], ex + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // wait for every peer's publication of this exchange (bounded)
                for (int p = 0; p < C; ++p) {
                    if (p == me) continue;
                    int spins = 0;
                    while (__hip_atomic_load(&gflag[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < ex + 1) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > (1 << 22)) { atomicExch(err, 1); break; }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            // collect the peers' slices (every thread 16 B per pass, 8 loads in flight)
            for (int p = 0; p < C; ++p) {
                if (p == me) continue;
                const unsigned char* theirs = gbuf + ((size_t)p * 2 + (ex & 1)) * SLICE;
                for (int o = tid * 16; o < SLICE; o += 64 * W * 16) acc += (*reinterpret_cast<const f4*>(theirs + o))[0];
            }
            ++ex;
            // (the DMA issued before the exchange is still in flight: its vmcnt accounting restarts with the loads above
            //  complete - they were consumed - so the counted wait of the next iteration still covers the ring)
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 123.456f) out[0] = acc;
}

template <int C>
int run(const unsigned char* d, int chunks, int G, int nex, unsigned char* xbuf, int* flags, int* err, float* out) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute((const void*)split_stream_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    float best = 1e9f;
    for (int it = 0; it < 12; ++it) {
        CK(hipMemsetAsync(flags, 0, 4096 * sizeof(int)));
        CK(hipEventRecord(a));
        // 96 KB of LDS per workgroup: ONE workgroup per CU, like the real kernel (140 KB)
        hipLaunchKernelGGL((split_stream_kernel<C>), dim3(G * C), dim3(64 * W), 96 * 1024, 0, d, chunks, nex, xbuf, flags, err, out);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (it >= 2 && ms < best) best = ms;
    }
    int herr = 0;
    CK(hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost));
    printf("C=%d groups=%3d (%3d workgroups) exchanges=%2d: %7.1f us%s\n", C, G, G * C, nex, best * 1e3, herr ? "  (SPIN TIMEOUT)" : "");
    return 0;
}

int main(int argc, char** argv) {
    const int chunks = argc > 1 ? atoi(argv[1]) : 528;   // 528 x 16 KB = 8.65 MB
    unsigned char *d, *xbuf; float* out; int *flags, *err;
    CK(hipMalloc(&d, (size_t)chunks * CHUNK)); CK(hipMemset(d, 1, (size_t)chunks * CHUNK)); CK(hipMalloc(&out, 64));
    CK(hipMalloc(&xbuf, (size_t)64 * 4 * 2 * ROWSTATE)); CK(hipMalloc(&flags, 4096 * sizeof(int))); CK(hipMalloc(&err, sizeof(int)));
    CK(hipMemset(err, 0, sizeof(int)));
    for (int G : {1, 32}) {                               // one tile; one request = 500 rows = 32 tiles of 16 rows
        if (run<1>(d, chunks, G, 0, xbuf, flags, err, out)) return 1;
        for (int nex : {0, 13}) {
            if (run<2>(d, chunks, G, nex, xbuf, flags, err, out)) return 1;
            if (run<4>(d, chunks, G, nex, xbuf, flags, err, out)) return 1;
        }
    }
    return 0;
}
