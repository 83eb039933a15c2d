// Small per-row kernels around the GEMM-shaped stages (HBM-bound, coalesced float4 traffic).
#include "../../include/amdrec.h"
#include "common.hpp"

namespace amdrec {

// faiss.normalize_L2 (faiss_retrieval.py:115, :147): x *= 1/sqrt(sum x^2) for rows with
// non-zero norm.  One wave per row, float4 lanes; out may alias in.
__global__ __launch_bounds__(256) void l2_normalize_kernel(const float* in, long long ldi, float* out,
                                                           long long ldo, long long rows, int d) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const f32x4* src = reinterpret_cast<const f32x4*>(in + row * ldi);
    f32x4* dst = reinterpret_cast<f32x4*>(out + row * ldo);
    const int d4 = d >> 2;
    float ss = 0.f;
    for (int c = lane; c < d4; c += 64) {
        f32x4 v = src[c];
        ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float inv = ss > 0.f ? 1.0f / sqrtf(ss) : 1.0f;
    for (int c = lane; c < d4; c += 64) {
        f32x4 v = src[c];
        dst[c] = f32x4{v[0] * inv, v[1] * inv, v[2] * inv, v[3] * inv};
    }
}

// FAISSIndex.search id remap (faiss_retrieval.py:159-160): ids[i] = id_map[pos[i]];
// pos == -1 (unfilled slot) reads id_map[n-1], exactly like Python's id_map[-1].
__global__ void remap_ids_kernel(const long long* pos, const long long* id_map, long long n_map, long long* out,
                                 long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    long long p = pos[i];
    if (p < 0) p += n_map;
    out[i] = (p >= 0 && p < n_map) ? id_map[p] : -1;
}

}  // namespace amdrec

using namespace amdrec;

extern "C" int amdrec_l2_normalize(const float* in, int64_t ld_in, float* out, int64_t ld_out, int64_t rows,
                                   int dim, void* stream) {
    REQUIRE(dim >= 4 && dim % 4 == 0, "dim=%d must be a positive multiple of 4", dim);
    if (rows <= 0) return AMDREC_OK;
    REQUIRE(ld_in % 4 == 0 && ld_out % 4 == 0 && ld_in >= dim && ld_out >= dim, "bad leading dimension");
    REQUIRE(in && out, "null pointer");
    REQUIRE(((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0, "pointers must be 16-byte aligned");
    hipLaunchKernelGGL(l2_normalize_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), in, (long long)ld_in, out, (long long)ld_out,
                       (long long)rows, dim);
    HIP_TRY(hipGetLastError());
    return AMDREC_OK;
}

extern "C" int amdrec_remap_ids(const int64_t* pos, const int64_t* id_map, int64_t n_map, int64_t* out,
                                int64_t n, void* stream) {
    if (n <= 0) return AMDREC_OK;
    REQUIRE(pos && out && (id_map || n_map == 0), "null pointer");
    hipLaunchKernelGGL(remap_ids_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), (const long long*)pos, (const long long*)id_map,
                       (long long)n_map, (long long*)out, (long long)n);
    HIP_TRY(hipGetLastError());
    return AMDREC_OK;
}
