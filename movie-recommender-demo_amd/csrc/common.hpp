// Common device/host helpers for libamdrec (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

namespace amdrec {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---- error plumbing (amdrec_last_error) -------------------------------------------
extern thread_local char g_err[512];
int set_error(int code, const char* fmt, ...);

#define AMDREC_OK 0
#define AMDREC_EINVAL (-1)
#define AMDREC_EHIP (-2)
#define AMDREC_EWORKSPACE (-3)
#define AMDREC_EINDEX (-4)

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e__ = (expr);                                                        \
        if (e__ != hipSuccess)                                                          \
            return amdrec::set_error(AMDREC_EHIP, "%s failed: %s (%s:%d)", #expr,       \
                                     hipGetErrorString(e__), __FILE__, __LINE__);       \
    } while (0)

#define REQUIRE(cond, ...)                                                              \
    do {                                                                                \
        if (!(cond)) return amdrec::set_error(AMDREC_EINVAL, __VA_ARGS__);              \
    } while (0)

// ---- order-preserving 64-bit candidate keys ---------------------------------------
// key = (orderable(score) << 32) | ~position : larger key == better candidate
// (higher score; equal score -> LOWER position wins).  Positions are unique, so keys of
// one query are a strict total order and every selection below is deterministic.
__host__ __device__ inline uint32_t f32_orderable(float f) {
    f += 0.0f;  // -0.0 -> +0.0 so equal floats give equal keys
    uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = __float_as_uint(f);
#else
    memcpy(&u, &f, 4);
#endif
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float f32_from_orderable(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f;
    memcpy(&f, &u, 4);
    return f;
#endif
}
// v0^2 + v1^2 + v2^2 + v3^2 as ONE explicit fma chain: the row-norm kernels (amdrec_l2_normalize, the towers' fused second
// normalisation) must round identically, and a sum of products left to the compiler is contracted differently from one
// context to the next
__device__ __forceinline__ float sumsq4(const f32x4& v) {
    return __builtin_fmaf(v[3], v[3], __builtin_fmaf(v[2], v[2], __builtin_fmaf(v[1], v[1], v[0] * v[0])));
}

__host__ __device__ inline unsigned long long make_key(float score, uint32_t pos) {
    return ((unsigned long long)f32_orderable(score) << 32) | (unsigned long long)(~pos);
}
__host__ __device__ inline float key_score(unsigned long long k) {
    return f32_from_orderable((uint32_t)(k >> 32));
}
__host__ __device__ inline uint32_t key_pos(unsigned long long k) { return ~(uint32_t)k; }

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// hipFuncSetAttribute applies to the CURRENT device only, so "done once" must be tracked per device (a process may
// drive several GPUs; a benign race between threads sets an attribute twice).
struct PerDeviceOnce {
    bool done[64] = {};
    bool pending() const {
        int d = 0;
        return hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64 || !done[d];
    }
    void mark() {
        int d = 0;
        if (hipGetDevice(&d) == hipSuccess && d >= 0 && d < 64) done[d] = true;
    }
};

// ---- optional per-launch timing (amdrec_profile_*): HIP events recorded on the launch stream
// around each GEMM launch, accumulated per kernel tag.  Off by default (zero cost).
struct ProfScope {
    hipEvent_t end;          // nullptr: not timed.  The scope keeps the EVENT, not an index into the record table: another
    hipStream_t st;          // thread may reset the table (amdrec_profile_enable) between this scope's two ends
    unsigned long long gen;  // generation of the record table the scope was opened in: a reset in between recycles its events,
                             // which may by then belong to another scope - the end record is skipped in that case
    ProfScope(const char* tag, double flops, double bytes, hipStream_t st);
    ~ProfScope();
};
extern std::atomic<bool> g_prof_on;

}  // namespace amdrec
