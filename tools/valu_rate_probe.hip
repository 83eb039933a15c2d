// Issue cost of the vector instructions the row-owner kernel's plane split uses, on one wave per SIMD (gfx950):
// N independent instances of one instruction per loop iteration, timed with s_memtime.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate_probe.hip -o tools/bin/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* cyc, int iters) {
    float v[16];
    unsigned int h[16];
    for (int i = 0; i < 16; ++i) { v[i] = threadIdx.x * 0.5f + i; h[i] = threadIdx.x + i; }
    const float s = out[0];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define OPX(i)                                                                                                      \
    if (OP == 0) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(h[i]) : "v"(v[i]), "v"(s));                    \
    if (OP == 1) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(h[i]) : "v"(v[i]));                                       \
    if (OP == 2) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h[i]) : "v"(v[i]), "v"(v[(i + 1) & 15]));           \
    if (OP == 3) asm volatile("v_med3_f32 %0, %1, 0, %2" : "=v"(v[i]) : "v"(v[i]), "v"(s));                           \
    if (OP == 4) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[i]) : "v"(v[i]), "v"(s));                               \
    if (OP == 5) asm volatile("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "+v"(h[i]) : "v"(v[i]), "v"(s), "v"(h[(i + 1) & 15])); \
    if (OP == 6) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(v[i]) : "v"(h[i]));                                       \
    if (OP == 7) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(v[(i + 1) & 15]), "v"(s));                 \
    if (OP == 8) asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(v[i]) : "v"(v[(i + 1) & 15]), "v"(s));                \
    if (OP == 9) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h[i]) : "v"(v[i]), "v"(s));                      \
    if (OP == 10) asm volatile("v_pk_mul_f32 %0, %1, %1" : "=v"(*(double*)&v[i & 14]) : "v"(*(double*)&v[(i + 2) & 14]));
        REP16(OPX)
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0;
    for (int i = 0; i < 16; ++i) acc += v[i] + (float)h[i];
    out[1 + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
int run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    k<OP><<<1, 64>>>(out, cyc, iters);
    CK(hipDeviceSynchronize());
    k<OP><<<1, 64>>>(out, cyc, iters);
    CK(hipDeviceSynchronize());
    unsigned long long c;
    CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    // s_memtime ticks at the constant 100 MHz reference: report ticks per instruction and an estimate in shader clocks
    printf("%-28s %8.3f memtime ticks / instr\n", name, (double)c / (iters * 16.0));
    return 0;
}

int main() {
    float* out; unsigned long long* cyc;
    CK(hipMalloc(&out, 4096)); CK(hipMalloc(&cyc, 64));
    CK(hipMemset(out, 0, 4096));
    run<4>("v_mul_f32 (reference)", out, cyc);
    run<7>("v_fma_f32", out, cyc);
    run<3>("v_med3_f32", out, cyc);
    run<8>("v_max3_f32", out, cyc);
    run<1>("v_cvt_f16_f32", out, cyc);
    run<6>("v_cvt_f32_f16", out, cyc);
    run<2>("v_cvt_pk_f16_f32", out, cyc);
    run<0>("v_fma_mixlo_f16 (c = 0)", out, cyc);
    run<9>("v_fma_mixhi_f16 (c = 0)", out, cyc);
    run<5>("v_fma_mixlo_f16 (c = -f16)", out, cyc);
    run<10>("v_pk_mul_f32", out, cyc);
    return 0;
}
